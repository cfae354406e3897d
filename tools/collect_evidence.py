"""Copy the summaries of a tools/evidence.sh run (gpurun_out/ev) into profiles/ under this round's names.
Usage: python tools/collect_evidence.py r03"""
import glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
ev, prof = os.path.join(ROOT, "gpurun_out", "ev"), os.path.join(ROOT, "profiles")


def cp(src, name):
    if os.path.exists(src):
        shutil.copy(src, os.path.join(prof, "%s_%s" % (tag, name)))
        print("  ", name)


def stats(d, name):
    f = glob.glob(os.path.join(ev, d, "**", "*kernel_stats.csv"), recursive=True)
    if f:
        cp(max(f, key=os.path.getmtime), name)  # (an earlier round's merge may have left older files in the same directory)


cp(os.path.join(ev, "bench.json"), "bench.json")
stats("prof_d4", "bench_kernel_stats.csv")
stats("prof_d1", "bench_depth1_kernel_stats.csv")
stats("prof_open", "open_check_kernel_stats.csv")
stats("fr_trace", "fr_kernels_kernel_stats.csv")
cp(os.path.join(ev, "open_timeline.txt"), "open_check_timeline.txt")
cp(os.path.join(ev, "pmc_traffic.json"), "pmc_traffic.json")
cp(os.path.join(ev, "pmc_fr.json"), "pmc_open.json")
cp(os.path.join(ev, "fr_kernels_events.json"), "fr_kernels_events.json")
cp(os.path.join(ev, "fr29_bench.txt"), "microbench_fr29_product.txt")
cp(os.path.join(ev, "asdl64.json"), "asdl64_n2_20.json")
for n in (2, 4, 8):
    cp(os.path.join(ev, "bench_oneproc%d.json" % n), "rehearsal_oneproc%d_shards_on_1gpu.json" % n)
for f in glob.glob(os.path.join(ev, "bench_gloo*.json")):
    cp(f, "rehearsal_%s_ranks_on_1gpu.json" % os.path.basename(f)[len("bench_"):-len(".json")])
t = os.path.join(ev, "pytest_gpu.txt")
if os.path.exists(t):
    with open(os.path.join(prof, "%s_pytest_gpu_summary.txt" % tag), "w") as o:
        o.write(open(t).read().strip().splitlines()[-1] + "\n")
