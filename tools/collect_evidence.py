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
cp(os.path.join(ev, "pmc_traffic_general.json"), "pmc_traffic_general.json")
stats("prof_d1_gen", "general_pipeline_depth1_kernel_stats.csv")
cp(os.path.join(ev, "bench_2_24.json"), "bench_2_24_one_gpu.json")
cp(os.path.join(ev, "bench_oneproc8_2_24.json"), "rehearsal_oneproc8_2_24_shards_on_1gpu.json")
cp(os.path.join(ev, "sq_msm.json"), "sq_msm.json")
cp(os.path.join(ev, "sq_open.json"), "sq_open.json")
cp(os.path.join(ev, "host_msm.txt"), "host_msm_pageable.txt")
cp(os.path.join(ev, "pmc_fr.json"), "pmc_open.json")
cp(os.path.join(ev, "fr_kernels_events.json"), "fr_kernels_events.json")
cp(os.path.join(ev, "pmc_open_kernels.json"), "pmc_open_loop.json")
cp(os.path.join(ev, "fr29_bench.txt"), "microbench_fr29_product.txt")
cp(os.path.join(ev, "asdl64.json"), "asdl64_n2_20.json")
for n in (2, 4, 8):
    cp(os.path.join(ev, "bench_oneproc%d.json" % n), "rehearsal_oneproc%d_shards_on_1gpu.json" % n)
cp(os.path.join(ev, "bench_rccl1.json"), "rehearsal_rccl_one_rank_collective_path.json")
for f in glob.glob(os.path.join(ev, "bench_gloo*.json")):
    cp(f, "rehearsal_%s_ranks_on_1gpu.json" % os.path.basename(f)[len("bench_"):-len(".json")])
t = os.path.join(ev, "pytest_gpu.txt")
if os.path.exists(t):
    with open(os.path.join(prof, "%s_pytest_gpu_summary.txt" % tag), "w") as o:
        o.write(open(t).read().strip().splitlines()[-1] + "\n")

# r03_pmc_open.json: join the PMC bytes of the Fr kernels with their rocprofv3 durations (tools/fr_kernels.py, launched back
# to back) and their algorithmic bytes -> GB/s per kernel, run alone, n = 2^20
try:
    import csv
    pj = os.path.join(prof, "%s_pmc_open.json" % tag)
    st = os.path.join(prof, "%s_fr_kernels_kernel_stats.csv" % tag)
    ev_json = os.path.join(prof, "%s_fr_kernels_events.json" % tag)
    if os.path.exists(pj) and os.path.exists(st) and os.path.exists(ev_json):
        d = json.load(open(pj))
        dur = {r["Name"].split("(")[0].replace("halo::", "").replace("void ", ""): r for r in csv.DictReader(open(st))}
        alg = {}
        for k in json.load(open(ev_json))["kernels"]:
            alg.setdefault(k["kernel"], []).append(k["algorithmic_bytes"])
        out = {}
        for name, v in d["kernels"].items():
            base = name.replace("void ", "").split("<")[0]
            if base not in alg:
                continue
            r = dur.get(name.replace("void ", ""))
            if not r:
                continue
            us = float(r["AverageNs"]) / 1e3
            a = alg[base][0] if len(alg[base]) == 1 or "<false>" in name or "<" not in name else alg[base][-1]
            v = dict(v)
            v.update({"rocprofv3_avg_us": us, "rocprofv3_min_us": float(r["MinNs"]) / 1e3, "rocprofv3_max_us": float(r["MaxNs"]) / 1e3,
                      "algorithmic_bytes": a, "GBps_algorithmic": a / us / 1e3, "GBps_hbm_side": v["traffic_bytes_per_launch"] / us / 1e3,
                      "frac_of_8TBs": a / us / 1e3 / 8000.0})
            out[name] = v
        d["kernels"] = out
        d["note"] += " Durations: rocprofv3 --kernel-trace --stats of the same program with 20 launches per kernel (%s_fr_kernels_kernel_stats.csv); " \
                     "algorithmic bytes as in DESIGN.md 4.4." % tag
        json.dump(d, open(pj, "w"), indent=1)
        print("   joined", os.path.basename(pj))
except Exception as e:  # the summaries above are already in place
    print("   (pmc_open join skipped: %s)" % e)
