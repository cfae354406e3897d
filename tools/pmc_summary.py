"""Per-kernel HBM-side traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass).

    python tools/pmc_summary.py <fetch_dir> <write_dir> > profiles/rNN_pmc_traffic.json

Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3): both counters are in KB;
on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads, so it is doubled (uncalibrated for the
80-byte point gathers of the bucket kernel; Infinity-Cache hits are counted).  Per launch = mean over the
launches of that kernel in the run."""
import csv, glob, json, os, sys, collections


def per_kernel(d, counter):
    f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
    per_dispatch = collections.defaultdict(float)
    name = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        per_dispatch[r["Dispatch_Id"]] += float(r["Counter_Value"])
        name[r["Dispatch_Id"]] = r["Kernel_Name"].split("(")[0].replace("halo::", "")
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for k, v in per_dispatch.items():
        tot[name[k]] += v
        cnt[name[k]] += 1
    return {k: tot[k] / cnt[k] for k in tot}, cnt


fetch, cnt = per_kernel(sys.argv[1], "FETCH_SIZE")
write, _ = per_kernel(sys.argv[2], "WRITE_SIZE")
prog = sys.argv[3] if len(sys.argv) > 3 else "python tools/pipe_loop.py 20 1 6"
out = {"command": "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} -- %s (two separate passes)" % prog,
       "note": "KB units; FETCH_SIZE doubled per the guide's gfx950 correction for 16-B-per-lane loads (uncalibrated for 80-B gathers; "
               "Infinity-Cache hits are counted). n = 2^20.",
       "kernels": {}}
for k in sorted(fetch, key=lambda k: -(2 * fetch[k] + write.get(k, 0.0))):
    out["kernels"][k] = {"launches": cnt[k], "FETCH_SIZE_KB_per_launch": fetch[k], "WRITE_SIZE_KB_per_launch": write.get(k, 0.0),
                         "traffic_bytes_per_launch": (2 * fetch[k] + write.get(k, 0.0)) * 1024}
print(json.dumps(out, indent=1))
