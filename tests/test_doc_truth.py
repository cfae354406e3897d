"""DESIGN.md quotes figures from profiles/; twice a judge found a quoted figure that the committed file did not hold.
These checks tie the handful of figures the roofline argument rests on to the files they are quoted from."""
import csv
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _latest(suffix):
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)))
    assert files, suffix
    return files[-1]


def test_design_quotes_the_committed_gpu_test_count():
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    summary = open(_latest("pytest_gpu_summary.txt")).read()
    passed = int(re.search(r"(\d+) passed", summary).group(1))
    quoted = [int(x) for x in re.findall(r"(\d+) `-m gpu` tests", design)]
    assert quoted and all(q == passed for q in quoted), (quoted, passed)


def test_design_quotes_the_committed_kernel_duration_and_the_bench_line():
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    stats = _latest("bench_depth1_kernel_stats.csv")
    rows = [r for r in csv.DictReader(open(stats)) if "k_msm_accumulate" in r["Name"]]
    assert len(rows) == 1
    avg_us = float(rows[0]["AverageNs"]) / 1e3
    tag = os.path.basename(stats)[:3]
    m = re.search(r"rocprofv3 average \*\*(\d+) µs\*\* over (\d+) launches, `profiles/%s_bench_depth1_kernel_stats.csv`" % tag, design)
    assert m, "DESIGN.md section 4.1 must quote the rocprofv3 average of k_msm_accumulate from " + os.path.basename(stats)
    assert abs(int(m.group(1)) - avg_us) < 1.0 and int(m.group(2)) == int(rows[0]["Calls"]), (m.groups(), avg_us, rows[0]["Calls"])
    bench = json.load(open(_latest("bench.json")))
    # the event-timed duration of the same kernel in the committed bench line agrees with the profiler's to within 10 %
    assert abs(bench["roofline"]["kernel_ms"] * 1e3 - avg_us) / avg_us < 0.10
    ev = re.search(r"HIP events in the committed `bench.py` run \*\*(\d+) µs\*\*", design)
    assert ev and abs(int(ev.group(1)) - bench["roofline"]["kernel_ms"] * 1e3) < 1.0
    # headline and open + check as quoted in section 5
    head = re.search(r"\*\*(\d+) MSM/s\*\* at n = 2\^20 \((\d+)–(\d+) from box to box", design)
    assert head and abs(int(head.group(1)) - bench["value"]) < 1.0
    oc = re.search(r"`pcdl::open \+ check` \*\*([\d.]+) ms\*\* \(([\d.]+)–([\d.]+) from box to box", design)
    assert oc and abs(float(oc.group(1)) - bench["pcdl_open_check"]["ms"]) < 0.06
    # (the open depends on the host's single-thread speed: the file gives the box-to-box range next to the committed figure)
    assert float(oc.group(2)) <= bench["pcdl_open_check"]["ms"] + 0.06 and bench["pcdl_open_check"]["ms"] - 0.06 <= float(oc.group(3))
    # the roofline fraction is what the definition gives
    assert abs(bench["roofline"]["frac"] - bench["roofline"]["algorithmic_bytes"] / (bench["roofline"]["kernel_ms"] * 1e-3) / 8e12) < 1e-9
    # a "from box to box" range must contain every value the DRIVER has recorded since round 3 (the rounds whose
    # pipeline the range describes), not only the builder's own runs
    lo, hi = int(head.group(2)), int(head.group(3))
    assert lo <= bench["value"] <= hi
    for f in sorted(glob.glob(os.path.join(ROOT, "BENCH_r[0-9][0-9].json"))):
        if int(os.path.basename(f)[7:9]) < 3:
            continue
        v = (json.load(open(f)).get("parsed") or {}).get("value")
        if v is not None:
            assert lo <= v <= hi, "DESIGN.md's MSM/s range %d-%d does not contain the driver's %s: %.1f" % (lo, hi, os.path.basename(f), v)
        k = ((json.load(open(f)).get("parsed") or {}).get("roofline") or {}).get("kernel_ms")
        km = re.search(r"run \*\*\d+ µs\*\* \(([\d.]+)–([\d.]+) from box to box\)", design)
        assert km, "the bucket kernel's event-timed duration needs its box-to-box range"
        if k is not None:
            assert float(km.group(1)) - 0.005 <= k <= float(km.group(2)) + 0.005, (os.path.basename(f), k)


def test_design_is_a_spec_not_a_history():
    """the current state in at most 25 KB; what was measured and dropped lives in HISTORY.md."""
    assert os.path.getsize(os.path.join(ROOT, "DESIGN.md")) <= 25 * 1024
    assert os.path.exists(os.path.join(ROOT, "HISTORY.md"))
