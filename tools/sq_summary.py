"""Per-kernel SQ counters of one rocprofv3 pass -> how busy the vector ALUs were (the bound of the integer kernels).

    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU \
              SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d <dir> -- python tools/pipe_loop.py 20 1 6
    python tools/sq_summary.py <dir> > profiles/rNN_sq_msm.json

Units as /opt/skills/guides/MI355X_MICROARCH.md: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed
over waves, SQ_INSTS_VALU counts wave-instructions, GRBM_GUI_ACTIVE is the sum over the 8 XCDs of busy shader cycles.
Derived, per launch (mean over the launches of a kernel in the run):
  clock_ghz        = GRBM_GUI_ACTIVE / 8 / duration            (the clock the chip held during the kernel)
  valu_per_quad    = SQ_INSTS_VALU / (GRBM_GUI_ACTIVE / 8 / 4 * 1024 SIMDs)   vector instructions issued per SIMD issue slot:
                     1.0 = every 4-cycle slot of every SIMD issued a VALU instruction for the whole kernel
  wave_active/parked/issue_stall = shares of SQ_WAVE_CYCLES (executing / waiting on s_waitcnt or a barrier / ready but the
                     SIMD was issuing another wave's instruction)
"""
import csv, glob, json, os, sys, collections

d = sys.argv[1]
f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
t = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)[0]
short = lambda s: s.split("(")[0].replace("halo::", "")
tot = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = short(r["Kernel_Name"])
    tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
dur = collections.defaultdict(list)
for r in csv.DictReader(open(t)):
    dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = {"command": "rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU "
                  "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- " + (sys.argv[2] if len(sys.argv) > 2 else "python tools/pipe_loop.py 20 1 6"),
       "note": "one MSM in flight (n = 2^20); per launch; durations of the same (counter) pass; 1024 SIMDs", "kernels": {}}
for k in sorted(tot, key=lambda k: -tot[k].get("SQ_INSTS_VALU", 0) / len(disp[k])):
    n = len(disp[k]); c = {a: b / n for a, b in tot[k].items()}
    us = sum(dur[k]) / len(dur[k]) / 1e3
    cyc = c["GRBM_GUI_ACTIVE"] / 8
    wc = max(c["SQ_WAVE_CYCLES"], 1.0)
    out["kernels"][k] = {"launches": n, "duration_us": round(us, 1), "valu_wave_instructions": c["SQ_INSTS_VALU"],
                         "clock_ghz": round(cyc / us / 1e3, 3), "valu_per_quad": round(c["SQ_INSTS_VALU"] / (cyc / 4 * 1024), 3),
                         "wave_active": round(c["SQ_ACTIVE_INST_ANY"] / wc, 3), "wave_parked": round(c["SQ_WAIT_ANY"] / wc, 3),
                         "wave_issue_stall": round(c["SQ_WAIT_INST_ANY"] / wc, 3)}
print(json.dumps(out, indent=1))
