"""Per-kernel SQ counters of one rocprofv3 pass -> how busy the vector ALUs were (the bound of the integer kernels).

    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU \
              SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d <dir> -- python tools/pipe_loop.py 20 1 6
    python tools/sq_summary.py <dir> "python tools/pipe_loop.py 20 1 6" "one MSM in flight (n = 2^20)" > profiles/rNN_sq_msm.json

The profiled program's command line (second argument) is REQUIRED and goes into the file's "command" as given; the third
argument says what the program had in flight and goes into "note".  (Round 4's r04_sq_open.json carried the MSM loop's command
and note over the open's kernels: both used to default to the MSM loop here.)

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

if len(sys.argv) < 3:
    sys.exit("usage: sq_summary.py <rocprofv3 output dir> \"<profiled command>\" [\"<what was in flight>\"]")
d, program = sys.argv[1], sys.argv[2]
what = sys.argv[3] if len(sys.argv) > 3 else "see command"
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
                  "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- " + program,
       "note": what + "; per launch; durations of the same (counter) pass; 1024 SIMDs; kernel names are the symbols rocprofv3 prints "
               "(the event profiler's label k_fold_points4_tab is the symbol k_fold_tab4)", "kernels": {}}
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
