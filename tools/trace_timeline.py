"""Timeline of the last `span_ms` of a rocprofv3 --kernel-trace CSV: every kernel with its start (relative), duration and the
idle gap since the previous kernel ended (all streams merged).  Usage: trace_timeline.py trace.csv [marker_kernel] [count]
The window starts at the `count`-th last launch of marker_kernel (default: the last k_poly_eval_partial = start of an open)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "k_poly_eval_partial"
count = int(sys.argv[3]) if len(sys.argv) > 3 else 1
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('halo::', '').replace('void ', ''), r.get('Stream_Id', r.get('Queue_Id', '?'))) for r in rows)
marks = [e for e in ev if e[2].startswith(marker)]
t0 = marks[-count][0]
sel = [e for e in ev if e[0] >= t0]
last_end = t0
busy = 0
agg = collections.OrderedDict()
for s, e, nm, q in sel:
    gap = s - last_end
    print("%10.1f us  %-26s %9.1f us  gap %8.1f us  q=%s" % ((s - t0) / 1e3, nm[:26], (e - s) / 1e3, gap / 1e3, q))
    if e > last_end:
        busy += e - max(s, last_end)
        last_end = e
    a = agg.setdefault(nm, [0, 0]); a[0] += 1; a[1] += e - s
print("window %.3f ms, busy (union) %.3f ms" % ((last_end - t0) / 1e6, busy / 1e6))
for nm, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("  %-28s x%4d total %9.1f us" % (nm[:28], c, t / 1e3))
