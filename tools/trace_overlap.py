"""Summarise a rocprofv3 --kernel-trace CSV of tools/pipe_loop.py: concurrency histogram and per-kernel time per MSM."""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
K = int(sys.argv[2]) if len(sys.argv) > 2 else 64
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('halo::', '')) for r in rows)
acc = [e for e in ev if e[2] == 'k_msm_accumulate']
t0 = acc[-K][0] - 300000
sel = [e for e in ev if e[0] >= t0]
tend = max(e[1] for e in sel); tbeg = min(e[0] for e in sel)
print('window %.3f ms, %.3f ms per MSM' % ((tend - tbeg) / 1e6, (tend - tbeg) / 1e6 / K))
pts = []
for s, e, _ in sel:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
cur = 0; last = tbeg; hist = collections.Counter()
for t, dv in pts:
    hist[cur] += t - last; last = t; cur += dv
for k in sorted(hist):
    print('  %d kernels running: %.1f%%' % (k, 100 * hist[k] / (tend - tbeg)))
d = collections.defaultdict(list)
for s, e, nm in sel:
    d[nm].append(e - s)
tot = 0
for nm, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print('  %-24s x%4d avg %8.1f us   per MSM %8.1f us' % (nm, len(v), sum(v) / len(v) / 1e3, sum(v) / K / 1e3)); tot += sum(v)
print('  sum of kernel time per MSM %.1f us' % (tot / K / 1e3))
# time with / without an accumulate kernel in flight, and what runs while there is none
pts = []
for s, e, nm in sel:
    pts.append((s, 1, nm)); pts.append((e, -1, nm))
pts.sort()
live = collections.Counter(); last = tbeg; with_acc = 0; without = collections.Counter(); n_acc = collections.Counter()
for t, dv, nm in pts:
    dt = t - last; last = t
    if dt > 0:
        n_acc[live['k_msm_accumulate']] += dt
        if live['k_msm_accumulate'] > 0: with_acc += dt
        else: without[tuple(sorted(k for k, v in live.items() if v > 0))] += dt
    live[nm] += dv
span = tend - tbeg
print('  accumulate in flight %.1f%% of the window' % (100 * with_acc / span))
for k in sorted(n_acc): print('    %d accumulate kernels: %.1f%%' % (k, 100 * n_acc[k] / span))
for k, v in sorted(without.items(), key=lambda kv: -kv[1])[:10]:
    print('    no accumulate, running %-60s %.1f%%  (%.1f us per MSM)' % ('+'.join(x.replace('k_msm_', '').replace('k_tmsm_', 't:') for x in k) or 'idle', 100 * v / span, v / K / 1e3))
