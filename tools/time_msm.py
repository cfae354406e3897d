"""Quick per-kernel timing of the MSM pipeline on the GPU box (development aid)."""
import sys, time, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import halo_accumulation_amd as h
import torch

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << lg
t = time.time(); ctx = h._lib.Context(urs_n=n); print("ctx urs %d: %.2fs" % (n, time.time() - t), flush=True)
d = torch.empty(n * 4, dtype=torch.int64, device="cuda")
ctx.rng_scalars_dev(2, n, d.data_ptr())  # the library's own SplitMix64 generator
if len(sys.argv) > 3: ctx.set_small_path(int(sys.argv[3]))  # 0: general pipeline at every size
for c in ([0] if len(sys.argv) < 3 else [int(x) for x in sys.argv[2].split(",")]):
    ctx.set_window_bits(c)
    ctx.msm_dev(d.data_ptr(), n)
    ctx.prof_enable(True); ctx.prof_reset()
    t = time.time(); K = 5
    for _ in range(K): out = ctx.msm_dev(d.data_ptr(), n)
    dt = (time.time() - t) / K
    print("c=%d  n=2^%d  %.3f ms per MSM (profiled)" % (c, lg, dt * 1e3))
    for k, (ms, cnt) in sorted(ctx.prof().items(), key=lambda kv: -kv[1][0]):
        print("   %-20s %8.3f ms  x%d" % (k, ms / max(cnt, 1), cnt // K))
    ctx.prof_enable(False)
    t = time.time()
    for _ in range(K): out = ctx.msm_dev(d.data_ptr(), n)
    print("   unprofiled: %.3f ms per MSM" % ((time.time() - t) / K * 1e3), flush=True)
    for depth in (2, 4):
        K2 = 40; t = time.time(); pend = []
        for i in range(K2):
            if len(pend) == depth: out2 = ctx.msm_dev_end(pend.pop(0))
            ctx.msm_dev_begin(i % depth, d.data_ptr(), n); pend.append(i % depth)
        while pend: out2 = ctx.msm_dev_end(pend.pop(0))
        print("   pipelined depth %d: %.3f ms per MSM   same result: %s" % (depth, (time.time() - t) / K2 * 1e3, out2.tolist() == out.tolist()), flush=True)
