"""Solo MSM latency with and without hipGraph replay, and an L/R-style pair on two slots (development aid)."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import halo_accumulation_amd as h
for lg in (12, 14, 16):
    n = 1 << lg
    ctx = h._lib.Context(urs_n=n)
    d = torch.empty(n * 4, dtype=torch.int64, device="cuda")
    d2 = torch.empty(n * 4, dtype=torch.int64, device="cuda")
    ctx.rng_scalars_dev(2, n, d.data_ptr()); ctx.rng_scalars_dev(3, n, d2.data_ptr())
    for g in (1, 0):
        ctx.lib.halo_set_graphs(ctx.h, g)
        for _ in range(4): ctx.msm_dev(d.data_ptr(), n)
        t = time.perf_counter()
        for _ in range(20): ctx.msm_dev(d.data_ptr(), n)
        solo = (time.perf_counter() - t) / 20 * 1e3
        for _ in range(4):
            ctx.msm_dev_begin(0, d.data_ptr(), n); ctx.msm_dev_begin(1, d2.data_ptr(), n); ctx.msm_dev_end(0); ctx.msm_dev_end(1)
        t = time.perf_counter()
        for _ in range(20):
            ctx.msm_dev_begin(0, d.data_ptr(), n); ctx.msm_dev_begin(1, d2.data_ptr(), n); ctx.msm_dev_end(0); ctx.msm_dev_end(1)
        pair = (time.perf_counter() - t) / 20 * 1e3
        print("lg=%d graphs=%d  solo %.3f ms   pair on two slots %.3f ms" % (lg, g, solo, pair), flush=True)
    ctx.close()
