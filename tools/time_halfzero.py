"""Table MSM with part of the scalars zero (the L / R MSMs of the first IPA rounds): per-kernel times."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import halo_accumulation_amd as h
import torch
lg = 20; n = 1 << lg
ctx = h._lib.Context(urs_n=n)
for frac, pattern in ((1.0, "all"), (0.5, "halves of 2^19"), (0.5, "alternate 2^10 runs"), (0.25, "quarter")):
    d = torch.empty(n * 4, dtype=torch.int64, device="cuda")
    ctx.rng_scalars_dev(2, n, d.data_ptr())
    v = d.view(n, 4)
    if pattern == "halves of 2^19": v[n // 2:] = 0
    elif pattern == "alternate 2^10 runs": v.view(n >> 11, 2, 1 << 10, 4)[:, 1] = 0
    elif pattern == "quarter": v[n // 4:] = 0
    torch.cuda.synchronize()
    ctx.msm_dev(d.data_ptr(), n)
    ctx.prof_enable(True); ctx.prof_reset()
    K = 5
    for _ in range(K): ctx.msm_dev(d.data_ptr(), n)
    pr = ctx.prof(); ctx.prof_enable(False)
    t = time.time()
    for _ in range(K): ctx.msm_dev(d.data_ptr(), n)
    dt = (time.time() - t) / K
    print("%-22s %.3f ms per MSM;  " % (pattern, dt * 1e3) + "  ".join("%s %.0f" % (k.replace("k_msm_", "").replace("k_tmsm_", ""), ms / cnt * 1e3) for k, (ms, cnt) in sorted(pr.items(), key=lambda kv: -kv[1][0])[:6]), flush=True)
