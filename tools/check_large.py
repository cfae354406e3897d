"""Size-independent property at large n: msm over [0, n) == msm over [0, n/2) + msm over [n/2, n)."""
import sys, time, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import halo_accumulation_amd as h
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
n = 1 << lg
t = time.time(); ctx = h._lib.Context(urs_n=n); print("ctx 2^%d: %.1fs" % (lg, time.time() - t), flush=True)
d = torch.empty(n * 4, dtype=torch.int64, device="cuda")
ctx.rng_scalars_dev(0x48414C4F00000005, n, d.data_ptr())
full = ctx.msm_dev(d.data_ptr(), n)
t = time.time(); full = ctx.msm_dev(d.data_ptr(), n); dt = time.time() - t
lo = ctx.msm_dev(d.data_ptr(), n // 2)
hi = ctx.msm_dev(d.data_ptr() + (n // 2) * 32, n // 2, off=n // 2)
print("n=2^%d: %.2f ms per MSM; split linearity: %s" % (lg, dt * 1e3, h._lib.point_sum(np.stack([lo, hi])).tolist() == full.tolist()), flush=True)
