"""A few unpipelined 2^LG MSMs (for builds with device-side phase timers)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import halo_accumulation_amd as h
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 20)
ctx = h._lib.Context(urs_n=n)
d = torch.empty(n * 4, dtype=torch.int64, device="cuda")
ctx.rng_scalars_dev(2, n, d.data_ptr())
for _ in range(3): ctx.msm_dev(d.data_ptr(), n)
torch.cuda.synchronize()
