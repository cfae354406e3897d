import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import halo_accumulation_amd as h
from halo_accumulation_amd import pcdl
n = 1 << 20; d = n - 1
ctx = h._lib.Context(urs_n=n)
_d = torch.empty((n + 2) * 4, dtype=torch.int64, device="cuda")
ctx.rng_scalars_dev(3, n + 2, _d.data_ptr())
co = np.ascontiguousarray(_d.cpu().numpy().view(np.uint64).reshape(n + 2, 4))
C = pcdl.commit_dev(ctx, _d.data_ptr(), n, d)
for mode in (0, 1):
    ctx.set_fold_table(mode)
    pcdl.open_dev(ctx, [7], _d.data_ptr(), n, C, d, co[n])
    ctx.prof_enable(1); ctx.prof_reset()
    pcdl.open_dev(ctx, [7], _d.data_ptr(), n, C, d, co[n])
    p = ctx.prof(); ctx.prof_enable(0)
    print("mode", mode, {k: (round(v[0], 3), v[1]) for k, v in p.items() if "fold_points" in k})
