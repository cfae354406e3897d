"""Event-bracketed durations of the fold kernels of one pcdl::open at n = 2^lg (development aid).
Usage: fold_times.py [lg=20]   (HALO_FOLD_SPLIT=0/1, HALO_FOLD_SPLIT_PCT select the kernels)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import halo_accumulation_amd as h
from halo_accumulation_amd import pcdl

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << lg; d = n - 1
ctx = h._lib.Context(urs_n=n)
ctx.set_fold_table(int(os.environ.get("FOLD_TABLE", "1")))
_d = torch.empty((n + 2) * 4, dtype=torch.int64, device="cuda")
ctx.rng_scalars_dev(3, n + 2, _d.data_ptr())
z = np.ascontiguousarray(_d[4 * n: 4 * n + 4].cpu().numpy().view(np.uint64))
C = pcdl.commit_dev(ctx, _d.data_ptr(), n, d)
for _ in range(3):
    pcdl.open_dev(ctx, [7], _d.data_ptr(), n, C, d, z)
ctx.prof_enable(1); ctx.prof_reset()
for _ in range(3):
    pcdl.open_dev(ctx, [7], _d.data_ptr(), n, C, d, z)
for k, (ms, cnt) in sorted(ctx.prof().items(), key=lambda kv: -kv[1][0])[:12]:
    print("%-26s %8.3f ms per open  (%d launches per open, %.1f us each)" % (k, ms / 3, cnt // 3, ms / max(cnt, 1) * 1e3))
ctx.close()
