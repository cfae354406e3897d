"""ShardedOpen with ONE rank against pcdl::open on the same GPU (development aid): what the per-round Python/partial path costs
when no collective is involved.  Usage: sharded_open_loop.py [lg=20] [K=6]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import halo_accumulation_amd as h
from halo_accumulation_amd import pcdl
from halo_accumulation_amd.sharded import ShardedOpen

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
n = 1 << lg
so = ShardedOpen(h._lib, 0, 1, None)
ctx = so.load_key(n)
d = torch.empty((n + 2) * 4, dtype=torch.int64, device="cuda")
ctx.rng_scalars_dev(3, n + 2, d.data_ptr())
co = np.ascontiguousarray(d.cpu().numpy().view(np.uint64).reshape(n + 2, 4))
coeffs, z = np.ascontiguousarray(co[:n]), co[n]
C = pcdl.commit(ctx, coeffs, n - 1)
for k in range(K):
    t0 = time.perf_counter(); p1, v = so.open(coeffs, C, z); t1 = time.perf_counter()
    so.check(C, n - 1, z, v, p1); t2 = time.perf_counter()
    p2 = pcdl.open(ctx, [1], coeffs, C, n - 1, z); t3 = time.perf_counter()
    pcdl.check_proof(ctx, C, n - 1, z, v, p2); t4 = time.perf_counter()
    assert p1.tolist() == p2.tolist()
    print("sharded(P=1): open %.2f check %.2f ms   plain: open %.2f check %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3), flush=True)
