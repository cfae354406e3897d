"""halo_msm with the scalars in PAGEABLE host memory (what integration/ffi.rs point_dot_affine does): latency of one call at
n = 2^lg, median of K, result compared with halo_msm_dev every time.  HALO_HOST_SPLIT="4,12" (csrc/tuning.hpp): the
stretches, in sixteenths of the points, whose copies run under the other stretches' kernels; "16" = one copy, one launch sequence.
Usage: host_msm.py LG [K=30]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import halo_accumulation_amd as h
lg = int(sys.argv[1]); K = int(sys.argv[2]) if len(sys.argv) > 2 else 30
n = 1 << lg
ctx = h._lib.Context(urs_n=n)
d = torch.empty(n * 4, dtype=torch.int64, device="cuda")
ctx.rng_scalars_dev(2, n, d.data_ptr())
want = ctx.msm_dev(d.data_ptr(), n)
sc = np.ascontiguousarray(d.cpu().numpy().view(np.uint64).reshape(n, 4)).copy()  # plain numpy memory: pageable
for _ in range(6):
    assert ctx.msm(sc).tolist() == want.tolist()
ts = []
for _ in range(K):
    t0 = time.perf_counter(); r = ctx.msm(sc); ts.append(time.perf_counter() - t0)
    assert r.tolist() == want.tolist()
ts2 = []
for _ in range(K):
    t0 = time.perf_counter(); r = ctx.msm_dev(d.data_ptr(), n); ts2.append(time.perf_counter() - t0)
med = lambda v: sorted(v)[len(v) // 2]
print("lg=%d HALO_HOST_SPLIT=%s halo_msm (pageable host scalars) %.3f ms (min %.3f)   halo_msm_dev %.3f ms" % (
    lg, os.environ.get("HALO_HOST_SPLIT", "default"), med(ts) * 1e3, min(ts) * 1e3, med(ts2) * 1e3))
ctx.close()
