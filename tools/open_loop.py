"""K x (pcdl::open + check) at n = 2^lg, nothing else (to be run under rocprofv3 --kernel-trace; development aid).
Usage: open_loop.py [lg=20] [K=3] [hiding]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import halo_accumulation_amd as h
from halo_accumulation_amd import pcdl

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
K = int(sys.argv[2]) if len(sys.argv) > 2 else 3
hiding = "hiding" in sys.argv[3:]
DEV = "host" not in sys.argv[3:]  # default: the polynomial is resident in device memory (halo_pcdl_open_dev)
n = 1 << lg; d = n - 1
ctx = h._lib.Context(urs_n=n)
if os.environ.get("REDUCE_SPAN"): ctx.set_reduce_span(int(os.environ["REDUCE_SPAN"]))  # development sweep
if os.environ.get("IPA_SWITCH"): ctx.set_ipa_switch(1 << int(os.environ["IPA_SWITCH"]))  # key size at which the IPA stops folding (log2)
if os.environ.get("FOLD_TABLE"): ctx.set_fold_table(int(os.environ["FOLD_TABLE"]))     # -1 default, 0 never, 1 at the first open
_d = torch.empty((n + 2) * 4, dtype=torch.int64, device="cuda")
ctx.rng_scalars_dev(3, n + 2, _d.data_ptr())
_co = np.ascontiguousarray(_d.cpu().numpy().view(np.uint64).reshape(n + 2, 4))
coeffs, zw = np.ascontiguousarray(_co[:n]), np.ascontiguousarray(_co[n:])
z, w = zw[0], (zw[1] if hiding else None)
C = pcdl.commit(ctx, coeffs, d, w)
v = ctx.poly_eval(coeffs, z)
opens, checks = [], []
for k in range(K):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pi = pcdl.open_dev(ctx, [7], _d.data_ptr(), n, C, d, z, w) if DEV else pcdl.open(ctx, [7], coeffs, C, d, z, w)
    t1 = time.perf_counter()
    pcdl.check_proof(ctx, C, d, z, v, pi)
    t2 = time.perf_counter()
    print("open %.2f ms  check %.2f ms   (fold table: %.1f GB, built in %.1f ms)" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, ctx.info(1) / 1e9, ctx.info(2) / 1e3), flush=True)
    if k >= 2: opens.append(t1 - t0); checks.append(t2 - t1)
if opens:
    print("median of %d: open %.3f ms  check %.3f ms" % (len(opens), sorted(opens)[len(opens) // 2] * 1e3, sorted(checks)[len(checks) // 2] * 1e3), flush=True)
ctx.close()
