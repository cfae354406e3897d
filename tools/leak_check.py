"""Device-memory leak check (development aid): free HBM before/after many create/open/check/destroy cycles."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import halo_accumulation_amd as h
from halo_accumulation_amd import pcdl, acc as A
def free_mb():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0] / 2**20
n = 1 << 14
d = torch.empty((n + 2) * 4, dtype=torch.int64, device="cuda")
base = None
for rep in range(6):
    ctx = h._lib.Context(urs_n=n)
    ctx.rng_scalars_dev(3 + rep, n + 2, d.data_ptr())
    co = np.ascontiguousarray(d.cpu().numpy().view(np.uint64).reshape(n + 2, 4))
    coeffs, z, w = np.ascontiguousarray(co[:n]), co[n], co[n + 1]
    for k in range(25):
        C = pcdl.commit(ctx, coeffs, n - 1, w)
        pi = pcdl.open(ctx, [k], coeffs, C, n - 1, z, w)
        pcdl.check_proof(ctx, C, n - 1, z, ctx.poly_eval(coeffs, z), pi)
    ptrs = [d.data_ptr()] * 4
    for k in range(10):
        ctx.msm_dev_batch_begin(k % 4, ptrs, n, part=k % 3, parts=3); ctx.msm_dev_batch_end(k % 4, 4)
    rng = [5]
    q = A.random_instance(ctx, rng, n - 1)
    a = A.prover(ctx, rng, n - 1, [q]); A.verifier(ctx, n - 1, [q], a); A.decider(ctx, a)
    ctx.close()
    f = free_mb()
    if base is None: base = f
    print("cycle %d: free HBM %.1f MiB (delta vs first cycle %.1f MiB)" % (rep, f, f - base), flush=True)
assert abs(f - base) < 64, "device memory leak"
print("no leak")
