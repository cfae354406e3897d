"""Per-kernel timing of pcdl::open + check on the GPU box (development aid)."""
import sys, time, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import halo_accumulation_amd as h
from halo_accumulation_amd import pcdl

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
hiding = len(sys.argv) > 2 and sys.argv[2] == "hiding"
n = 1 << lg; d = n - 1
t = time.time(); ctx = h._lib.Context(urs_n=n); print("ctx urs %d: %.2fs" % (n, time.time() - t), flush=True)
import torch
_d = torch.empty((n + 2) * 4, dtype=torch.int64, device="cuda")
ctx.rng_scalars_dev(3, n + 2, _d.data_ptr())  # n coefficients, then z, w from the library's own generator
_co = np.ascontiguousarray(_d.cpu().numpy().view(np.uint64).reshape(n + 2, 4))
coeffs, zw = np.ascontiguousarray(_co[:n]), np.ascontiguousarray(_co[n:])
z, w = zw[0], (zw[1] if hiding else None)
t = time.time(); C = pcdl.commit(ctx, coeffs, d, w); print("commit %.2f ms" % ((time.time() - t) * 1e3))
pi = pcdl.open(ctx, [7], coeffs, C, d, z, w)  # warm-up
ctx.prof_enable(True); ctx.prof_reset()
t = time.time(); pi = pcdl.open(ctx, [7], coeffs, C, d, z, w); t_open = time.time() - t
print("open (profiled) %.2f ms" % (t_open * 1e3))
for k, (ms, cnt) in sorted(ctx.prof().items(), key=lambda kv: -kv[1][0]):
    print("   %-22s total %9.3f ms  launches %4d  avg %8.3f ms" % (k, ms, cnt, ms / max(cnt, 1)))
ctx.prof_enable(False)
t = time.time(); pi = pcdl.open(ctx, [7], coeffs, C, d, z, w); print("open (unprofiled) %.2f ms" % ((time.time() - t) * 1e3))
for sw in (1 << 11, 1 << 12, 1 << 13, 1 << 14, 1 << 15, 1 << 16, 1 << 17):
    ctx.set_ipa_switch(sw)
    pcdl.open(ctx, [7], coeffs, C, d, z, w)
    t = time.time(); pi2 = pcdl.open(ctx, [7], coeffs, C, d, z, w)
    print("switch 2^%d: open %.2f ms  same proof: %s" % (sw.bit_length() - 1, (time.time() - t) * 1e3, pi2.tolist() == pi.tolist()), flush=True)
ctx.set_ipa_switch(1 << 16)
v = ctx.poly_eval(coeffs, z)
t = time.time(); pcdl.succinct_check(ctx, C, d, z, v, pi); print("succinct_check %.2f ms" % ((time.time() - t) * 1e3))
t = time.time(); pcdl.check_proof(ctx, C, d, z, v, pi); print("check %.2f ms" % ((time.time() - t) * 1e3), flush=True)
