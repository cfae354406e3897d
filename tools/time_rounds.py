"""Per-round wall time of the IPA loop through the C ABI (development aid): round_lr_partial / combine (host) / round_fold."""
import sys, time, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import halo_accumulation_amd as h
L = h._lib
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << lg
ctx = L.Context(urs_n=n)
d = torch.empty((n + 1) * 4, dtype=torch.int64, device="cuda")
ctx.rng_scalars_dev(3, n + 1, d.data_ptr())
co = np.ascontiguousarray(d.cpu().numpy().view(np.uint64).reshape(n + 1, 4))
coeffs, z = np.ascontiguousarray(co[:n]), co[n]
S, H = L.public_points()
for rep in range(2):
    ipa = L.Ipa(ctx, n, coeffs, z)
    xi = z.copy()
    rows = []
    torch.cuda.synchronize()
    t_all = time.perf_counter()
    for r in range(lg):
        t0 = time.perf_counter(); rec = ipa.round_lr_partial()
        t1 = time.perf_counter(); Lp, Rp, xi, xi_inv = L.open_combine(rec[None], H, xi)
        t2 = time.perf_counter(); ipa.round_fold(xi, xi_inv)
        t3 = time.perf_counter()
        rows.append((r, len(ipa), (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
    tot = (time.perf_counter() - t_all) * 1e3
    ipa.finish(); ipa.close()
for r, m, a, b, c in rows:
    print("round %2d  m after=%8d  lr %.3f ms  combine(host) %.3f ms  fold %.3f ms" % (r, m, a, b, c))
print("total %.2f ms: lr %.2f, combine %.2f, fold %.2f" % (tot, sum(x[2] for x in rows), sum(x[3] for x in rows), sum(x[4] for x in rows)))
