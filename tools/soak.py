"""Soak: the same open + check, MSMs (alone and four in flight) and a clone's open on a second thread, over and over on one
context; every result must equal the first of its kind.  Usage: soak.py [seconds=120] [lg=20]  (development aid)"""
import os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import halo_accumulation_amd as h
from halo_accumulation_amd import pcdl

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
lg = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n = 1 << lg; d = n - 1
ctx = h._lib.Context(urs_n=n)
clone = ctx.clone()
buf = torch.empty((n + 2) * 4, dtype=torch.int64, device="cuda")
ctx.rng_scalars_dev(0x50AC, n + 2, buf.data_ptr())
co = np.ascontiguousarray(buf.cpu().numpy().view(np.uint64).reshape(n + 2, 4))
coeffs, zw = np.ascontiguousarray(co[:n]), co[n:]
C = pcdl.commit(ctx, coeffs, d)
Ch = pcdl.commit(ctx, coeffs, d, zw[1])
v = ctx.poly_eval(coeffs, zw[0])
ref = {}
def same(kind, val):
    val = np.asarray(val).tolist()
    if kind not in ref: ref[kind] = val
    assert ref[kind] == val, "result of '%s' changed in iteration %d" % (kind, it)
errors = []
def on_clone():
    try:
        for _ in range(3):
            p = pcdl.open_dev(clone, [7], buf.data_ptr(), n, C, d, zw[0])
            assert np.asarray(p).tolist() == ref["open"], "clone's proof differs"
    except Exception as e:  # noqa
        errors.append(e)
t_end = time.time() + secs
it = 0
while time.time() < t_end:
    p = pcdl.open_dev(ctx, [7], buf.data_ptr(), n, C, d, zw[0]); same("open", p)
    pcdl.check_proof(ctx, C, d, zw[0], v, p)
    same("msm", ctx.msm_dev(buf.data_ptr(), n))
    if it % 5 == 0:
        th = threading.Thread(target=on_clone); th.start()
        ph = pcdl.open(ctx, [9], coeffs, Ch, d, zw[0], zw[1]); same("hiding", ph)
        pcdl.check_proof(ctx, Ch, d, zw[0], v, ph)
        th.join()
        if errors: raise errors[0]
    if it % 7 == 0:
        half = n // 2
        same("msm_half", ctx.msm_dev(buf.data_ptr(), half))
        same("msm_odd", ctx.msm_dev(buf.data_ptr(), half + 4 * 123))
    it += 1
print("soak ok: %d iterations in %.0f s at n = 2^%d" % (it, secs, lg))
clone.close(); ctx.close()
