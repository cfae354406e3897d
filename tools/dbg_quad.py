import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import halo_accumulation_amd as h
import orc
ctx = h._lib.Context(urs_n=4096)
gs = ctx.read_bases()
n = 300
a = np.zeros((n, 12), dtype=np.uint64); b = np.zeros((n, 12), dtype=np.uint64)
for i in range(n):
    orc.lib().orc_affine_to_jac(orc.ptr(gs[i]), orc.ptr(a[i])); orc.lib().orc_affine_to_jac(orc.ptr(gs[i + n]), orc.ptr(b[i]))
b[5] = a[5]
inf = np.array(list(a[2][:8]) + [0, 0, 0, 0], dtype=np.uint64)
a[9] = inf; b[10] = inf
for op in (0, 4, 5, 6):
    got = ctx.point_op(op, a, b)
    bad = []
    for i in range(n):
        bb = a[i] if (op == 5 or (op == 6 and i % 4 == 3)) else b[i]
        want = orc.z(12); orc.lib().orc_point_add(orc.ptr(a[i]), orc.ptr(bb), orc.ptr(want))
        if orc.point_canonical(got[i]) != orc.point_canonical(want): bad.append(i)
    print("op", op, "bad", len(bad), bad[:20], flush=True)
