import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import halo_accumulation_amd as h
import orc
ctx = h._lib.Context(urs_n=4096)
gs = ctx.read_bases()
for n, vals in [(1, [1]), (1, [2]), (1, [3]), (1, [4]), (1, [7]), (1, [8]), (1, [9]), (1, [17]), (1, [2**200 + 5]), (2, [1, 2]), (2, [2, 2]), (2, [3, 5]), (3, [1, 2, 3]), (3, [0, 1, orc.fr_from_mont(orc.fr_to_mont(0)) - 1 + 28948022309329048855892746252171976963363056481941647379679742748393362948097])]:
    sc = np.ascontiguousarray(np.stack([orc.fr_to_mont(v % 28948022309329048855892746252171976963363056481941647379679742748393362948097) for v in vals]))
    want = orc.msm_affine(gs[:n], sc).tolist()
    res = []
    for c in (0, 4, 5, 8):
        ctx.set_window_bits(c)
        ctx.set_small_path(-1); a = ctx.msm(sc).tolist() == want
        ctx.set_small_path(0); b = ctx.msm(sc).tolist() == want
        res.append((c, a, b))
    print(n, [v if v < 1000 else "big" for v in vals], res, flush=True)
