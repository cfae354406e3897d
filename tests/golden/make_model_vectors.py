#!/usr/bin/env python3
"""Generate tests/golden/model_vectors.json from the Python big-int model (oracle/pallas_model.py).

    python tests/golden/make_model_vectors.py

Inputs are seeded (SplitMix64); bases are the URS points G_i of main.rs:18-45
(the first 16,384 are the reference's consts.rs table, see urs_kat.json).  Values are canonical
(non-Montgomery) hex.  The vectors pin, independently of any C or HIP code: MSM results incl. edge
scalars, one IPA fold round, h(X) coefficients/evaluation and one complete non-hiding
open + check transcript at n = 8 (with this repository's rendering of the ark-serialize encoding)."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import pallas_model as pm  # noqa: E402

hx = lambda v: "%064x" % v
pt = lambda p: None if p is None else [hx(p[0]), hx(p[1])]


def main():
    S, H, G = pm.get_pp(256)
    out = {"generator": "tests/golden/make_model_vectors.py", "scalar_seed_base": "0x48414C4F00000002 + n"}
    edge = [0, 1, pm.R_ORDER - 1, 1 << 254, 2, pm.R_ORDER - 2]
    msm = []
    for n in (1, 2, 3, 31, 32, 33, 256):
        rng = pm.SplitMix64(0x48414C4F00000002 + n)
        xs = [rng.next_scalar() for _ in range(n)]
        for i, e in enumerate(edge[: min(n, len(edge))]):
            xs[(i * 7) % n] = e % pm.R_ORDER
        msm.append({"n": n, "scalars": [hx(x) for x in xs], "result": pt(pm.point_dot(xs, G[:n]))})
    out["msm"] = msm
    # one fold round, m = 4 (pcdl.rs:216-224)
    rng = pm.SplitMix64(0xF01D)
    cs = [rng.next_scalar() for _ in range(8)]
    z = rng.next_scalar()
    xi = rng.next_scalar()
    zs = pm.construct_powers(z, 8)
    xi_inv = pm.inv_mod(xi, pm.R_ORDER)
    out["fold"] = {"m": 4, "c": [hx(c) for c in cs], "z": hx(z), "xi": hx(xi),
                   "G_out": [pt(pm.add(G[j], pm.mul(G[j + 4], xi))) for j in range(4)],
                   "c_out": [hx((cs[j] + cs[j + 4] * xi_inv) % pm.R_ORDER) for j in range(4)],
                   "z_out": [hx((zs[j] + zs[j + 4] * xi) % pm.R_ORDER) for j in range(4)],
                   "L": pt(pm.add(pm.point_dot(cs[4:], G[:4]), pm.mul(H, pm.scalar_dot(cs[4:], zs[:4])))),
                   "R": pt(pm.add(pm.point_dot(cs[:4], G[4:8]), pm.mul(H, pm.scalar_dot(cs[:4], zs[4:]))))}
    # h(X)
    hv = []
    for lg in (2, 3, 10):
        rng = pm.SplitMix64(0x4800 + lg)
        xis = [rng.next_scalar() for _ in range(lg + 1)]
        zz = rng.next_scalar()
        co = pm.h_coeffs(xis)
        hv.append({"lg_n": lg, "xis": [hx(x) for x in xis], "z": hx(zz), "eval": hx(pm.h_eval(xis, zz)),
                   "coeffs_head": [hx(c) for c in co[:8]], "coeffs_tail": [hx(c) for c in co[-2:]],
                   "commit": pt(pm.point_dot(co, G[: 1 << lg])) if lg <= 3 else None})
    out["h"] = hv
    # complete non-hiding open + check at n = 8
    pp = pm.PublicParams(S, H, G[:8])
    rng = pm.SplitMix64(0x0BE1)
    coeffs = [rng.next_scalar() for _ in range(6)]
    zz = rng.next_scalar()
    C = pm.pcdl_commit(pp, coeffs, 7, None)
    pi = pm.pcdl_open(pp, coeffs, C, 7, zz, None)
    v = pm.poly_eval(coeffs, zz)
    pm.pcdl_check(pp, C, 7, zz, v, pi)
    xis, _ = pm.pcdl_succinct_check(pp, C, 7, zz, v, pi)
    out["open_n8"] = {"coeffs": [hx(c) for c in coeffs], "z": hx(zz), "v": hx(v), "C": pt(C),
                      "Ls": [pt(p) for p in pi["Ls"]], "Rs": [pt(p) for p in pi["Rs"]], "U": pt(pi["U"]), "c": hx(pi["c"]),
                      "xis": [hx(x) for x in xis], "note": "transcript bytes follow this repo's ark-serialize rendering (parity unpinned)"}
    with open(os.path.join(HERE, "model_vectors.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote model_vectors.json")


if __name__ == "__main__":
    main()
