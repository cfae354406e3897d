#!/usr/bin/env python3
"""Generate tests/golden/urs_kat.json from the reference's public-parameter table.

Run in the build container only (needs /root/reference, which never travels):

    python tests/golden/make_urs_kat.py

It reads the *data* in code/src/consts.rs (S, H and the 16,384 GS entries:
Montgomery-form u64x4 limbs), decodes them to canonical coordinates and commits

* S, H, GS[0..64) and GS[16383] as canonical big-endian hex (a small KAT subset),
* one SHA-256 digest over the raw little-endian limb bytes of the WHOLE GS table
  (x limbs then y limbs, 64 bytes per point) and one over S||H Jacobian limbs,

so that the C restatement and the HIP URS kernel can be pinned on all 16,386
reference points without the 1.7 MB table itself entering this repository.
"""
import hashlib
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import pallas_model as pm  # noqa: E402

SRC = "/root/reference/code/src/consts.rs"


def main() -> None:
    text = open(SRC).read()
    aff = re.findall(r"mk_aff!\(\[([0-9, ]+)\], \[([0-9, ]+)\]\)", text)
    assert len(aff) == 16384, len(aff)
    projs = re.findall(r"pub const (S|H): Projective = mk_proj!\(\s*\[([0-9,\s]+)\],\s*\[([0-9,\s]+)\],\s*\[([0-9,\s]+)\]\s*\);", text)
    assert [p[0] for p in projs] == ["S", "H"]

    def limbs(s):
        return [int(t) for t in s.replace("\n", " ").split(",") if t.strip()]

    out = {"source": "code/src/consts.rs (rasmus-kirk/halo-accumulation @ 2025-02-22)",
           "encoding": "canonical (non-Montgomery) big-endian hex; affine"}
    h_sh = hashlib.sha256()
    for name, xs, ys, zs in projs:
        lx, ly, lz = limbs(xs), limbs(ys), limbs(zs)
        for l in (lx, ly, lz):
            h_sh.update(b"".join(int(v).to_bytes(8, "little") for v in l))
        X, Y, Z = (pm.from_mont_limbs(l, pm.P) for l in (lx, ly, lz))
        pt = pm.jacobian_to_affine(X, Y, Z)
        assert pm.is_on_curve(pt)
        out[name] = ["%064x" % pt[0], "%064x" % pt[1]]
        out[name + "_jacobian_mont_limbs"] = [lx, ly, lz]
    out["SH_jacobian_limbs_sha256"] = h_sh.hexdigest()

    h_gs = hashlib.sha256()
    pts = []
    for xs, ys in aff:
        lx, ly = limbs(xs), limbs(ys)
        h_gs.update(b"".join(int(v).to_bytes(8, "little") for v in lx + ly))
        pts.append((pm.from_mont_limbs(lx, pm.P), pm.from_mont_limbs(ly, pm.P)))
    out["GS_count"] = len(pts)
    out["GS_mont_limbs_sha256"] = h_gs.hexdigest()
    out["GS_head"] = [["%064x" % x, "%064x" % y] for x, y in pts[:64]]
    out["GS_16383"] = ["%064x" % pts[16383][0], "%064x" % pts[16383][1]]
    # first entry also as raw limbs: pins the Montgomery radix / limb order
    out["GS_0_mont_limbs"] = [limbs(aff[0][0]), limbs(aff[0][1])]
    for p in pts[:64] + [pts[16383]]:
        assert pm.is_on_curve(p)

    with open(os.path.join(HERE, "urs_kat.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote urs_kat.json:", out["GS_mont_limbs_sha256"])


if __name__ == "__main__":
    main()
