"""Pins the oracle (Python big-int model + C restatement) on the reference's golden data.

The only hard-coded numeric truth in the reference is code/src/consts.rs
(SURVEY.md section 8c); everything else is the algebraic identities of its unit tests
(pcdl.rs:352-509, acc.rs:299-315, pedersen.rs:30-63), restated here.
"""
import hashlib

import numpy as np
import pytest

import orc
import pallas_model as pm


def hx(pt):
    return ["%064x" % pt[0], "%064x" % pt[1]]


# ---------------------------------------------------------------- consts.rs KAT
def test_model_reproduces_consts_samples(kat):
    S, H, G = pm.get_pp(8)
    assert hx(S) == kat["S"] and hx(H) == kat["H"]
    for i in range(8):
        assert hx(G[i]) == kat["GS_head"][i]
    assert hx(pm.get_generator_hash(16383 + 2)) == kat["GS_16383"]
    assert pm.to_mont_limbs(G[0][0], pm.P) == kat["GS_0_mont_limbs"][0]
    assert pm.to_mont_limbs(G[0][1], pm.P) == kat["GS_0_mont_limbs"][1]


def test_survey_anchor_constants():
    # SURVEY.md 8c "concrete anchors"
    assert "%064x" % (pm.MONT_R % pm.P) == "3fffffffffffffffffffffffffffffff992c350be41914ad34786d38fffffffd"
    assert "%064x" % (pm.MONT_R % pm.R_ORDER) == "3fffffffffffffffffffffffffffffff992c350be34205675b2b3e9cfffffffd"


def test_c_restatement_reproduces_whole_gs_table(kat):
    """All 16,384 GS entries, bit-for-bit in the reference's Montgomery limb encoding."""
    gs = orc.urs_affine(2, kat["GS_count"])
    assert hashlib.sha256(gs.tobytes()).hexdigest() == kat["GS_mont_limbs_sha256"]
    for i in range(64):
        assert hx(orc.affine_canonical(gs[i])) == kat["GS_head"][i]
    assert hx(orc.affine_canonical(gs[16383])) == kat["GS_16383"]


def test_c_restatement_S_H(kat, urs4096):
    pp = orc.make_pp(urs4096)
    assert hx(orc.point_canonical(np.array(pp.S, dtype=np.uint64))) == kat["S"]
    assert hx(orc.point_canonical(np.array(pp.H, dtype=np.uint64))) == kat["H"]
    # the reference stores S, H as un-normalised Jacobian limbs: same point
    for name in ("S", "H"):
        lx, ly, lz = kat[name + "_jacobian_mont_limbs"]
        j = np.array(lx + ly + lz, dtype=np.uint64)
        assert hx(orc.point_canonical(j)) == kat[name]


# ------------------------------------------------------------------ field / rng
def test_field_ops_vs_model():
    rng = pm.SplitMix64(0x1234)
    for _ in range(50):
        a, b = rng.next_scalar(), rng.next_scalar()
        am, bm = orc.fr_to_mont(a), orc.fr_to_mont(b)
        assert am.tolist() == pm.to_mont_limbs(a, pm.R_ORDER)
        o = orc.z(4); orc.lib().orc_fr_mul(orc.ptr(am), orc.ptr(bm), orc.ptr(o))
        assert orc.fr_from_mont(o) == a * b % pm.R_ORDER
        orc.lib().orc_fr_add(orc.ptr(am), orc.ptr(bm), orc.ptr(o))
        assert orc.fr_from_mont(o) == (a + b) % pm.R_ORDER
        assert orc.lib().orc_fr_inv(orc.ptr(am), orc.ptr(o)) == 0
        assert orc.fr_from_mont(o) == pm.inv_mod(a, pm.R_ORDER)
    for edge in (0, 1, pm.R_ORDER - 1, pm.MONT_R % pm.R_ORDER, 1 << 254):
        assert orc.fr_from_mont(orc.fr_to_mont(edge)) == edge % pm.R_ORDER


def test_rng_matches_model():
    sc, st = orc.rng_scalars(0x48414C4F00000001, 16)
    rng = pm.SplitMix64(0x48414C4F00000001)
    assert [orc.fr_from_mont(s) for s in sc] == [rng.next_scalar() for _ in range(16)]
    assert st == rng.state


def test_sha3_matches_hashlib():
    for n in (0, 1, 135, 136, 137, 300):
        data = bytes(range(256)) * 2
        data = data[:n]
        out = (orc.C.c_uint8 * 32)()
        orc.lib().orc_sha3_256(data, orc.C.c_size_t(n), out)
        assert bytes(out) == hashlib.sha3_256(data).digest()


# ------------------------------------------------------------------------ MSM
EDGE = [0, 1, pm.R_ORDER - 1, 1 << 254, 2, pm.R_ORDER - 2]


@pytest.mark.parametrize("n", [1, 2, 3, 31, 32, 33, 256])
def test_msm_pippenger_vs_model(urs4096, n):
    sc, _ = orc.rng_scalars(0x48414C4F00000002 + n, n)
    for i, e in enumerate(EDGE[: min(n, len(EDGE))]):
        sc[(i * 7) % n] = orc.fr_to_mont(e)
    xs = [orc.fr_from_mont(s) for s in sc]
    G = [orc.affine_canonical(g) for g in urs4096[:n]]
    want = pm.point_dot(xs, G)
    assert orc.point_canonical(orc.msm_affine(urs4096[:n], sc)) == want
    assert orc.point_canonical(orc.msm_naive(urs4096[:n], sc)) == want


def test_msm_4096_pippenger_vs_naive(urs4096):
    sc, _ = orc.rng_scalars(7, 4096)
    assert orc.point_canonical(orc.msm_affine(urs4096, sc)) == orc.point_canonical(orc.msm_naive(urs4096, sc))


def test_msm_degenerate_inputs(urs4096):
    n = 64
    zero = np.zeros((n, 4), dtype=np.uint64)
    assert orc.point_canonical(orc.msm_affine(urs4096[:n], zero)) is None
    one = np.tile(orc.fr_to_mont(1), (n, 1))
    G = [orc.affine_canonical(g) for g in urs4096[:n]]
    acc = None
    for g in G:
        acc = pm.add(acc, g)
    assert orc.point_canonical(orc.msm_affine(urs4096[:n], one)) == acc
    minus1 = np.tile(orc.fr_to_mont(pm.R_ORDER - 1), (n, 1))
    assert orc.point_canonical(orc.msm_affine(urs4096[:n], minus1)) == pm.neg(acc)
    same = np.ascontiguousarray(np.tile(urs4096[5], (n, 1)))
    sc, _ = orc.rng_scalars(9, n)
    tot = sum(orc.fr_from_mont(s) for s in sc) % pm.R_ORDER
    assert orc.point_canonical(orc.msm_affine(same, sc)) == pm.mul(G[5], tot)
    # +s and -s on the same base cancel
    sc2 = np.concatenate([sc[:1], orc.scalars_to_mont([pm.R_ORDER - orc.fr_from_mont(sc[0])])])
    assert orc.point_canonical(orc.msm_affine(same[:2], sc2)) is None


# -------------------------------------------------- pcdl.rs unit-test identities
def test_u_check_anchor(urs4096):
    """pcdl.rs:382-438 with xi = [0,1,2,3]: fold of G == MSM(GS, h_coeffs) == SURVEY anchor."""
    xis = [0, 1, 2, 3]
    assert pm.h_coeffs(xis) == [1, 3, 2, 6, 1, 3, 2, 6]
    G = [orc.affine_canonical(g) for g in urs4096[:8]]
    gs = list(G)
    for i in range(3):
        half = len(gs) // 2
        gs = [pm.add(gs[j], pm.mul(gs[j + half], xis[i + 1])) for j in range(half)]
    U = gs[0]
    want = ("18cef7a91c998eab6266eaa5c7523a520b6f9b56aefe02b7cb48b226b9c0530c",
            "2cc9cee89d461087f1312759efb678ec548f5cda04f99a56429ae889cc2d7da3")
    assert tuple(hx(U)) == want
    assert pm.point_dot(pm.h_coeffs(xis), G) == U
    xm = orc.scalars_to_mont(xis)
    hc = orc.h_coeffs(xm)
    assert [orc.fr_from_mont(c) for c in hc] == [1, 3, 2, 6, 1, 3, 2, 6]
    assert tuple(hx(orc.point_canonical(orc.msm_affine(urs4096[:8], hc)))) == want
    # same fold through the C restatement's round function
    gj = np.zeros((8, 12), dtype=np.uint64)
    for i in range(8):
        orc.lib().orc_affine_to_jac(orc.ptr(urs4096[i]), orc.ptr(gj[i]))
    cs = np.zeros((8, 4), dtype=np.uint64); zs = np.zeros((8, 4), dtype=np.uint64)
    m = 4
    for i in range(3):
        one = orc.fr_to_mont(1)
        orc.lib().orc_ipa_round_fold(orc.ptr(gj), orc.ptr(cs), orc.ptr(zs), orc.C.c_size_t(m), orc.ptr(xm[i + 1]), orc.ptr(one))
        m //= 2
    assert tuple(hx(orc.point_canonical(gj[0]))) == want


def test_construct_h_with_degree_7():
    """pcdl.rs:486-509."""
    rng = pm.SplitMix64(77)
    xis = [rng.next_scalar() for _ in range(4)]
    want = [1, xis[3], xis[2], xis[2] * xis[3], xis[1], xis[1] * xis[3], xis[1] * xis[2], xis[1] * xis[2] * xis[3]]
    want = [w % pm.R_ORDER for w in want]
    assert pm.h_coeffs(xis) == want
    assert [orc.fr_from_mont(c) for c in orc.h_coeffs(orc.scalars_to_mont(xis))] == want


@pytest.mark.parametrize("lg_n", [1, 2, 3, 5, 9])
def test_h_eval_product_formula(lg_n):
    """pcdl.rs:352-379 (test_test) + h.eval == evaluate(get_poly)."""
    rng = pm.SplitMix64(100 + lg_n)
    z = rng.next_scalar()
    xis = [rng.next_scalar() for _ in range(lg_n + 1)]
    v2 = 1
    for i in range(lg_n):
        v2 = v2 * (1 + xis[lg_n - i] * pow(z, 1 << i, pm.R_ORDER)) % pm.R_ORDER
    assert pm.h_eval(xis, z) == v2
    xm, zm = orc.scalars_to_mont(xis), orc.fr_to_mont(z)
    assert orc.fr_from_mont(orc.h_eval(xm, zm)) == v2
    assert orc.fr_from_mont(orc.poly_eval(orc.h_coeffs(xm), zm)) == v2


def test_pedersen_homomorphism(urs4096):
    """pedersen.rs:30-63, l = 64."""
    pp = orc.make_pp(urs4096)
    l = 64
    m1, s = orc.rng_scalars(11, l)
    m2, s = orc.rng_scalars(s, l)
    ws, _ = orc.rng_scalars(s, 2)
    msum = np.zeros_like(m1)
    for i in range(l):
        orc.lib().orc_fr_add(orc.ptr(m1[i]), orc.ptr(m2[i]), orc.ptr(msum[i]))
    wsum = orc.z(4); orc.lib().orc_fr_add(orc.ptr(ws[0]), orc.ptr(ws[1]), orc.ptr(wsum))

    def commit(w, ms):
        o = orc.z(12)
        rc = orc.lib().orc_pedersen_commit(orc.C.byref(pp), orc.ptr(w), orc.ptr(urs4096[:l]), orc.C.c_size_t(l), orc.ptr(ms), orc.C.c_size_t(l), orc.ptr(o))
        assert rc == 0
        return o
    inner = commit(wsum, msum)
    o = orc.z(12); orc.lib().orc_point_add(orc.ptr(commit(ws[0], m1)), orc.ptr(commit(ws[1], m2)), orc.ptr(o))
    assert orc.point_canonical(inner) == orc.point_canonical(o)
    # length mismatch is the reference's assert! (pedersen.rs:7-12)
    bad = orc.z(12)
    assert orc.lib().orc_pedersen_commit(orc.C.byref(pp), None, orc.ptr(urs4096[:l]), orc.C.c_size_t(l), orc.ptr(m1), orc.C.c_size_t(l - 1), orc.ptr(bad)) < 0


# ------------------------------------------------ open / check vs the model
def _model_pp(urs, n):
    S, H, _ = pm.get_pp(0)
    return pm.PublicParams(S, H, [orc.affine_canonical(g) for g in urs[:n]])


def _proof_to_model(pf, lg):
    w = pf[2:].reshape(-1)
    Ls = [orc.point_canonical(w[12 * i: 12 * i + 12]) for i in range(lg)]
    Rs = [orc.point_canonical(w[12 * lg + 12 * i: 12 * lg + 12 * i + 12]) for i in range(lg)]
    o = 24 * lg
    hiding = bool(pf[0])
    return dict(Ls=Ls, Rs=Rs, U=orc.point_canonical(w[o:o + 12]), c=orc.fr_from_mont(w[o + 12:o + 16]),
                C_bar=orc.point_canonical(w[o + 16:o + 28]) if hiding else None,
                w_prime=orc.fr_from_mont(w[o + 28:o + 32]) if hiding else None)


@pytest.mark.parametrize("n,hiding", [(4, False), (8, True), (16, False), (32, True)])
def test_open_matches_model_and_checks(urs4096, n, hiding):
    """pcdl.rs:441-483 completeness + C restatement == big-int model on the same inputs."""
    d = n - 1
    pp = orc.make_pp(urs4096)
    mpp = _model_pp(urs4096, n)
    deg = max(1, d - 2)
    coeffs, s = orc.rng_scalars(0x48414C4F00000003 + n, deg + 1)
    zw, s = orc.rng_scalars(s, 2)
    zm, wm = zw[0], (zw[1] if hiding else None)
    C = orc.pcdl_commit(pp, coeffs, d, wm)
    open_seed = 4242 + n
    pf, _ = orc.pcdl_open(pp, open_seed, coeffs, C, d, zm, wm)
    v = orc.poly_eval(coeffs, zm)
    orc.pcdl_check(pp, C, d, zm, v, pf)
    # model side, same randomness stream (q coefficients then w_bar)
    cs = [orc.fr_from_mont(c) for c in coeffs]
    z_i = orc.fr_from_mont(zm)
    w_i = orc.fr_from_mont(wm) if hiding else None
    rng = pm.SplitMix64(open_seed)
    q = [rng.next_scalar() for _ in range(deg)] if hiding else None
    w_bar = rng.next_scalar() if hiding else None
    Cm = pm.pcdl_commit(mpp, cs, d, w_i)
    assert Cm == orc.point_canonical(C)
    pim = pm.pcdl_open(mpp, cs, Cm, d, z_i, w_i, q, w_bar)
    assert pim == _proof_to_model(pf, n.bit_length() - 1)
    pm.pcdl_check(mpp, Cm, d, z_i, orc.fr_from_mont(v), pim)
    # soundness smoke: wrong v is rejected
    bad_v = orc.fr_to_mont((orc.fr_from_mont(v) + 1) % pm.R_ORDER)
    with pytest.raises(ValueError):
        orc.pcdl_check(pp, C, d, zm, bad_v, pf)
    with pytest.raises(ValueError):
        pm.pcdl_check(mpp, Cm, d, z_i, (orc.fr_from_mont(v) + 1) % pm.R_ORDER, pim)


@pytest.mark.parametrize("n", [64, 512])
def test_open_check_completeness_larger(urs4096, n):
    d = n - 1
    pp = orc.make_pp(urs4096)
    coeffs, s = orc.rng_scalars(n, n // 2 + 3)
    zw, s = orc.rng_scalars(s, 2)
    for w in (None, zw[1]):
        C = orc.pcdl_commit(pp, coeffs, d, w)
        pf, _ = orc.pcdl_open(pp, 99, coeffs, C, d, zw[0], w)
        orc.pcdl_check(pp, C, d, zw[0], orc.poly_eval(coeffs, zw[0]), pf)


def test_commit_asserts(urs4096):
    pp = orc.make_pp(urs4096[:16])
    coeffs, _ = orc.rng_scalars(5, 8)
    with pytest.raises(AssertionError):
        orc.pcdl_commit(pp, coeffs, 6)       # n = 7 not a power of two (pcdl.rs:102)
    with pytest.raises(AssertionError):
        orc.pcdl_commit(pp, coeffs, 3)       # degree 7 > d (pcdl.rs:103)
    with pytest.raises(AssertionError):
        orc.pcdl_commit(pp, coeffs, 31)      # d > D (pcdl.rs:104)


# ------------------------------------------------------------------- acc.rs
@pytest.mark.parametrize("n,steps", [(4, 3), (16, 2)])
def test_acc_scheme(urs4096, n, steps):
    """acc.rs:299-315: chained prover+verifier, then decider."""
    d = n - 1
    pp = orc.make_pp(urs4096)
    seed = 1000 + n
    acc = None
    lg = n.bit_length() - 1
    for _ in range(steps):
        q, seed = orc.random_instance(pp, seed, d)
        qs = [q] if acc is None else [acc[: orc.instance_words(lg)].copy(), q]
        acc, seed = orc.acc_prover(pp, seed, d, qs)
        orc.acc_verifier(pp, d, qs, acc)
        bad = acc.copy(); bad[17] ^= 1
        with pytest.raises(ValueError):
            orc.acc_verifier(pp, d, qs, bad)
    orc.acc_decider(pp, acc)
