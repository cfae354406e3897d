"""Committed golden vectors (tests/golden/model_vectors.json, produced by the big-int model) against
the C restatement (CPU) and the HIP path through the C ABI (GPU)."""
import json
import os

import numpy as np
import pytest

import orc
import pallas_model as pm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def vec():
    with open(os.path.join(ROOT, "tests", "golden", "model_vectors.json")) as f:
        return json.load(f)


def P(p):
    return None if p is None else (int(p[0], 16), int(p[1], 16))


def mont(hexes):
    return orc.scalars_to_mont([int(h, 16) for h in hexes])


# ------------------------------------------------------------------ CPU: oracle vs fixtures
def test_oracle_msm_vectors(vec, urs4096):
    for t in vec["msm"]:
        n = t["n"]
        sc = mont(t["scalars"])
        assert orc.point_canonical(orc.msm_affine(urs4096[:n], sc)) == P(t["result"])


def test_oracle_fold_vector(vec, urs4096):
    f = vec["fold"]
    pp = orc.make_pp(urs4096[:8])
    gj = np.zeros((8, 12), dtype=np.uint64)
    for i in range(8):
        orc.lib().orc_affine_to_jac(orc.ptr(urs4096[i]), orc.ptr(gj[i]))
    cs = mont(f["c"]); zs = orc.powers(orc.fr_to_mont(int(f["z"], 16)), 8)
    L, R = orc.z(12), orc.z(12)
    Hp = np.array(pp.H, dtype=np.uint64)
    orc.lib().orc_ipa_round_lr(orc.ptr(gj), orc.ptr(cs), orc.ptr(zs), orc.C.c_size_t(4), orc.ptr(Hp), orc.ptr(L), orc.ptr(R))
    assert orc.point_canonical(L) == P(f["L"]) and orc.point_canonical(R) == P(f["R"])
    xi = orc.fr_to_mont(int(f["xi"], 16)); xi_inv = orc.z(4); orc.lib().orc_fr_inv(orc.ptr(xi), orc.ptr(xi_inv))
    orc.lib().orc_ipa_round_fold(orc.ptr(gj), orc.ptr(cs), orc.ptr(zs), orc.C.c_size_t(4), orc.ptr(xi), orc.ptr(xi_inv))
    assert [orc.point_canonical(gj[j]) for j in range(4)] == [P(p) for p in f["G_out"]]
    assert [orc.fr_from_mont(cs[j]) for j in range(4)] == [int(h, 16) for h in f["c_out"]]
    assert [orc.fr_from_mont(zs[j]) for j in range(4)] == [int(h, 16) for h in f["z_out"]]


def test_oracle_h_vectors(vec, urs4096):
    for t in vec["h"]:
        xis = mont(t["xis"]); z = orc.fr_to_mont(int(t["z"], 16))
        co = orc.h_coeffs(xis)
        assert [orc.fr_from_mont(c) for c in co[:8]] == [int(h, 16) for h in t["coeffs_head"]][: len(co)]
        assert [orc.fr_from_mont(c) for c in co[-2:]] == [int(h, 16) for h in t["coeffs_tail"]]
        assert orc.fr_from_mont(orc.h_eval(xis, z)) == int(t["eval"], 16)
        if t["commit"]:
            assert orc.point_canonical(orc.msm_affine(urs4096[: 1 << t["lg_n"]], co)) == P(t["commit"])


def _check_open_vector(t, commit, open_, succinct_check):
    coeffs = mont(t["coeffs"]); z = orc.fr_to_mont(int(t["z"], 16))
    C = commit(coeffs)
    assert orc.point_canonical(C) == P(t["C"])
    pi = open_(coeffs, C, z)
    w = pi[2:]
    assert [orc.point_canonical(w[12 * i: 12 * i + 12]) for i in range(3)] == [P(p) for p in t["Ls"]]
    assert [orc.point_canonical(w[36 + 12 * i: 48 + 12 * i]) for i in range(3)] == [P(p) for p in t["Rs"]]
    assert orc.point_canonical(w[72:84]) == P(t["U"]) and orc.fr_from_mont(w[84:88]) == int(t["c"], 16)
    xis, _ = succinct_check(C, z, orc.fr_to_mont(int(t["v"], 16)), pi)
    assert [orc.fr_from_mont(x) for x in xis] == [int(h, 16) for h in t["xis"]]


def test_oracle_open_transcript_vector(vec, urs4096):
    pp = orc.make_pp(urs4096[:8])
    _check_open_vector(vec["open_n8"], lambda c: orc.pcdl_commit(pp, c, 7), lambda c, C, z: orc.pcdl_open(pp, 1, c, C, 7, z)[0],
                       lambda C, z, v, pi: orc.pcdl_succinct_check(pp, C, 7, z, v, pi))


# ------------------------------------------------------------------ GPU: HIP path vs fixtures
@pytest.fixture(scope="module")
def gctx():
    import halo_accumulation_amd as h
    c = h._lib.Context(urs_n=1024)
    yield c
    c.close()


@pytest.mark.gpu
def test_gpu_msm_vectors(vec, gctx):
    for t in vec["msm"]:
        assert orc.point_canonical(gctx.msm(mont(t["scalars"]))) == P(t["result"])
    assert orc.point_canonical(gctx.msm(np.zeros((0, 4), dtype=np.uint64))) is None  # empty input


@pytest.mark.gpu
@pytest.mark.parametrize("switch", [0, 1 << 16])
def test_gpu_fold_vector(vec, gctx, switch):
    import halo_accumulation_amd as h
    f = vec["fold"]
    gctx.set_ipa_switch(switch)
    try:
        _, H = h._lib.public_points()
        ipa = h._lib.Ipa(gctx, 8, mont(f["c"]), orc.fr_to_mont(int(f["z"], 16)))
        L, R = ipa.round_lr(H)
        assert orc.point_canonical(L) == P(f["L"]) and orc.point_canonical(R) == P(f["R"])
    finally:
        gctx.set_ipa_switch(1 << 14)


@pytest.mark.gpu
def test_gpu_h_vectors(vec, gctx):
    for t in vec["h"]:
        xis = mont(t["xis"]); z = orc.fr_to_mont(int(t["z"], 16))
        co = gctx.h_coeffs(xis)
        assert [orc.fr_from_mont(c) for c in co[-2:]] == [int(h, 16) for h in t["coeffs_tail"]]
        assert orc.fr_from_mont(gctx.h_eval_batch(xis[None], z)[0]) == int(t["eval"], 16)
        if t["commit"]:
            assert orc.point_canonical(gctx.h_commit(xis)) == P(t["commit"])


@pytest.mark.gpu
def test_gpu_open_transcript_vector(vec, gctx):
    from halo_accumulation_amd import pcdl
    _check_open_vector(vec["open_n8"], lambda c: pcdl.commit(gctx, c, 7), lambda c, C, z: pcdl.open(gctx, [1], c, C, 7, z),
                       lambda C, z, v, pi: pcdl.succinct_check(gctx, C, 7, z, v, pi))
