"""GPU parity for the kernels behind pcdl::open / check and the acc scheme, written after the
reference's own unit tests (pcdl.rs:352-509, acc.rs:299-315, pedersen.rs:30-63), plus
blob-for-blob equality with the CPU restatement on the same seeded inputs."""
import numpy as np
import pytest

import os

import orc
import pallas_model as pm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hal():
    import halo_accumulation_amd as h
    return h


@pytest.fixture(scope="module")
def ctx(hal):
    c = hal._lib.Context(urs_n=4096)
    yield c
    c.close()


@pytest.fixture(scope="module")
def pp(ctx):
    gs = ctx.read_bases()
    return orc.make_pp(gs)


def canon(j):
    return orc.point_canonical(j)


# ------------------------------------------------------------------ Fr kernels (K4-K9)
@pytest.mark.parametrize("m", [1, 2, 63, 64, 65, 1000, 4096])
def test_scalar_dot_powers_poly_eval(ctx, m):
    xs, s = orc.rng_scalars(m, m)
    ys, s = orc.rng_scalars(s, m)
    z, _ = orc.rng_scalars(s, 1)
    assert ctx.scalar_dot(xs, ys).tolist() == orc.scalar_dot(xs, ys).tolist()
    assert ctx.powers(z[0], m).tolist() == orc.powers(z[0], m).tolist()
    assert ctx.poly_eval(xs, z[0]).tolist() == orc.poly_eval(xs, z[0]).tolist()


@pytest.mark.parametrize("lg_n", [1, 2, 3, 7, 8, 9, 12])
def test_h_coeffs_eval_commit(ctx, lg_n):
    """pcdl.rs:486-509 ordering, pcdl.rs:352-379 eval, pcdl.rs:338 commit."""
    xis, s = orc.rng_scalars(100 + lg_n, lg_n + 1)
    z, _ = orc.rng_scalars(s, 1)
    want = orc.h_coeffs(xis)
    assert ctx.h_coeffs(xis).tolist() == want.tolist()
    assert ctx.h_eval_batch(xis[None], z[0])[0].tolist() == orc.h_eval(xis, z[0]).tolist()
    gs = ctx.read_bases(0, 1 << lg_n)
    assert ctx.h_commit(xis).tolist() == orc.msm_affine(gs, want).tolist()


def test_h_eval_batch_and_accumulate(ctx):
    lg_n, m = 10, 37
    flat, s = orc.rng_scalars(5, m * (lg_n + 1))
    xis = flat.reshape(m, lg_n + 1, 4)
    z, s = orc.rng_scalars(s, 1)
    got = ctx.h_eval_batch(xis, z[0])
    for i in range(m):
        assert got[i].tolist() == orc.h_eval(np.ascontiguousarray(xis[i]), z[0]).tolist()
    # acc.rs:85-94: h0 + sum alpha_i h_i
    al, s = orc.rng_scalars(s, 3)
    h0, _ = orc.rng_scalars(s, 2)
    got = ctx.h_accumulate(h0, xis[:3], al)
    want = [0] * (1 << lg_n)
    want[0], want[1] = orc.fr_from_mont(h0[0]), orc.fr_from_mont(h0[1])
    for i in range(3):
        a = orc.fr_from_mont(al[i])
        hc = orc.h_coeffs(np.ascontiguousarray(xis[i]))
        for k in range(1 << lg_n):
            want[k] = (want[k] + a * orc.fr_from_mont(hc[k])) % pm.R_ORDER
    assert [orc.fr_from_mont(g) for g in got] == want


def test_u_check(hal, ctx, ipa_mode):
    """pcdl.rs:382-438 with xi = [0,1,2,3]: device fold of G == MSM(GS, h_coeffs) == SURVEY anchor."""
    xm = orc.scalars_to_mont([0, 1, 2, 3])
    zero = np.zeros((8, 4), dtype=np.uint64)
    one = orc.fr_to_mont(1)
    ipa = hal._lib.Ipa(ctx, 8, zero, one)
    for i in range(3):
        ipa.round_fold(xm[i + 1], one)
    U, c = ipa.finish()
    want = ("18cef7a91c998eab6266eaa5c7523a520b6f9b56aefe02b7cb48b226b9c0530c",
            "2cc9cee89d461087f1312759efb678ec548f5cda04f99a56429ae889cc2d7da3")
    assert tuple("%064x" % v for v in canon(U)) == want
    assert tuple("%064x" % v for v in canon(ctx.h_commit(xm))) == want
    assert [orc.fr_from_mont(c) for c in ctx.h_coeffs(xm)] == [1, 3, 2, 6, 1, 3, 2, 6]


@pytest.fixture(params=[(0, 1, 0), (1 << 16, 2, -1), (16, 1, 0), (0, 2, 0), (16, 2, 0), (0, 2, 1), (16, 2, 1), (64, 2, 0, 1), (16, 2, 1, 1), (0, 2, 0, 1)],
                ids=["always-fold", "default", "switch-at-16", "two-level-folds-to-the-end", "two-level-folds-switch-at-16",
                     "fold-table-two-level-folds-to-the-end", "fold-table-switch-at-16", "folds-beside-the-rounds-switch-at-64",
                     "fold-table-folds-beside-the-rounds-switch-at-16", "folds-beside-the-rounds-to-the-end"])
def ipa_mode(request, ctx):
    """Every IPA strategy must give the reference's results: folding G every round (k_fold_points), every other round
    (two halvings per pass, k_fold_points4, L/R from MSMs over the unfolded key in between), the first of those passes from
    the comb table over the context's key (k_fold_tab4: only when the open has the size of the key), the no-fold late
    rounds (MSMs over the fixed folded key), and the two-level folds launched BESIDE the next two rounds (halo_set_fold_async)."""
    ctx.set_ipa_switch(request.param[0])
    ctx.set_fold_levels(request.param[1])
    ctx.set_fold_table(request.param[2])
    ctx.set_fold_async(request.param[3] if len(request.param) > 3 else 0)  # (folds of >= 64 outputs on the fourth stream, beside the next two rounds)
    yield request.param
    ctx.set_fold_async(-1)
    ctx.set_ipa_switch(1 << 14)
    ctx.set_fold_levels(2)
    ctx.set_fold_table(0)   # (release the table)
    ctx.set_fold_table(-1)


@pytest.mark.parametrize("n", [2, 8, 64, 1024])
def test_ipa_rounds_vs_oracle(hal, ctx, pp, n, ipa_mode):
    """Kernel-level parity of pcdl.rs:195-227: L, R and the folded state, round by round."""
    gs = ctx.read_bases(0, n)
    coeffs, s = orc.rng_scalars(n, n - 1 if n > 2 else n)
    zx, s = orc.rng_scalars(s, 2)
    z = zx[0]
    Hp = np.array(pp.H, dtype=np.uint64)
    ipa = hal._lib.Ipa(ctx, n, coeffs, z)
    gj = np.zeros((n, 12), dtype=np.uint64)
    for i in range(n):
        orc.lib().orc_affine_to_jac(orc.ptr(gs[i]), orc.ptr(gj[i]))
    cs = np.zeros((n, 4), dtype=np.uint64); cs[: len(coeffs)] = coeffs
    zs = orc.powers(z, n)
    m = n // 2
    seed = s
    while m >= 1:
        L, R = ipa.round_lr(Hp)
        Lw, Rw = orc.z(12), orc.z(12)
        orc.lib().orc_ipa_round_lr(orc.ptr(gj), orc.ptr(cs), orc.ptr(zs), orc.C.c_size_t(m), orc.ptr(Hp), orc.ptr(Lw), orc.ptr(Rw))
        assert L.tolist() == Lw.tolist() and R.tolist() == Rw.tolist()
        xi, seed = orc.rng_scalars(seed, 1)
        xi_inv = orc.z(4); assert orc.lib().orc_fr_inv(orc.ptr(xi[0]), orc.ptr(xi_inv)) == 0
        ipa.round_fold(xi[0], xi_inv)
        orc.lib().orc_ipa_round_fold(orc.ptr(gj), orc.ptr(cs), orc.ptr(zs), orc.C.c_size_t(m), orc.ptr(xi[0]), orc.ptr(xi_inv))
        m //= 2
    U, c = ipa.finish()
    assert canon(U) == canon(gj[0]) and c.tolist() == cs[0].tolist()


# ------------------------------------------------------------------ pcdl level
def test_pedersen_homomorphism(hal, ctx):
    """pedersen.rs:30-63"""
    from halo_accumulation_amd import pedersen
    l = 64
    m1, s = orc.rng_scalars(11, l)
    m2, s = orc.rng_scalars(s, l)
    ws, _ = orc.rng_scalars(s, 2)
    msum = ctx.field_op(1, 1, m1, m2)
    wsum = ctx.field_op(1, 1, ws[:1], ws[1:2])[0]
    inner = pedersen.commit(ctx, wsum, l, msum)
    o = orc.z(12)
    orc.lib().orc_point_add(orc.ptr(pedersen.commit(ctx, ws[0], l, m1)), orc.ptr(pedersen.commit(ctx, ws[1], l, m2)), orc.ptr(o))
    assert inner.tolist() == o.tolist()
    with pytest.raises(AssertionError):
        pedersen.commit(ctx, None, l, m1[:-1])


@pytest.mark.parametrize("l", [1, 3, 64, 1000, 4096])
def test_pedersen_commit_over_callers_generators(hal, ctx, pp, l):
    """pedersen::commit takes ANY `&[PallasAffine]` (pedersen.rs:6): generators that are not a stretch of the context's key
    go through halo_msm_affine (bases uploaded per call).  Reversed key, a key derived from other URS indices, an
    infinity among the bases -- against the oracle's arkworks-style Pippenger over the same arrays."""
    from halo_accumulation_amd import group, pedersen
    other = hal._lib.Context(urs_n=l, first_index=100000)
    try:
        foreign = other.read_bases()
    finally:
        other.close()
    rev = ctx.read_bases()[::-1][:l].copy()
    holes = foreign.copy()
    holes[l // 2] = 0  # (0, 0) = point at infinity
    ms, s = orc.rng_scalars(900 + l, l)
    w, _ = orc.rng_scalars(s, 1)
    for Gs in (foreign, rev, holes):
        assert group.point_dot_affine(ctx, ms, Gs=Gs).tolist() == orc.msm_affine(Gs, ms).tolist()
        for ww in (None, w[0]):
            want = orc.z(12)
            assert orc.lib().orc_pedersen_commit(orc.C.byref(pp), None if ww is None else orc.ptr(ww), orc.ptr(Gs), orc.C.c_size_t(l), orc.ptr(ms),
                                                 orc.C.c_size_t(l), orc.ptr(want)) == 0
            assert canon(pedersen.commit(ctx, ww, Gs, ms)) == canon(want)
    # homomorphism over the caller's generators (pedersen.rs:30-63 with Gs that are not consts::GS)
    m2, s = orc.rng_scalars(s, l)
    o = orc.z(12)
    orc.lib().orc_point_add(orc.ptr(pedersen.commit(ctx, None, foreign, ms)), orc.ptr(pedersen.commit(ctx, None, foreign, m2)), orc.ptr(o))
    assert pedersen.commit(ctx, None, foreign, ctx.field_op(1, 1, ms, m2)).tolist() == o.tolist()
    with pytest.raises(AssertionError):
        pedersen.commit(ctx, None, foreign, ms[:-1]) if l > 1 else pedersen.commit(ctx, None, foreign, np.zeros((2, 4), dtype=np.uint64))


@pytest.mark.parametrize("n,hiding", [(4, False), (4, True), (16, True), (512, False), (512, True), (4096, True)])
def test_open_check_matches_oracle(hal, ctx, pp, n, hiding, ipa_mode):
    """pcdl.rs:441-483 completeness, and proof blobs identical to the CPU restatement's."""
    from halo_accumulation_amd import pcdl
    d = n - 1
    deg = max(1, d - 2) if n <= 16 else n // 2 + 3
    coeffs, s = orc.rng_scalars(0x48414C4F00000003 + n, deg + 1)
    zw, s = orc.rng_scalars(s, 2)
    z, w = zw[0], (zw[1] if hiding else None)
    C = pcdl.commit(ctx, coeffs, d, w)
    assert C.tolist() == orc.pcdl_commit(pp, coeffs, d, w).tolist()
    rng = [4242 + n]
    pi = pcdl.open(ctx, rng, coeffs, C, d, z, w)
    pi_ref, st_ref = orc.pcdl_open(pp, 4242 + n, coeffs, C, d, z, w)
    assert pi.tolist() == pi_ref.tolist()
    assert rng[0] == st_ref
    v = ctx.poly_eval(coeffs, z)
    pcdl.check_proof(ctx, C, d, z, v, pi)
    orc.pcdl_check(pp, C, d, z, v, pi)
    xis, U = pcdl.succinct_check(ctx, C, d, z, v, pi)
    xis_ref, U_ref = orc.pcdl_succinct_check(pp, C, d, z, v, pi)
    assert xis.tolist() == xis_ref.tolist() and U.tolist() == U_ref.tolist()
    bad_v = orc.fr_to_mont((orc.fr_from_mont(v) + 1) % pm.R_ORDER)
    with pytest.raises(ValueError):
        pcdl.check_proof(ctx, C, d, z, bad_v, pi)
    bad = pi.copy(); bad[2 + 24 * (n.bit_length() - 1) + 12] ^= 1  # c
    with pytest.raises(ValueError):
        pcdl.succinct_check(ctx, C, d, z, v, bad)
    # a proof whose U is not the commitment to h passes the succinct check only if C matches: flip U and c consistently is
    # infeasible, so just check that check() rejects a wrong U
    bad = pi.copy(); bad[2 + 24 * (n.bit_length() - 1): 2 + 24 * (n.bit_length() - 1) + 12] = C
    with pytest.raises(ValueError):
        pcdl.check_proof(ctx, C, d, z, v, bad)


@pytest.mark.parametrize("lg", [16, 18])
def test_folds_beside_the_rounds_give_the_same_proofs(hal, lg):
    """halo_set_fold_async: a two-level fold of a key of <= 2^18 points runs on the fourth stream while the next two rounds take
    their L, R from the key it reads; the proof (hiding and not) must be the one the in-line folds give, and the default
    (automatic: opens of <= 2^18 points) is one of the two."""
    import torch
    from halo_accumulation_amd import pcdl
    n = 1 << lg
    d = n - 1
    c = hal._lib.Context(urs_n=n)
    try:
        c.set_fold_table(0)
        dv = torch.empty((n + 2) * 4, dtype=torch.int64, device="cuda")
        c.rng_scalars_dev(0xA51C + lg, n + 2, dv.data_ptr())
        zw = np.ascontiguousarray(dv[4 * n:].cpu().numpy().view(np.uint64).reshape(2, 4))
        v = None
        for w in (None, zw[1]):
            C = pcdl.commit_dev(c, dv.data_ptr(), n, d, w)
            proofs = {}
            for mode in (0, 1, -1):
                c.set_fold_async(mode)
                for rep in range(2):  # (the second open replays the launch graphs of the first)
                    rng = [99]
                    proofs[(mode, rep)] = (pcdl.open_dev(c, rng, dv.data_ptr(), n, C, d, zw[0], w).tolist(), rng[0])
            assert all(p == proofs[(0, 0)] for p in proofs.values())
            if v is None:
                v = c.poly_eval(np.ascontiguousarray(dv[: 4 * n].cpu().numpy().view(np.uint64).reshape(n, 4)), zw[0])
            pcdl.check_proof(c, C, d, zw[0], v, np.array(proofs[(1, 0)][0], dtype=np.uint64))
    finally:
        c.close()


def test_commit_open_asserts(hal, ctx):
    from halo_accumulation_amd import pcdl
    coeffs, _ = orc.rng_scalars(5, 8)
    with pytest.raises(AssertionError):
        pcdl.commit(ctx, coeffs, 6)          # pcdl.rs:102
    with pytest.raises(AssertionError):
        pcdl.commit(ctx, coeffs, 3)          # pcdl.rs:103
    with pytest.raises(AssertionError):
        pcdl.commit(ctx, coeffs, 8191)       # pcdl.rs:104 (D = 4095 here)
    C = pcdl.commit(ctx, coeffs, 7)
    with pytest.raises(AssertionError):
        pcdl.open(ctx, [1], coeffs, C, 6, coeffs[0])   # pcdl.rs:130


# ------------------------------------------------------------------ acc level
@pytest.mark.parametrize("n,steps", [(4, 3), (16, 3), (1024, 2)])
def test_acc_scheme_matches_oracle(hal, ctx, pp, n, steps):
    """acc.rs:299-315: chained prover + verifier, then decider; accumulators equal the oracle's."""
    from halo_accumulation_amd import acc as A
    d = n - 1
    rng = [1000 + n]
    seed = 1000 + n
    acc = acc_ref = None
    for _ in range(steps):
        q = A.random_instance(ctx, rng, d)
        q_ref, seed = orc.random_instance(pp, seed, d)
        assert q.tolist() == q_ref.tolist() and rng[0] == seed
        qs = [q] if acc is None else [A.instance_from_accumulator(ctx, acc, d), q]
        acc = A.prover(ctx, rng, d, qs)
        acc_ref, seed = orc.acc_prover(pp, seed, d, qs)
        assert acc.tolist() == acc_ref.tolist() and rng[0] == seed
        A.verifier(ctx, d, qs, acc)
        orc.acc_verifier(pp, d, qs, acc)
        bad = acc.copy(); bad[17] ^= 1
        with pytest.raises(ValueError):
            A.verifier(ctx, d, qs, bad)
    A.decider(ctx, acc)
    orc.acc_decider(pp, acc)


def _plain_instance(ctx, seed, d):
    """an Instance with a non-hiding proof (random_instance always hides)"""
    from halo_accumulation_amd import pcdl
    coeffs, s = orc.rng_scalars(seed, d + 1 - (seed % 3))
    z, _ = orc.rng_scalars(s, 1)
    Cm = pcdl.commit(ctx, coeffs, d)
    pi = pcdl.open(ctx, [seed], coeffs, Cm, d, z[0])
    v = ctx.poly_eval(coeffs, z[0])
    return np.concatenate([Cm, np.array([d], dtype=np.uint64), z[0], v, pi])


@pytest.mark.parametrize("lg", [10, 20])
def test_succinct_check_batch_64_on_device(hal, lg):
    """SURVEY 8f-2 / acc.rs:158-170: 64 succinct checks in two device launches (k_h_eval_z, k_batch_small_msm), instance by
    instance equal to the oracle's pcdl::succinct_check; tampered instances are the ones -- and the only ones -- rejected,
    exactly as by the host path."""
    import json, os, time
    from halo_accumulation_amd import acc as A, pcdl
    n, m = 1 << lg, 64
    d = n - 1
    c = hal._lib.Context(urs_n=n)
    try:
        pp = orc.make_pp(c.read_bases())
        rng = [0x48414C4F00000006 + lg]
        qs = [A.random_instance(c, rng, d) if i % 4 else _plain_instance(c, 900 + i, d) for i in range(m)]
        t0 = time.perf_counter()
        xis, Us, status = pcdl.succinct_check_batch(c, d, qs)
        t_dev = time.perf_counter() - t0
        assert status == [0] * m
        for i, q in enumerate(qs):
            xr, Ur = orc.pcdl_succinct_check(pp, q[:12], d, q[13:17], q[17:21], q[21:])
            assert xis[i].tolist() == xr.tolist() and Us[i].tolist() == Ur.tolist(), i
        c.set_batch_verify(False)
        t0 = time.perf_counter()
        xis2, Us2, status2 = pcdl.succinct_check_batch(c, d, qs)
        t_host = time.perf_counter() - t0
        assert xis2.tolist() == xis.tolist() and Us2.tolist() == Us.tolist()
        # tamper: v of instance 7, c of instance 20, swap L_0 and R_0 of instance 33 (valid points, wrong relation)
        bad = [q.copy() for q in qs]
        bad[7][17] ^= 1
        bad[20][21 + 2 + 24 * lg + 12] ^= 1
        bad[33][23:35], bad[33][23 + 12 * lg: 35 + 12 * lg] = qs[33][23 + 12 * lg: 35 + 12 * lg].copy(), qs[33][23:35].copy()
        want = [0] * m
        for i in (7, 20, 33):
            want[i] = hal._lib.HALO_E_REJECT
            with pytest.raises(ValueError):
                orc.pcdl_succinct_check(pp, bad[i][:12], d, bad[i][13:17], bad[i][17:21], bad[i][21:])
        for on in (True, False):
            c.set_batch_verify(on)
            with pytest.raises(hal._lib.HaloReject) as e:
                pcdl.succinct_check_batch(c, d, bad)
            assert e.value.args[1] == want and "instance 7" in e.value.args[0]
        c.set_batch_verify(True)
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "batch_verify_lg%d.json" % lg), "w") as f:
            json.dump({"config": "64 succinct checks, n=2^%d" % lg, "device_ms_each": t_dev / m * 1e3, "host_pool_ms_each": t_host / m * 1e3}, f)
    finally:
        c.close()


def test_acc_with_64_instances_matches_oracle(hal, ctx, pp):
    """acc::prover / verifier over m = 64 instances at n = 1024 (the common subroutine takes the device-batched succinct
    checks): accumulator equal to the oracle's, accepted by both verifiers, by the decider, and with the host path."""
    from halo_accumulation_amd import acc as A
    n, m = 1024, 64
    d = n - 1
    rng, seed = [31], 31
    qs = []
    for _ in range(m):
        q = A.random_instance(ctx, rng, d)
        q_ref, seed = orc.random_instance(pp, seed, d)
        assert q.tolist() == q_ref.tolist()
        qs.append(q)
    acc = A.prover(ctx, rng, d, qs)
    acc_ref, seed = orc.acc_prover(pp, seed, d, qs)
    assert acc.tolist() == acc_ref.tolist() and rng[0] == seed
    A.verifier(ctx, d, qs, acc)
    orc.acc_verifier(pp, d, qs, acc)
    A.decider(ctx, acc)
    ctx.set_batch_verify(False)
    try:
        A.verifier(ctx, d, qs, acc)
        assert A.prover(ctx, [rng[0] - 0], d, qs) is not None
    finally:
        ctx.set_batch_verify(True)
    bad = [q.copy() for q in qs]; bad[40][17] ^= 1
    with pytest.raises(ValueError):
        A.verifier(ctx, d, bad, acc)


def test_open_2_17_both_strategies_agree(hal):
    """n = 2^17: one real fold round then no-fold rounds, against folding all the way; the
    verifier (succinct check + U == commit(h)) accepts and the proofs are identical."""
    from halo_accumulation_amd import pcdl
    n = 1 << 17
    d = n - 1
    c = hal._lib.Context(urs_n=n)
    try:
        coeffs, s = orc.rng_scalars(0x48414C4F00000003, n)
        zw, _ = orc.rng_scalars(s, 2)
        C = pcdl.commit(c, coeffs, d, zw[1])
        proofs = []
        for switch, levels in ((1 << 16, 2), (0, 1), (1 << 10, 1), (1 << 10, 2), (1 << 16, 1)):
            c.set_ipa_switch(switch)
            c.set_fold_levels(levels)
            proofs.append(pcdl.open(c, [5], coeffs, C, d, zw[0], zw[1]))
        assert all(p.tolist() == proofs[0].tolist() for p in proofs)
        pcdl.check_proof(c, C, d, zw[0], c.poly_eval(coeffs, zw[0]), proofs[0])
    finally:
        c.close()


def test_acc_chain_2_20_64_instances(hal):
    """BASELINE config 4 at full size, the whole shape of benches/acc.rs:64-98 with k = 64: 64 x (random_instance +
    prover([prev_acc.into(), q])), then 64 x verifier + 1 x decider (the "fast" check), plus a few deciders of
    earlier accumulators (the "slow" check, benches/acc.rs:100-106) and rejection of tampered inputs."""
    import json, os, time
    from halo_accumulation_amd import acc as A
    n, K = 1 << 20, 64
    d = n - 1
    c = hal._lib.Context(urs_n=n)
    try:
        rng = [0x48414C4F00000004]
        accs, qss, acc = [], [], None
        t0 = time.perf_counter()
        for _ in range(K):
            q = A.random_instance(c, rng, d)
            qs = [q] if acc is None else [A.instance_from_accumulator(c, acc, d), q]
            acc = A.prover(c, rng, d, qs)
            accs.append(acc); qss.append(qs)
        t_chain = time.perf_counter() - t0
        t0 = time.perf_counter()
        for a, qs in zip(accs, qss):
            A.verifier(c, d, qs, a)
        t_ver = time.perf_counter() - t0
        t0 = time.perf_counter()
        A.decider(c, accs[-1])
        t_dec = time.perf_counter() - t0
        for a in accs[:3]:
            A.decider(c, a)
        bad = accs[-1].copy(); bad[13] ^= 1  # z
        with pytest.raises(ValueError):
            A.decider(c, bad)
        with pytest.raises(ValueError):
            A.verifier(c, d, qss[-1], accs[-2])  # an accumulator of other instances
        # an instance that claims another degree must be rejected before it is parsed (acc.rs:169)
        q_bad = qss[-1][1].copy(); q_bad[12] = 2 * n - 1; q_bad[22] = 21
        with pytest.raises(ValueError):
            A.verifier(c, d, [qss[-1][0], q_bad], accs[-1])
        # a proof point off the curve is rejected, not computed with
        q_bad = qss[-1][1].copy(); q_bad[21 + 2] ^= 1  # X of L_0
        with pytest.raises(ValueError):
            A.verifier(c, d, [qss[-1][0], q_bad], accs[-1])
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "asdl64_test_timing.json"), "w") as f:
            json.dump({"config": "ASDL 64 accumulated instances n=2^20 (tests/test_gpu_pcdl_acc.py)", "prover_chain_s": t_chain,
                       "verifier_ms_each": t_ver / K * 1e3, "decider_ms": t_dec * 1e3}, f)
    finally:
        c.close()


def test_open_2_19_matches_oracle_fixture(hal):
    """pcdl::open at n = 2^19 against the oracle's proof (tests/golden/open_2_19.json, generated in the build
    container by tests/golden/make_open_fixture.py): the first fold of this size runs k_fold_points with two
    points per lane sharing one inversion (m = 2^18, ipa.hip ipa_fold_points) -- compared bit for bit with the
    CPU restatement's double-and-add fold (pcdl.rs:216-224), hiding and non-hiding."""
    import hashlib, json, os
    from halo_accumulation_amd import pcdl
    with open(os.path.join(ROOT, "tests", "golden", "open_2_19.json")) as f:
        fx = json.load(f)
    lg = fx["lg_n"]
    n = 1 << lg
    d = n - 1
    words = lambda h: np.array([int(x, 16) for x in h], dtype=np.uint64)
    c = hal._lib.Context(urs_n=n)
    try:
        coeffs, s = orc.rng_scalars(fx["coeff_seed"], fx["deg"] + 1)
        zw, _ = orc.rng_scalars(s, 2)
        for name in ("plain", "hiding"):
            case = fx["cases"][name]
            w = zw[1] if case["hiding"] else None
            C = pcdl.commit(c, coeffs, d, w)
            assert C.tolist() == words(case["C"]).tolist()
            rng = [fx["open_seed"]]
            pi = pcdl.open(c, rng, coeffs, C, d, zw[0], w)
            o = 2 + 24 * lg
            assert pi[2:14].tolist() == words(case["L0"]).tolist(), "L of the first round"
            assert pi[2 + 12 * lg: 14 + 12 * lg].tolist() == words(case["R0"]).tolist(), "R of the first round"
            assert pi[14:26].tolist() == words(case["L1"]).tolist(), "L of the second round (after the two-per-lane fold)"
            assert pi[2 + 12 * (lg - 1): 2 + 12 * lg].tolist() == words(case["L_last"]).tolist()
            assert pi[o: o + 12].tolist() == words(case["U"]).tolist() and pi[o + 12: o + 16].tolist() == words(case["c"]).tolist()
            assert hashlib.sha256(pi.tobytes()).hexdigest() == case["proof_sha256"]
            assert rng[0] == int(case["rng_state_after"], 16)
            v = c.poly_eval(coeffs, zw[0])
            assert v.tolist() == words(case["v"]).tolist()
            pcdl.check_proof(c, C, d, zw[0], v, pi)
    finally:
        c.close()


def _open_fixture_cases(hal, fx, c, coeffs, zw, label):
    """one context configuration against every case of an open fixture (tests/golden/open_2_<lg>.json)"""
    import hashlib
    from halo_accumulation_amd import pcdl
    lg = fx["lg_n"]
    d = (1 << lg) - 1
    words = lambda h: np.array([int(x, 16) for x in h], dtype=np.uint64)
    for name in ("plain", "hiding"):
        case = fx["cases"][name]
        w = zw[1] if case["hiding"] else None
        C = pcdl.commit(c, coeffs, d, w)
        assert C.tolist() == words(case["C"]).tolist(), label
        rng = [fx["open_seed"]]
        pi = pcdl.open(c, rng, coeffs, C, d, zw[0], w)
        o = 2 + 24 * lg
        assert pi[2:14].tolist() == words(case["L0"]).tolist(), label + ": L of the first round"
        assert pi[2 + 12 * lg: 14 + 12 * lg].tolist() == words(case["R0"]).tolist(), label + ": R of the first round"
        assert pi[14:26].tolist() == words(case["L1"]).tolist(), label + ": L of the second round"
        assert pi[2 + 12 * (lg - 1): 2 + 12 * lg].tolist() == words(case["L_last"]).tolist(), label
        assert pi[o: o + 12].tolist() == words(case["U"]).tolist() and pi[o + 12: o + 16].tolist() == words(case["c"]).tolist(), label
        assert hashlib.sha256(pi.tobytes()).hexdigest() == case["proof_sha256"], label + ": the whole proof"
        assert rng[0] == int(case["rng_state_after"], 16)
        v = c.poly_eval(coeffs, zw[0])
        assert v.tolist() == words(case["v"]).tolist()
        pcdl.check_proof(c, C, d, zw[0], v, pi)


def test_open_2_20_matches_oracle_fixture(hal):
    """BASELINE config 3 at its own size against the ORACLE's proof (tests/golden/open_2_20.json: orc_pcdl_open, 4.5 minutes of CPU
    per proof in the build container, tests/golden/make_open_fixture.py 20 --acc) -- pcdl.rs:120-242.  n = 2^20 is the only
    size at which an open takes the tagged L/R launch over the c = 20 table (msm.hip MsmBatch::tagged, ipa.hip
    k_nofold_expand_tagged), the c = 20 table plan inside an open and (from the K-th open on) the comb-table fold.  Compared,
    hiding and not: (a) the default configuration as a fresh context runs it, (b) with the fold table forced in, (c) with
    neither table (halo_set_table_mode(0), halo_set_fold_table(0): general pipeline, Straus fold)."""
    import json, os
    with open(os.path.join(ROOT, "tests", "golden", "open_2_20.json")) as f:
        fx = json.load(f)
    assert fx["lg_n"] == 20
    n = 1 << 20
    c = hal._lib.Context(urs_n=n)
    try:
        coeffs, s = orc.rng_scalars(fx["coeff_seed"], fx["deg"] + 1)
        zw, _ = orc.rng_scalars(s, 2)
        _open_fixture_cases(hal, fx, c, coeffs, zw, "default configuration")
        assert c.info(0) > 0, "the c = 20 table is in place: the opens above took the tagged launch"
        c.set_fold_table(1)
        _open_fixture_cases(hal, fx, c, coeffs, zw, "fold table forced")
        assert c.info(1) > 0, "the fold table was built"
        c.set_fold_table(0)
        c.set_table_mode(0)
        assert c.info(0) == 0 and c.info(1) == 0
        _open_fixture_cases(hal, fx, c, coeffs, zw, "no tables")
    finally:
        c.close()


def test_acc_prover_2_20_matches_oracle_fixture(hal):
    """One acc::prover step (acc.rs:190-220) over two random instances (benches/acc.rs:15-29) at n = 2^20 against the oracle's
    accumulator (tests/golden/open_2_20.json "acc": ~13 minutes of CPU): both instances and the accumulator blob for blob
    (SHA-256), RNG states included; verifier and decider accept it."""
    import hashlib, json, os
    from halo_accumulation_amd import acc as A
    with open(os.path.join(ROOT, "tests", "golden", "open_2_20.json")) as f:
        fx = json.load(f)
    a = fx["acc"]
    n = 1 << fx["lg_n"]
    d = n - 1
    words = lambda h: np.array([int(x, 16) for x in h], dtype=np.uint64)
    c = hal._lib.Context(urs_n=n)
    try:
        qs = []
        for k in range(2):
            rng = [int(a["q_seeds"][k], 16)]
            q = A.random_instance(c, rng, d)
            assert q[:21].tolist() == words(a["q_head"][k]).tolist(), "C, d, z, v of instance %d" % k
            assert hashlib.sha256(q.tobytes()).hexdigest() == a["q_sha256"][k], "instance %d" % k
            assert rng[0] == int(a["q_rng_state_after"][k], 16)
            qs.append(q)
        rng = [int(a["acc_seed"], 16)]
        acc = A.prover(c, rng, d, qs)
        iw = 21 + 2 + 24 * fx["lg_n"] + 32
        assert acc[:21].tolist() == words(a["acc_head"]).tolist(), "C_bar, d, z, v of the accumulator"
        assert acc[iw:].tolist() == words(a["acc_tail"]).tolist(), "h0, U0, w of the accumulator"
        assert hashlib.sha256(acc.tobytes()).hexdigest() == a["acc_sha256"]
        assert rng[0] == int(a["acc_rng_state_after"], 16)
        A.verifier(c, d, qs, acc)
        A.decider(c, acc)
    finally:
        c.close()


@pytest.mark.parametrize("lg", [6, 9, 12, 15])
def test_open_with_fold_table_matches_oracle(hal, lg):
    """The first fold of an open from the comb table over the context's key (foldtab.hip: 64 entries added up per scalar, no
    doubling chain) against the CPU restatement's double-and-add fold (pcdl.rs:216-224): proofs equal the oracle's and the
    generic kernel's, hiding and not; built on demand, released on request."""
    from halo_accumulation_amd import pcdl
    n, d = 1 << lg, (1 << lg) - 1
    c = hal._lib.Context(urs_n=n)
    try:
        if lg <= 14:
            c.set_ipa_switch(4)  # (keys up to 2^14 points are not folded at all by default: fold them, two rounds at a time)
        gs = c.read_bases()
        ppl = orc.make_pp(gs) if lg <= 12 else None  # (the oracle's open is seconds per proof above 2^12)
        coeffs, s = orc.rng_scalars(0xF01D + lg, n - 1)
        zw, _ = orc.rng_scalars(s, 2)
        for hiding in (False, True):
            w = zw[1] if hiding else None
            C = pcdl.commit(c, coeffs, d, w)
            c.set_fold_table(0)
            want = pcdl.open(c, [77], coeffs, C, d, zw[0], w)
            assert c.info(1) == 0
            c.set_fold_table(1)
            got = pcdl.open(c, [77], coeffs, C, d, zw[0], w)
            assert c.info(1) == 704 * 64 * (n - n // 4), "the table covers the upper three quarters of the key"
            assert got.tolist() == want.tolist()
            assert pcdl.open(c, [77], coeffs, C, d, zw[0], w).tolist() == want.tolist()  # (table already there)
            if ppl is not None:
                ref, _ = orc.pcdl_open(ppl, 77, coeffs, C, d, zw[0], w)
                assert got.tolist() == ref.tolist()
            pcdl.check_proof(c, C, d, zw[0], c.poly_eval(coeffs, zw[0]), got)
        # default mode: the second full-size open builds it (contexts of >= 2^18 points only: not this one)
        c.set_fold_table(0)
        c.set_fold_table(-1)
        pcdl.open(c, [77], coeffs, C, d, zw[0], w); pcdl.open(c, [77], coeffs, C, d, zw[0], w)
        assert c.info(1) == 0
        # an open over a prefix of the key does not use (or build) the table
        if lg >= 9:
            c.set_fold_table(1)
            d2 = n // 2 - 1
            C2 = pcdl.commit(c, coeffs[:d2], d2)
            p2 = pcdl.open(c, [5], coeffs[:d2], C2, d2, zw[0])
            assert c.info(1) == 0
            pcdl.check_proof(c, C2, d2, zw[0], c.poly_eval(coeffs[:d2], zw[0]), p2)
    finally:
        c.close()


def test_open_2_20_first_rounds_as_one_tagged_launch_equal_two_plain_launches(hal):
    """Rounds 0 and 1 of an open over the full 2^20-point key: L and R from ONE launch sequence over the fixed-base table (one
    scalar array, bit 255 of an element = its bucket set: msm.hip MsmBatch::tagged, ipa.hip k_nofold_expand_tagged) against the
    same open with the table switched off (two plain launches through the table-free pipeline): proofs equal word for word,
    for a dense, a hiding, a half-empty (zero scalars carry a tag too) and a constant-coefficient polynomial (every point of a
    set in ONE bucket per window: the oversized-run and many-task paths of the sort with two sets); all verify."""
    import torch
    from halo_accumulation_amd import pcdl
    n = 1 << 20; d = n - 1
    c = hal._lib.Context(urs_n=n)
    try:
        c.set_fold_table(0)  # (not what is tested here; saves its 35 GB and 0.2 s)
        buf = torch.empty((n + 2) * 4, dtype=torch.int64, device="cuda")
        c.rng_scalars_dev(0x7A66ED, n + 2, buf.data_ptr())
        co = np.ascontiguousarray(buf.cpu().numpy().view(np.uint64).reshape(n + 2, 4))
        coeffs, zw = np.ascontiguousarray(co[:n]), co[n:]
        sparse = coeffs.copy()
        sparse[: n // 2] = 0
        sparse[n // 2 + 5 :: 7] = 0
        flat = np.ascontiguousarray(np.broadcast_to(coeffs[5], coeffs.shape))  # one value: 2^19 points per bucket and set in round 0
        for poly, w in ((coeffs, None), (coeffs, zw[1]), (sparse, None), (flat, None)):
            c.set_table_mode(-1)
            C = pcdl.commit(c, poly, d, w)  # (the first MSM over the key builds the table)
            c.prof_enable(1); c.prof_reset()
            got = pcdl.open(c, [11], poly, C, d, zw[0], w)
            ran = c.prof()
            c.prof_enable(0)
            assert ran["k_nofold_expand_tagged"][1] == 2 and ran["k_tmsm_coarse_scatter2"][1] == 2, "rounds 0 and 1 did not take the tagged launch"
            assert "k_nofold_expand" in ran and ran["k_tmsm_recode"][1] == (2 if w is None else 3), "only rounds 0 and 1 (and a hiding open's commitment) run over the table"
            assert pcdl.open(c, [11], poly, C, d, zw[0], w).tolist() == got.tolist()  # (replayed launch graphs)
            c.set_table_mode(0)
            c.prof_enable(1); c.prof_reset()
            want = pcdl.open(c, [11], poly, C, d, zw[0], w)
            ran = c.prof()
            c.prof_enable(0)
            assert ran.get("k_nofold_expand_tagged", (0, 0))[1] == 0 and ran.get("k_tmsm_recode", (0, 0))[1] == 0
            assert got.tolist() == want.tolist()
            pcdl.check_proof(c, C, d, zw[0], c.poly_eval(poly, zw[0]), got)
    finally:
        c.close()


def test_open_over_the_first_half_of_a_2_21_point_key_equals_the_2_20_point_context(hal):
    """The generators of a context are a prefix-consistent sequence (main.rs:18-32), so an open of a degree-(2^20 - 1) polynomial
    over the first half of a 2^21-point key must be the open of a 2^20-point context, word for word -- through the larger
    context's table (built for 2^21 points, indexed with its own stride) and the tagged launch of rounds 0 and 1 over a prefix."""
    import torch
    from halo_accumulation_amd import pcdl
    n = 1 << 20; d = n - 1
    big, small = hal._lib.Context(urs_n=2 * n), hal._lib.Context(urs_n=n)
    try:
        for c in (big, small):
            c.set_fold_table(0)
        buf = torch.empty((n + 2) * 4, dtype=torch.int64, device="cuda")
        small.rng_scalars_dev(0x21F0, n + 2, buf.data_ptr())
        co = np.ascontiguousarray(buf.cpu().numpy().view(np.uint64).reshape(n + 2, 4))
        coeffs, zw = np.ascontiguousarray(co[:n]), co[n:]
        for w in (None, zw[1]):
            Cs, Cb = pcdl.commit(small, coeffs, d, w), pcdl.commit(big, coeffs, d, w)
            assert Cs.tolist() == Cb.tolist()
            want = pcdl.open(small, [3], coeffs, Cs, d, zw[0], w)
            big.prof_enable(1); big.prof_reset()
            got = pcdl.open(big, [3], coeffs, Cb, d, zw[0], w)
            ran = big.prof()
            big.prof_enable(0)
            assert ran.get("k_nofold_expand_tagged", (0, 0))[1] == 2, "rounds 0 and 1 over the prefix did not take the tagged launch"
            assert got.tolist() == want.tolist()
            pcdl.check_proof(big, Cb, d, zw[0], small.poly_eval(coeffs, zw[0]), got)
    finally:
        big.close(); small.close()


def test_msm_affine_more_generators_than_the_context_holds(hal, ctx):
    """halo_msm_affine over more generators than the context has points: consecutive chunks, partial sums added on the host"""
    from halo_accumulation_amd import group
    m = 3 * 4096 + 77
    Gs = orc.urs_affine(50000, m)
    ms, _ = orc.rng_scalars(0xC0FFEE, m)
    assert group.point_dot_affine(ctx, ms, Gs=Gs).tolist() == orc.msm_affine(Gs, ms).tolist()


def test_open_with_fold_table_and_an_infinity_in_the_key(hal):
    """a key that holds the point at infinity (halo_ctx_create takes any bases): its table entries are infinity, the folds
    skip them, proofs equal the generic kernel's and the oracle's"""
    from halo_accumulation_amd import pcdl
    n, d = 256, 255
    gs = orc.urs_affine(2, n).copy()
    gs[200] = 0
    gs[77] = 0
    c = hal._lib.Context(gs)
    try:
        c.set_ipa_switch(4)
        ppl = orc.make_pp(gs)
        coeffs, s = orc.rng_scalars(0x1F, n - 1)
        zw, _ = orc.rng_scalars(s, 2)
        for w in (None, zw[1]):
            C = pcdl.commit(c, coeffs, d, w)
            assert C.tolist() == orc.pcdl_commit(ppl, coeffs, d, w).tolist()
            c.set_fold_table(0)
            want = pcdl.open(c, [3], coeffs, C, d, zw[0], w)
            c.set_fold_table(1)
            got = pcdl.open(c, [3], coeffs, C, d, zw[0], w)
            ref, _ = orc.pcdl_open(ppl, 3, coeffs, C, d, zw[0], w)
            assert got.tolist() == want.tolist() == ref.tolist()
    finally:
        c.close()


def test_fold_table_memory_failure_falls_back(hal, monkeypatch):
    """no memory for the comb table: one line on stderr, the generic fold kernel runs, same proof, no attempt per open -- and a
    new attempt after the back-off, which succeeds once the memory is there"""
    from halo_accumulation_amd import pcdl
    n, d = 1 << 15, (1 << 15) - 1
    c = hal._lib.Context(urs_n=n)
    try:
        coeffs, s = orc.rng_scalars(0xFA11, n - 3)
        zw, _ = orc.rng_scalars(s, 1)
        C = pcdl.commit(c, coeffs, d)
        c.set_fold_table(0)
        want = pcdl.open(c, [9], coeffs, C, d, zw[0])
        hal._lib.dev_hook("table_fail", 1)
        c.set_fold_table(1)
        for _ in range(2):
            assert pcdl.open(c, [9], coeffs, C, d, zw[0]).tolist() == want.tolist()
        assert c.info(1) == 0 and c.info(5) == 4, "allocation failed: status 4, tried again later"
        # no latch.  The memory is back (the hook is gone): the table is tried again after the back-off -- eight more
        # opens -- without any call from the caller, and nothing it held during the failed attempts stayed on the budget's books
        hal._lib.dev_hook("table_fail", 0)
        used = c.info(4)
        built_after = None
        for k in range(12):
            assert pcdl.open(c, [9], coeffs, C, d, zw[0]).tolist() == want.tolist()
            if c.info(1):
                built_after = k + 1
                break
        assert built_after is not None and 2 <= built_after <= 9, built_after
        assert c.info(5) == 2 and c.info(4) == used + c.info(1)
        assert pcdl.open(c, [9], coeffs, C, d, zw[0]).tolist() == want.tolist()  # through the table
    finally:
        c.close()


def test_open_2_19_fold_table_matches_oracle_fixture(hal):
    """the same fixture as above with the comb table in use from the first open on"""
    import hashlib, json, os
    from halo_accumulation_amd import pcdl
    with open(os.path.join(ROOT, "tests", "golden", "open_2_19.json")) as f:
        fx = json.load(f)
    lg = fx["lg_n"]
    n, d = 1 << lg, (1 << lg) - 1
    c = hal._lib.Context(urs_n=n)
    try:
        c.set_fold_table(1)
        coeffs, s = orc.rng_scalars(fx["coeff_seed"], fx["deg"] + 1)
        zw, _ = orc.rng_scalars(s, 2)
        for name in ("plain", "hiding"):
            case = fx["cases"][name]
            w = zw[1] if case["hiding"] else None
            C = pcdl.commit(c, coeffs, d, w)
            pi = pcdl.open(c, [fx["open_seed"]], coeffs, C, d, zw[0], w)
            assert hashlib.sha256(pi.tobytes()).hexdigest() == case["proof_sha256"]
        assert c.info(1) == 704 * 64 * (n - n // 4) and c.info(2) > 0
    finally:
        c.close()


# ------------------------------------------------------------------ sharded open
def _sharded_worker(rank, world, port, n, q):
    import os, sys
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import halo_accumulation_amd as h
    from halo_accumulation_amd import pcdl
    from halo_accumulation_amd.sharded import ShardedOpen
    import orc

    def allgather(arr):
        t = torch.from_numpy(np.ascontiguousarray(arr).view(np.int64).copy())
        out = torch.empty(world * t.numel(), dtype=torch.int64)
        dist.all_gather_into_tensor(out, t)
        return out.numpy().view(np.uint64).reshape(world, -1)

    coeffs, s = orc.rng_scalars(0x48414C4F00000003, n - 3)   # every rank derives the same polynomial
    z, _ = orc.rng_scalars(s, 1)
    full = np.zeros((n, 4), dtype=np.uint64); full[: n - 3] = coeffs
    so = ShardedOpen(h._lib, rank, world, allgather)
    so.load_key(n)
    wz, _ = orc.rng_scalars(s + 99, 1)
    ok, proofs = True, []
    for w in (None, wz[0]):  # non-hiding, then hiding (pcdl.rs:137-164)
        if rank == 0:
            ref = h._lib.Context(urs_n=n)
            C = pcdl.commit(ref, coeffs, n - 1, w)
            rng_ref = [4242]
            want = pcdl.open(ref, rng_ref, coeffs, C, n - 1, z[0], w)
            Cs = torch.from_numpy(C.view(np.int64).copy())
        else:
            Cs = torch.zeros(12, dtype=torch.int64)
        dist.broadcast(Cs, 0)
        C = Cs.numpy().view(np.uint64)
        rng = [4242]
        proof, v = so.open(np.ascontiguousarray(full[rank::world]), C, z[0], w=w, rng=rng, deg=n - 4)
        rng2 = [4242]
        proof_py, v_py = so.open_by_rounds(np.ascontiguousarray(full[rank::world]), C, z[0], w=w, rng=rng2, deg=n - 4)
        ok = ok and proof_py.tolist() == proof.tolist() and v_py.tolist() == v.tolist() and rng2 == rng
        # pcdl::check against the sharded key: accepted by every rank; a proof with another U is rejected by every rank
        so.check(C, n - 1, z[0], v, proof)
        bad = proof.copy()
        lg = n.bit_length() - 1
        bad[2 + 24 * lg: 14 + 24 * lg] = proof[2:14]  # U := L_0 (a valid point, the wrong one)
        try:
            so.check(C, n - 1, z[0], v, bad)
            ok = False
        except ValueError:
            pass
        if rank == 0:
            ok = ok and proof.tolist() == want.tolist() and rng[0] == rng_ref[0]
            pcdl.check_proof(ref, C, n - 1, z[0], v, proof)
            ref.close()
        proofs.append(proof.tolist())
    q.put((rank, ok, proofs))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 1 << 10), (4, 1 << 17)])
def test_sharded_open_two_and_four_ranks_on_one_gpu(world, n):
    """Ranks share the single GPU of the test box and talk over gloo: the sharded open must return, on
    every rank, the proof the single-GPU pcdl::open returns."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert all(pr == res[0][2] for _, _, pr in res)


def _sharded_fail_worker(rank, world, port, n, q):
    """Rank 1 is made to fail locally at several points of a sharded open / check (the development library's shard_fail hooks, and a Python
    exception in the by-rounds driver): EVERY rank must come back with an error from the same collective -- nobody hangs."""
    import os, sys
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import halo_accumulation_amd as h
    from halo_accumulation_amd.sharded import ShardedOpen
    import orc

    calls = [0]

    def allgather(arr):
        calls[0] += 1
        t = torch.from_numpy(np.ascontiguousarray(arr).view(np.int64).copy())
        out = torch.empty(world * t.numel(), dtype=torch.int64)
        dist.all_gather_into_tensor(out, t)
        return out.numpy().view(np.uint64).reshape(world, -1)

    coeffs, s = orc.rng_scalars(0x48414C4F00000003, n)
    z, _ = orc.rng_scalars(s, 1)
    wz, _ = orc.rng_scalars(s + 99, 1)
    so = ShardedOpen(h._lib, rank, world, allgather)
    so.load_key(n)
    local = np.ascontiguousarray(coeffs[rank::world])
    # the commitment: the sum of the ranks' MSMs (non-hiding)
    C = h._lib.point_sum(allgather(so.ctx.msm(local)))
    proof, v = so.open(local, C, z[0])  # no failure injected: the reference outcome
    so.check(C, n - 1, z[0], v, proof)
    lg_l = (n // world).bit_length() - 1
    outcomes = []
    for step, hiding in ((0, False), (0, True), (1, False), (lg_l // 2, False), (lg_l, True), (lg_l + 1, False)):
        h._lib.dev_hook("shard_fail_rank", 1); h._lib.dev_hook("shard_fail_at", step)  # offset 1 fails before collective `step`
        calls[0] = 0
        try:
            so.open(local, C, z[0], w=wz[0] if hiding else None, rng=[7], deg=n - 1)
            outcomes.append(("open", step, "returned", calls[0]))
        except h._lib.HaloError as e:
            outcomes.append(("open", step, "HaloError", calls[0], "injected" in str(e), "rank 1" in str(e)))
    h._lib.dev_hook("shard_fail_at", -2)  # ... in the check
    calls[0] = 0
    try:
        so.check(C, n - 1, z[0], v, proof)
        outcomes.append(("check", "returned"))
    except h._lib.HaloError as e:
        outcomes.append(("check", "HaloError", calls[0], "injected" in str(e), "rank 1" in str(e)))
    h._lib.dev_hook("reset", 0)
    # the same rule in the by-rounds Python driver: rank 1's third round raises
    if rank == 1:
        orig, seen = h._lib.Ipa.round_lr_partial, [0]

        def flaky(self):
            seen[0] += 1
            if seen[0] == 3:
                raise RuntimeError("local failure on rank 1")
            return orig(self)

        h._lib.Ipa.round_lr_partial = flaky
    calls[0] = 0
    try:
        so.open_by_rounds(local, C, z[0])
        outcomes.append(("by_rounds", "returned"))
    except (RuntimeError, h._lib.HaloError) as e:
        outcomes.append(("by_rounds", type(e).__name__, calls[0]))
    if rank == 1:
        h._lib.Ipa.round_lr_partial = orig
    # and the group is still in step: a clean open afterwards gives the reference proof on every rank
    again, _ = so.open(local, C, z[0])
    q.put((rank, outcomes, again.tolist() == proof.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_open_a_failing_rank_fails_every_rank_at_the_same_collective():
    """VERDICT r3 weak #5 / ADVICE r3: a rank-local failure must not leave the peers in an all-gather.  Two gloo ranks on the
    one GPU; rank 1 fails before the first collective, in the hiding branch, in an early / middle / last round, at the tail,
    in the check, and (by-rounds driver) with a Python exception.  Both ranks must return an error after the SAME number of
    collectives, within the timeout, and a clean open afterwards still works."""
    import socket
    import torch.multiprocessing as mp
    world, n = 2, 1 << 10
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_fail_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = dict((r, (o, ok)) for r, o, ok in [q.get(timeout=240) for _ in range(world)])
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    o0, ok0 = res[0]
    o1, ok1 = res[1]
    assert ok0 and ok1
    lg_l = (n // world).bit_length() - 1
    for k, step in enumerate((0, 0, 1, lg_l // 2, lg_l, lg_l + 1)):
        assert o0[k][:3] == ("open", step, "HaloError") and o1[k][:3] == ("open", step, "HaloError"), (o0[k], o1[k])
        assert o0[k][3] == o1[k][3] == step + 1, "both ranks stop after the same collective"
        assert o1[k][4] and o0[k][5], "the failing rank keeps its own message, the peer names the failing rank"
    assert o0[6][:3] == ("check", "HaloError", 1) and o1[6][:3] == ("check", "HaloError", 1) and o1[6][3] and o0[6][4]
    assert o0[7] == ("by_rounds", "HaloError", 4) and o1[7] == ("by_rounds", "RuntimeError", 4)


def test_sharded_open_world_one_is_plain_open(hal, ctx):
    from halo_accumulation_amd import pcdl
    from halo_accumulation_amd.sharded import ShardedOpen
    n = 512
    coeffs, s = orc.rng_scalars(77, n)
    z, _ = orc.rng_scalars(s, 1)
    C = pcdl.commit(ctx, coeffs, n - 1)
    so = ShardedOpen(hal._lib, 0, 1, None)
    so.load_key(n)
    proof, v = so.open(coeffs, C, z[0])
    assert proof.tolist() == pcdl.open(ctx, [1], coeffs, C, n - 1, z[0]).tolist()
    assert v.tolist() == ctx.poly_eval(coeffs, z[0]).tolist()
    # the same protocol call by call from Python (what halo_pcdl_open_sharded runs inside)
    proof2, v2 = so.open_by_rounds(coeffs, C, z[0])
    assert proof2.tolist() == proof.tolist() and v2.tolist() == v.tolist()
    # hiding branch: same proof and same final rng state as pcdl::open
    w, _ = orc.rng_scalars(5, 1)
    Ch = pcdl.commit(ctx, coeffs, n - 1, w[0])
    r1, r2 = [77], [77]
    proof, _ = so.open(coeffs, Ch, z[0], w=w[0], rng=r1, deg=n - 1)
    assert proof.tolist() == pcdl.open(ctx, r2, coeffs, Ch, n - 1, z[0], w[0]).tolist() and r1 == r2
    so.check(Ch, n - 1, z[0], v, proof)  # world 1: the whole of pcdl::check


def test_native_rccl_allgather_world_one(hal, ctx):
    """libhalo_rccl.so's halo_allgather_rccl -- ncclCommInitRank with ONE rank on this box's GPU -- carries the
    collectives of halo_pcdl_open_sharded / _check_sharded as a C function pointer (no Python frame in the collective path; the
    symbol a Rust host links).  Same proof as halo_pcdl_open, hiding and not; the record it gathers is the record it was given."""
    from halo_accumulation_amd import pcdl, rccl
    from halo_accumulation_amd.sharded import ShardedOpen
    if not rccl.available():
        pytest.skip("libhalo_rccl.so not built (no librccl in this image)")
    g = rccl.RcclGather(rccl.unique_id(), 0, 1, device=0)
    try:
        rec = np.arange(33, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        assert g(rec).tolist() == [rec.tolist()] and g.calls == 1
        n = 512
        coeffs, s = orc.rng_scalars(77, n)
        z, _ = orc.rng_scalars(s, 1)
        w, _ = orc.rng_scalars(5, 1)
        so = ShardedOpen(hal._lib, 0, 1, g, always_collective=True)
        so.load_key(n)
        C = pcdl.commit(ctx, coeffs, n - 1)
        before = g.calls
        proof, v = so.open(coeffs, C, z[0])
        assert g.calls - before == 1 + 9, "the share of p(z), then one record per round (P = 1: no tail collective)"
        assert proof.tolist() == pcdl.open(ctx, [1], coeffs, C, n - 1, z[0]).tolist()
        so.check(C, n - 1, z[0], v, proof)
        Ch = pcdl.commit(ctx, coeffs, n - 1, w[0])
        r1, r2 = [77], [77]
        proof_h, _ = so.open(coeffs, Ch, z[0], w=w[0], rng=r1, deg=n - 1)
        assert proof_h.tolist() == pcdl.open(ctx, r2, coeffs, Ch, n - 1, z[0], w[0]).tolist() and r1 == r2
        bad = proof.copy(); bad[2 + 24 * 9 + 3] ^= 1  # U
        with pytest.raises(ValueError):
            so.check(C, n - 1, z[0], v, bad)
        so.ctx.close()
    finally:
        g.close()


def test_fold_table_automatic_mode_allocates_in_the_background(hal):
    """Default mode on a context of >= 2^18 points: the table costs 64 opens' worth of savings, so a key earns it
    by being opened again and again -- the first 7 full-size opens over a key request, reserve and build NOTHING
    (halo_ctx_info(ctx, 5) == 0, no optional memory on the device's books), the 8th asks for the table's memory on a helper
    thread and takes the generic fold, and the table is built at the first later open that finds the memory there.  Every
    open returns the same proof; closing a context while the request is still pending joins the thread and frees what it got."""
    import time
    import torch
    from halo_accumulation_amd import pcdl
    n = 1 << 18
    d = n - 1
    c = hal._lib.Context(urs_n=n)
    try:
        dv = torch.empty((n + 1) * 4, dtype=torch.int64, device="cuda")
        c.rng_scalars_dev(0xF01D, n + 1, dv.data_ptr())
        z = np.ascontiguousarray(dv[4 * n:].cpu().numpy().view(np.uint64))
        C = pcdl.commit_dev(c, dv.data_ptr(), n, d)
        used0 = c.info(4)
        first = pcdl.open_dev(c, [1], dv.data_ptr(), n, C, d, z)
        for k in range(2, 8):  # opens 2..7: a caller with a handful of opens never pays for, or reserves memory for, the table
            assert c.info(5) == 0 and c.info(1) == 0 and c.info(4) == used0, "open %d: nothing requested, nothing reserved" % (k - 1)
            assert pcdl.open_dev(c, [1], dv.data_ptr(), n, C, d, z).tolist() == first.tolist()
        assert c.info(5) == 0 and c.info(1) == 0 and c.info(4) == used0, "seven opens: still nothing"
        assert pcdl.open_dev(c, [1], dv.data_ptr(), n, C, d, z).tolist() == first.tolist()  # the 8th asks for the memory
        assert c.info(5) in (1, 2) and c.info(4) > used0, "the 8th full-size open requests (and reserves) the table's memory"
        built_at = None
        for k in range(40):
            p = pcdl.open_dev(c, [1], dv.data_ptr(), n, C, d, z)
            assert p.tolist() == first.tolist()
            if c.info(1):
                built_at = k
                break
            time.sleep(0.1)
        assert built_at is not None and c.info(1) == 704 * 64 * (n - n // 4)
        assert pcdl.open_dev(c, [1], dv.data_ptr(), n, C, d, z).tolist() == first.tolist()  # through the table
        c.set_fold_table(0)
        assert c.info(1) == 0
    finally:
        c.close()
    # a context closed (or told to do without) while its request may still be running
    for how in ("close", "mode0", "mode1"):
        c2 = hal._lib.Context(urs_n=n)
        dv = torch.empty((n + 1) * 4, dtype=torch.int64, device="cuda")
        c2.rng_scalars_dev(0xF01D, n + 1, dv.data_ptr())
        z = np.ascontiguousarray(dv[4 * n:].cpu().numpy().view(np.uint64))
        C2 = pcdl.commit_dev(c2, dv.data_ptr(), n, d)
        for _ in range(8):  # (the 8th open starts the request)
            assert pcdl.open_dev(c2, [1], dv.data_ptr(), n, C2, d, z).tolist() == first.tolist()
        if how == "mode0":
            c2.set_fold_table(0)
            assert pcdl.open_dev(c2, [1], dv.data_ptr(), n, C2, d, z).tolist() == first.tolist() and c2.info(1) == 0
        if how == "mode1":  # "at once": waits for the request under way and builds from what it brought
            c2.set_fold_table(1)
            assert pcdl.open_dev(c2, [1], dv.data_ptr(), n, C2, d, z).tolist() == first.tolist() and c2.info(1) > 0
        c2.close()


def test_memory_budget_bounds_the_optional_tables_of_all_contexts_on_a_device(hal):
    """VERDICT r3 #6 / ADVICE r3: the fold table and the MSM table are OPTIONAL memory under a per-device, process-wide budget
    (halo_set_memory_budget; default 1/6 of the device).  Two contexts on the one GPU with room for ONE fold table: the first
    gets it, the second runs the generic fold (status 3 = over the budget, same proof) and never pushes the device's total past
    the budget; raising the budget lets it build at the next open; releasing gives the bytes back; budget 0 = no tables."""
    import torch
    from halo_accumulation_amd import pcdl
    n = 1 << 18
    d = n - 1
    ft = 704 * 64 * (n - n // 4)           # 33 KiB per point of the upper three quarters of the key
    tmp = 704 * 32768 * 200                 # the build's temporary
    msm_tab = 15 * 128 * n                  # small-key MSM table (built by the commit)
    a, b = hal._lib.Context(urs_n=n), hal._lib.Context(urs_n=n)
    orig = a.info(3)
    total = torch.cuda.mem_get_info()[1]
    assert orig == total // 6 or os.environ.get("HALO_MEMORY_BUDGET"), "default: one sixth of the device's memory"
    try:
        used0 = a.info(4)  # what other live contexts of this process hold
        dv = torch.empty((n + 1) * 4, dtype=torch.int64, device="cuda")
        a.rng_scalars_dev(0xB0D6E7, n + 1, dv.data_ptr())
        z = np.ascontiguousarray(dv[4 * n:].cpu().numpy().view(np.uint64))
        budget = used0 + ft + tmp + 2 * msm_tab + (1 << 20)
        a.set_memory_budget(budget)
        assert b.info(3) == budget, "the budget belongs to the device, not to one context"
        a.set_fold_table(1); b.set_fold_table(1)
        Ca = pcdl.commit_dev(a, dv.data_ptr(), n, d)
        pa = pcdl.open_dev(a, [1], dv.data_ptr(), n, Ca, d, z)
        assert a.info(1) == ft and a.info(5) == 2 and a.info(0) == msm_tab and a.info(6) == 2
        assert a.info(4) == used0 + ft + msm_tab, "the temporary is off the books again"
        Cb = pcdl.commit_dev(b, dv.data_ptr(), n, d)
        assert Cb.tolist() == Ca.tolist() and b.info(0) == msm_tab
        pb = pcdl.open_dev(b, [1], dv.data_ptr(), n, Cb, d, z)
        assert pb.tolist() == pa.tolist(), "the generic fold gives the same proof"
        assert b.info(1) == 0 and b.info(5) == 3, "second fold table: over the budget"
        assert a.info(4) == used0 + ft + 2 * msm_tab <= budget
        for _ in range(3):  # refused: not asked again at every open (back-off), still the same proof
            assert pcdl.open_dev(b, [1], dv.data_ptr(), n, Cb, d, z).tolist() == pa.tolist() and b.info(1) == 0
        b.set_memory_budget(budget + ft)  # room for the second table: considered again at once
        assert pcdl.open_dev(b, [1], dv.data_ptr(), n, Cb, d, z).tolist() == pa.tolist()
        assert b.info(1) == ft and b.info(5) == 2 and b.info(4) == used0 + 2 * ft + 2 * msm_tab <= b.info(3)
        a.set_fold_table(0)
        assert a.info(1) == 0 and a.info(5) == 5 and a.info(4) == used0 + ft + 2 * msm_tab
        a.set_table_mode(0)
        assert a.info(0) == 0 and a.info(4) == used0 + ft + msm_tab
        b.close()
        assert a.info(4) == used0, "a destroyed context gives everything back"
        # budget 0: no optional memory at all -- the table-free MSM pipeline, same point
        a.set_memory_budget(0)
        a.set_table_mode(-1)
        assert pcdl.commit_dev(a, dv.data_ptr(), n, d).tolist() == Ca.tolist() and a.info(0) == 0 and a.info(6) == 3
        assert a.info(4) == used0
    finally:
        a.set_memory_budget(orig)
        a.close()
        b.close()


def test_clones_share_the_key_and_its_tables(hal):
    """halo_ctx_clone: contexts over ONE resident key.  The clone gives the same points and proofs; the tables exist once
    (the device's optional-memory books do not move when the clone uses them, whoever built them); two threads open at the
    same time on the two contexts; the original may go first -- the clone keeps working and the last one frees everything."""
    import threading
    import torch
    from halo_accumulation_amd import pcdl
    n = 1 << 18
    d = n - 1
    ft, msm_tab = 704 * 64 * (n - n // 4), 15 * 128 * n
    a = hal._lib.Context(urs_n=n)
    used0 = a.info(4)
    free0 = torch.cuda.mem_get_info()[0]
    b = a.clone()
    try:
        assert b.size == n and b.read_bases(5, 3).tolist() == a.read_bases(5, 3).tolist()
        dv = torch.empty((n + 1) * 4, dtype=torch.int64, device="cuda")
        a.rng_scalars_dev(0xC10E, n + 1, dv.data_ptr())
        z = np.ascontiguousarray(dv[4 * n:].cpu().numpy().view(np.uint64))
        # the CLONE builds the MSM table (first commit) and the fold table (first open, mode 1); the original adopts both
        b.set_fold_table(1); a.set_fold_table(1)
        Cb = pcdl.commit_dev(b, dv.data_ptr(), n, d)
        pb = pcdl.open_dev(b, [1], dv.data_ptr(), n, Cb, d, z)
        assert b.info(0) == msm_tab and b.info(1) == ft and a.info(4) == used0 + ft + msm_tab
        Ca = pcdl.commit_dev(a, dv.data_ptr(), n, d)
        pa = pcdl.open_dev(a, [1], dv.data_ptr(), n, Ca, d, z)
        assert Ca.tolist() == Cb.tolist() and pa.tolist() == pb.tolist()
        assert a.info(0) == msm_tab and a.info(1) == ft and a.info(2) == b.info(2), "adopted, not built again"
        assert a.info(4) == used0 + ft + msm_tab, "one copy of each table on the device's books"
        torch.cuda.synchronize()
        assert free0 - torch.cuda.mem_get_info()[0] < ft + msm_tab + (3 << 30), "the clone costs workspaces and IPA buffers, not a second set of tables"
        v = a.poly_eval(np.ascontiguousarray(dv[: 4 * n].cpu().numpy().view(np.uint64).reshape(n, 4)), z)
        # both at once, one thread each
        out, errs = {}, []

        def work(c, key):
            try:
                for _ in range(3):
                    p = pcdl.open_dev(c, [1], dv.data_ptr(), n, Ca, d, z)
                    pcdl.check_proof(c, Ca, d, z, v, p)
                out[key] = p.tolist()
            except Exception as e:  # noqa: BLE001
                errs.append(repr(e))

        ths = [threading.Thread(target=work, args=(c, k)) for c, k in ((a, "a"), (b, "b"))]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        assert not errs and out["a"] == out["b"] == pa.tolist()
        # one context gives up the fold table: the other keeps it, nothing is freed
        a.set_fold_table(0)
        assert a.info(1) == 0 and b.info(1) == ft and a.info(4) == used0 + ft + msm_tab
        assert pcdl.open_dev(a, [1], dv.data_ptr(), n, Ca, d, z).tolist() == pa.tolist()  # generic fold, same proof
        # the original goes first
        a.close()
        assert pcdl.open_dev(b, [1], dv.data_ptr(), n, Cb, d, z).tolist() == pb.tolist()
        assert b.msm_dev(dv.data_ptr(), n).tolist() == Cb.tolist()
        assert b.info(4) == used0 + ft + msm_tab
        c3 = b.clone()  # a clone of the clone, after the original is gone
        assert c3.msm_dev(dv.data_ptr(), n).tolist() == Cb.tolist() and c3.info(0) == msm_tab
        c3.close()
        used_probe = b.info(4)
        b.close()
        probe = hal._lib.Context(urs_n=64)
        assert used_probe == used0 + ft + msm_tab and probe.info(4) == used0, "the last context over the key frees the tables"
        probe.close()
    finally:
        a.close()
        b.close()
    multi = hal._lib.Context(urs_n=1 << 12, devices=[0, 0])
    with pytest.raises(hal._lib.HaloError):
        multi.clone()
    multi.close()


def test_memory_budget_from_the_environment():
    """HALO_MEMORY_BUDGET replaces the default budget for the whole process (for hosts that cannot call the setter: the Rust
    shim); 0 = no optional memory: the table-free pipeline, same point as the oracle's."""
    import subprocess
    import sys
    code = (
        "import sys; sys.path[:0] = [%r, %r]\n"
        "import numpy as np, halo_accumulation_amd as h, orc\n"
        "n = 1 << 17\n"
        "c = h._lib.Context(urs_n=n)\n"
        "sc, _ = orc.rng_scalars(11, n)\n"
        "got = c.msm(sc)\n"
        "assert got.tolist() == orc.msm_affine(c.read_bases(), sc).tolist()\n"
        "print(c.info(3), c.info(0), c.info(6), c.info(4))\n"
    ) % (ROOT, os.path.join(ROOT, "oracle"))
    for env_val, want_budget, table in (("0", 0, False), ("64M", 64 << 20, False), ("1g", 1 << 30, True)):
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, HALO_MEMORY_BUDGET=env_val))
        assert out.returncode == 0, out.stderr[-1500:]
        budget, tab, status, used = [int(x) for x in out.stdout.split()[-4:]]
        assert budget == want_budget
        if table:
            assert tab == 15 * 128 * (1 << 17) and status == 2 and used == tab
        else:
            assert tab == 0 and status == 3 and used == 0


def test_sharded_entry_points_report_misuse_and_a_failing_collective(hal):
    """halo_pcdl_open_sharded / _check_sharded: a stride that is no power of two, an offset past it, ranks without an
    all-gather, and a collective that fails (a Python exception inside the callback must surface, not unwind the C frame)."""
    from halo_accumulation_amd import pcdl
    from halo_accumulation_amd.sharded import ShardedOpen
    n = 64
    coeffs, s = orc.rng_scalars(31, n)
    z, _ = orc.rng_scalars(s, 1)
    full = hal._lib.Context(urs_n=n)
    C = pcdl.commit(full, coeffs, n - 1)
    proof = pcdl.open(full, [1], coeffs, C, n - 1, z[0])
    v = full.poly_eval(coeffs, z[0])

    def broken(arr):
        raise RuntimeError("fabric down")

    so = ShardedOpen(hal._lib, 1, 2, broken)
    so.load_key(n)
    with pytest.raises(RuntimeError, match="fabric down"):
        so.open(np.ascontiguousarray(coeffs[1::2]), C, z[0])
    with pytest.raises(RuntimeError, match="fabric down"):
        so.check(C, n - 1, z[0], v, proof)
    lib = hal.load()
    out, vv = np.zeros(hal.load().halo_proof_words(6), dtype=np.uint64), np.zeros(4, dtype=np.uint64)
    a = lambda x: hal._lib.ptr(np.ascontiguousarray(x, dtype=np.uint64))
    half = np.ascontiguousarray(coeffs[1::2])
    for stride, offset in ((3, 0), (2, 2), (2, 1)):  # (2, 1) without an all-gather
        rc = lib.halo_pcdl_open_sharded(so.ctx.h, stride, offset, None, a(half), half.shape[0], 0, a(C), n - 1, a(z[0]), None, None, None, a(out), a(vv))
        assert rc != 0
        rc = lib.halo_pcdl_check_sharded(so.ctx.h, stride, offset, a(C), n - 1, a(z[0]), a(v), a(proof), None, None)
        assert rc != 0
    so.ctx.close()
    full.close()


def test_check_partial_shares_add_up_to_the_commitment_of_h(hal):
    """halo_pcdl_check_partial on the P cyclic shards of a key (all on this one GPU): every rank returns the proof's U,
    and the P shares add up to CM.Commit(ck, h) = U (pcdl.rs:338-339) -- P = 1, 2, 8, and P = n (one point per rank)."""
    from halo_accumulation_amd import pcdl
    n = 64
    full = hal._lib.Context(urs_n=n)
    coeffs, s = orc.rng_scalars(404, n)
    z, _ = orc.rng_scalars(s, 1)
    C = pcdl.commit(full, coeffs, n - 1)
    proof = pcdl.open(full, [9], coeffs, C, n - 1, z[0])
    v = full.poly_eval(coeffs, z[0])
    xis, U = pcdl.succinct_check(full, C, n - 1, z[0], v, proof)
    hc = pcdl.HPoly(full, xis).get_poly()
    for P in (1, 2, 8, n):
        parts = []
        for r in range(P):
            sh = hal._lib.Context(urs_n=n // P, first_index=2 + r, stride=P)
            Ur, part = pcdl.check_partial(sh, C, n - 1, z[0], v, proof, P, r)
            assert Ur.tolist() == np.asarray(U).tolist()
            # the share is the MSM of the coefficients r, r + P, ... over the rank's points
            assert part.tolist() == sh.msm(np.ascontiguousarray(hc[r::P])).tolist()
            parts.append(part)
            sh.close()
        assert hal._lib.point_sum(np.stack(parts)).tolist() == np.asarray(U).tolist()
    # argument errors and the size rule: d + 1 may be up to stride * shard size, not more
    sh = hal._lib.Context(urs_n=n // 4, first_index=2, stride=4)
    with pytest.raises(hal._lib.HaloError):
        pcdl.check_partial(sh, C, n - 1, z[0], v, proof, 3, 0)
    with pytest.raises(hal._lib.HaloError):
        pcdl.check_partial(sh, C, n - 1, z[0], v, proof, 4, 4)
    with pytest.raises(ValueError):
        pcdl.check_partial(sh, C, n - 1, z[0], v, proof, 2, 0)  # 2 * 16 points < 64 coefficients: "d was larger than D"
    sh.close()
    full.close()
