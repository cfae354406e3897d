"""GPU parity: the HIP path (through the C ABI) against the oracle on the same seeded inputs.

Bit-exactness is defined on canonical affine coordinates (x, y mod p) or the infinity flag
(SURVEY.md section 7); the library additionally normalises what it writes, so Jacobian
outputs are compared limb-for-limb where the oracle normalises too.
"""
import hashlib
import os

import numpy as np
import pytest

import orc
import pallas_model as pm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hal():
    import halo_accumulation_amd as h
    return h._lib


@pytest.fixture(scope="module")
def ctx16k(hal):
    """The reference's own key size: N = 16384 (consts.rs:23), derived on the GPU."""
    c = hal.Context(urs_n=16384)
    yield c
    c.close()


def canon(j):
    return orc.point_canonical(j)


def unnormalise(jac, lam):
    """(X, Y, Z) -> (lam^2 X, lam^3 Y, lam Z): the same point with Z != 1, as the reference's `Projective`
    values are after any arithmetic (group.rs:18-21 takes them un-normalised, acc.rs:178)."""
    L = orc.lib()
    l2, l3, o = orc.z(4), orc.z(4), np.array(jac, dtype=np.uint64).copy()
    lam = np.ascontiguousarray(lam, dtype=np.uint64)
    L.orc_fq_mul(orc.ptr(lam), orc.ptr(lam), orc.ptr(l2))
    L.orc_fq_mul(orc.ptr(l2), orc.ptr(lam), orc.ptr(l3))
    for off, f in ((0, l2), (4, l3), (8, lam)):
        t = orc.z(4)
        L.orc_fq_mul(orc.ptr(np.ascontiguousarray(o[off:off + 4])), orc.ptr(f), orc.ptr(t))
        o[off:off + 4] = t
    return o


# ------------------------------------------------------------------ K11 + group law
def test_urs_kernel_reproduces_consts_table(ctx16k, kat):
    """All 16,384 GS entries of consts.rs, bit-for-bit, out of the HIP fixed-base kernel."""
    gs = ctx16k.read_bases()
    assert hashlib.sha256(gs.tobytes()).hexdigest() == kat["GS_mont_limbs_sha256"]


def test_field_ops(ctx16k):
    n = 4096
    a, s = orc.rng_scalars(1, n)
    b, _ = orc.rng_scalars(s, n)
    edge = [0, 1, pm.R_ORDER - 1, pm.MONT_R % pm.R_ORDER, 2, pm.R_ORDER - 2, (1 << 254) % pm.R_ORDER]
    for i, e in enumerate(edge):
        a[i] = orc.fr_to_mont(e)
        b[(i + 3) % len(edge)] = orc.fr_to_mont(e)
    for field, mod in ((1, pm.R_ORDER), (0, pm.P)):
        # the same limb patterns are valid elements of both fields (all < 2^254 + small): reduce for Fq
        A = [int.from_bytes(x.tobytes(), "little") % mod for x in a]
        B = [int.from_bytes(x.tobytes(), "little") % mod for x in b]
        am = np.array([[(v >> (64 * k)) & (2**64 - 1) for k in range(4)] for v in A], dtype=np.uint64)
        bm = np.array([[(v >> (64 * k)) & (2**64 - 1) for k in range(4)] for v in B], dtype=np.uint64)
        rinv = pm.inv_mod(pm.MONT_R, mod)
        ints = lambda arr: [int.from_bytes(x.tobytes(), "little") for x in arr]
        assert ints(ctx16k.field_op(field, 0, am, bm)) == [x * y * rinv % mod for x, y in zip(A, B)]
        assert ints(ctx16k.field_op(field, 1, am, bm)) == [(x + y) % mod for x, y in zip(A, B)]
        assert ints(ctx16k.field_op(field, 2, am, bm)) == [(x - y) % mod for x, y in zip(A, B)]
        assert ints(ctx16k.field_op(field, 4, am)) == [x * rinv % mod for x in A]
        assert ints(ctx16k.field_op(field, 5, am)) == [x * pm.MONT_R % mod for x in A]
        inv = ints(ctx16k.field_op(field, 3, am[:256]))
        # Montgomery inverse: (xR)^-1 R^2 ... the kernel returns a^(p-2) in Montgomery arithmetic = a^-1 * R^2 / R
        for x, got in zip(A[:256], inv):
            if x == 0:
                assert got == 0
            else:
                assert got * x * rinv % mod == pm.MONT_R % mod  # got (*) x == one


def test_native_radix29_field(ctx16k):
    """The lazy 9 x 29-bit Fq of fq29.hpp against big-int arithmetic, through arkworks-form I/O."""
    n = 4096
    a, s = orc.rng_scalars(3, n)
    b, _ = orc.rng_scalars(s, n)
    mod = pm.P
    A = [int.from_bytes(x.tobytes(), "little") % mod for x in a]
    B = [int.from_bytes(x.tobytes(), "little") % mod for x in b]
    edge = [0, 1, mod - 1, pm.MONT_R % mod, 2, mod - 2, (1 << 254) % mod, (1 << 29) - 1, 1 << 29, (1 << 232) + 5]
    for i, e in enumerate(edge):
        A[i] = e
        B[(i + 3) % len(edge)] = e
    tolimbs = lambda vals: np.array([[(v >> (64 * k)) & (2**64 - 1) for k in range(4)] for v in vals], dtype=np.uint64)
    am, bm = tolimbs(A), tolimbs(B)
    rinv = pm.inv_mod(pm.MONT_R, mod)
    ints = lambda arr: [int.from_bytes(x.tobytes(), "little") for x in arr]
    assert ints(ctx16k.field_op(2, 9, am, bm)) == A                                              # round trip
    assert ints(ctx16k.field_op(2, 0, am, bm)) == [x * y * rinv % mod for x, y in zip(A, B)]    # mul
    assert ints(ctx16k.field_op(2, 6, am, bm)) == [x * x * rinv % mod for x in A]               # sqr
    assert ints(ctx16k.field_op(2, 1, am, bm)) == [(x + y) % mod for x, y in zip(A, B)]
    assert ints(ctx16k.field_op(2, 2, am, bm)) == [(x - y) % mod for x, y in zip(A, B)]
    assert ints(ctx16k.field_op(2, 7, am, bm)) == [12 * (x + y) % mod for x, y in zip(A, B)]
    assert ints(ctx16k.field_op(2, 8, am, bm)) == [(2 * x - y) % mod for x, y in zip(A, B)]
    inv = ints(ctx16k.field_op(2, 3, am[:256], bm[:256]))
    for x, got in zip(A[:256], inv):
        assert got == (0 if x == 0 else pm.inv_mod(x, mod) * pm.MONT_R * pm.MONT_R % mod)


def test_point_ops(ctx16k, urs4096):
    n = 512
    ks, s = orc.rng_scalars(77, n)
    # a_i = k_i * G_i (Jacobian, un-normalised on purpose: scale by lambda^2, lambda^3)
    a = np.zeros((n, 12), dtype=np.uint64)
    b = np.zeros((n, 12), dtype=np.uint64)
    lams, _ = orc.rng_scalars(s + 1, 2 * n)  # any limb pattern < 2^254 + small is a valid Fq Montgomery element too
    lams[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
    for i in range(n):
        orc.lib().orc_affine_to_jac(orc.ptr(urs4096[i]), orc.ptr(a[i]))
        orc.lib().orc_affine_to_jac(orc.ptr(urs4096[i + n]), orc.ptr(b[i]))
        if i % 4 != 3:  # three in four inputs have Z != 1 (every fourth keeps the Z = 1 fast paths covered)
            a[i] = unnormalise(a[i], lams[i])
            b[i] = unnormalise(b[i], lams[n + i])
    # special cases: P + P (two different representatives of the same point), P + (-P), inf + P, P + inf
    b[0] = unnormalise(a[0], lams[5])
    neg = orc.z(12); orc.lib().orc_point_mul(orc.ptr(a[1]), orc.ptr(orc.fr_to_mont(pm.R_ORDER - 1)), orc.ptr(neg)); b[1] = neg
    inf = np.array([1, 0, 0, 0] * 0 + list(a[2][:8]) + [0, 0, 0, 0], dtype=np.uint64)
    a[2] = inf
    b[3] = inf
    got = ctx16k.point_op(0, a, b)
    for i in range(n):
        want = orc.z(12); orc.lib().orc_point_add(orc.ptr(a[i]), orc.ptr(b[i]), orc.ptr(want))
        assert canon(got[i]) == canon(want), i
    assert canon(got[1]) is None and canon(got[2]) == canon(b[2]) and canon(got[3]) == canon(a[3])
    # mixed add (XYZZ and Jacobian forms must agree inside the kernel) incl. P + P and P + (-P)
    baff = np.ascontiguousarray(urs4096[n:2 * n].copy())
    baff[0] = urs4096[0]
    baff[1] = urs4096[1]; baff[1, 4:] = np.array([(pm.P - orc.fq_from_mont(urs4096[1, 4:])) * pm.MONT_R % pm.P >> (64 * k) & (2**64 - 1) for k in range(4)], dtype=np.uint64)
    baff[4] = 0  # affine infinity
    a2 = a.copy(); a2[2] = inf
    got = ctx16k.point_op(1, a2, baff)
    for i in range(n):
        bj = orc.z(12); orc.lib().orc_affine_to_jac(orc.ptr(baff[i]), orc.ptr(bj))
        want = orc.z(12); orc.lib().orc_point_add(orc.ptr(a2[i]), orc.ptr(bj), orc.ptr(want))
        assert canon(got[i]) == canon(want), i
    got = ctx16k.point_op(2, a2)
    for i in range(n):
        want = orc.z(12); orc.lib().orc_point_add(orc.ptr(a2[i]), orc.ptr(a2[i]), orc.ptr(want))
        assert canon(got[i]) == canon(want), i
    ks[5] = 0; ks[6] = orc.fr_to_mont(1); ks[7] = orc.fr_to_mont(pm.R_ORDER - 1)
    got = ctx16k.point_op(3, a, ks)
    for i in range(64):
        want = orc.z(12); orc.lib().orc_point_mul(orc.ptr(a[i]), orc.ptr(ks[i]), orc.ptr(want))
        assert canon(got[i]) == canon(want), i


def test_quad_parallel_point_ops(ctx16k, urs4096):
    """curve_quad.hpp: one XYZZ addition / doubling shared by the 4 lanes of a quad (DPP broadcasts), against the oracle.
    op 4 = a + b, op 5 = 2a, op 6 = a + b with every fourth pair replaced by (a, a), so that general additions and the
    wave-wide doubling branch mix inside one wave; un-normalised inputs, infinity on either side, P + (-P)."""
    n = 300
    lams, _ = orc.rng_scalars(4321, 2 * n)
    lams[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
    a = np.zeros((n, 12), dtype=np.uint64); b = np.zeros((n, 12), dtype=np.uint64)
    for i in range(n):
        orc.lib().orc_affine_to_jac(orc.ptr(urs4096[i]), orc.ptr(a[i])); orc.lib().orc_affine_to_jac(orc.ptr(urs4096[i + n]), orc.ptr(b[i]))
        if i % 3:
            a[i] = unnormalise(a[i], lams[i]); b[i] = unnormalise(b[i], lams[n + i])
    b[5] = unnormalise(a[5], lams[7])                      # P + P, two representatives
    neg = orc.z(12); orc.lib().orc_point_mul(orc.ptr(a[6]), orc.ptr(orc.fr_to_mont(pm.R_ORDER - 1)), orc.ptr(neg)); b[6] = neg
    inf = np.array(list(a[2][:8]) + [0, 0, 0, 0], dtype=np.uint64)
    a[9] = inf; b[10] = inf; a[11] = inf; b[11] = inf
    for op in (4, 5, 6):
        got = ctx16k.point_op(op, a, b)
        for i in range(n):
            bb = a[i] if (op == 5 or (op == 6 and i % 4 == 3)) else b[i]
            want = orc.z(12); orc.lib().orc_point_add(orc.ptr(a[i]), orc.ptr(bb), orc.ptr(want))
            assert canon(got[i]) == canon(want), (op, i)


# ------------------------------------------------------------------ K1/K2 MSM
EDGE = [0, 1, pm.R_ORDER - 1, 1 << 254, 2, pm.R_ORDER - 2]


@pytest.mark.parametrize("n", [1, 2, 3, 31, 32, 33, 63, 64, 65, 256, 1000, 4096, 16384])
def test_msm_vs_oracle(ctx16k, n):
    gs = ctx16k.read_bases(0, n)
    sc, _ = orc.rng_scalars(0x48414C4F00000002 + n, n)
    for i, e in enumerate(EDGE[: min(n, len(EDGE))]):
        sc[(i * 7) % n] = orc.fr_to_mont(e)
    want = orc.msm_affine(gs, sc)
    got = ctx16k.msm(sc)
    assert canon(got) == canon(want)
    assert got.tolist() == want.tolist()  # both sides normalise: limb-exact
    # canonical (non-Montgomery) scalars give the same answer
    sc_canon = np.array([[(orc.fr_from_mont(s) >> (64 * k)) & (2**64 - 1) for k in range(4)] for s in sc[: min(n, 256)]], dtype=np.uint64)
    if n <= 256:
        assert ctx16k.msm(sc_canon, mont=False).tolist() == want.tolist()


@pytest.mark.parametrize("c", [4, 7, 9, 12, 13, 16])
def test_msm_every_window_size(ctx16k, c):
    n = 2048
    gs = ctx16k.read_bases(100, n)
    sc, _ = orc.rng_scalars(c, n)
    want = orc.msm_affine(gs, sc)
    ctx16k.set_window_bits(c)
    try:
        assert ctx16k.msm(sc, off=100).tolist() == want.tolist()
    finally:
        ctx16k.set_window_bits(0)


@pytest.mark.parametrize("span,task_len", [(1, 8), (2, 16), (16, 32), (64, 64), (0, 8)])
def test_msm_tuning_knobs_do_not_change_results(hal, ctx16k, urs4096, span, task_len):
    """halo_set_reduce_span / halo_set_task_len only move work around (the workspace grows for short tasks)."""
    n = 4096
    sc, _ = orc.rng_scalars(4242 + span, n)
    sc[:1000] = sc[7]  # a fat bucket in every window: many tasks, the big-combine path
    want = orc.msm_affine(urs4096, sc).tolist()
    ctx16k.set_reduce_span(span)
    ctx16k.set_task_len(task_len)
    try:
        for small in (-1, 0):  # the 4-launch pipeline of smsm.hip, then the general one
            ctx16k.set_small_path(small)
            for c in (0, 13):
                ctx16k.set_window_bits(c)
                assert ctx16k.msm(sc).tolist() == want
    finally:
        ctx16k.set_reduce_span(0); ctx16k.set_task_len(0); ctx16k.set_window_bits(0); ctx16k.set_small_path(-1)
    with pytest.raises(hal.HaloError):
        ctx16k.set_task_len(12)
    with pytest.raises(hal.HaloError):
        ctx16k.set_reduce_span(3)


@pytest.mark.parametrize("n", [1, 5, 64, 65, 1000, 4096, 16384])
def test_msm_small_path_equals_general_path(hal, ctx16k, n):
    """smsm.hip (sort per window in LDS, balanced tasks, quad-parallel window sums, last-block combine) against the general
    pipeline and the oracle: random scalars; a fat bucket in every window (the whole-block pre-sum of heavy buckets);
    two buckets holding equal values (P + P inside the quad additions); zeros; every window size the plan may pick."""
    gs = ctx16k.read_bases(0, n)
    sets = []
    sc, s = orc.rng_scalars(31337 + n, n)
    sets.append(sc)
    fat = sc.copy(); fat[: max(1, n * 3 // 4)] = sc[0]; sets.append(fat)
    z = sc.copy(); z[::3] = 0; z[1::3] = orc.fr_to_mont(1); sets.append(z)
    eq = np.ascontiguousarray(np.tile(orc.fr_to_mont(3), (n, 1))); eq[n // 2:] = orc.fr_to_mont(5); sets.append(eq)
    for k, scal in enumerate(sets):
        want = orc.msm_affine(gs, scal).tolist()
        for c in ((0, 4, 6, 8, 10, 13, 14) if n in (1000, 4096) else (0,)):
            ctx16k.set_window_bits(c)
            try:
                ctx16k.set_small_path(-1)
                a = ctx16k.msm(scal).tolist()
                a2 = ctx16k.msm(scal).tolist()  # replayed as a graph on the third call
                a3 = ctx16k.msm(scal).tolist()
                ctx16k.set_small_path(0)
                b = ctx16k.msm(scal).tolist()
            finally:
                ctx16k.set_small_path(-1); ctx16k.set_window_bits(0)
            assert a == want and a2 == want and a3 == want and b == want, (n, k, c)


def test_msm_small_path_same_base(hal, urs4096):
    """every base equal: bucket values are multiples of one point, so equal buckets (P + P) and opposite ones occur in
    the running sums of the quad-parallel reduce"""
    n = 2048
    same = np.ascontiguousarray(np.tile(urs4096[11], (n, 1)))
    c = hal.Context(same)
    try:
        sc, _ = orc.rng_scalars(99, n)
        sc[: n // 2] = orc.fr_to_mont(7)
        neg = np.array([orc.fr_to_mont((pm.R_ORDER - orc.fr_from_mont(x)) % pm.R_ORDER) for x in sc[: n // 4]])
        sc[n // 2: n // 2 + n // 4] = neg
        for cbits in (0, 5, 9):
            c.set_window_bits(cbits)
            assert c.msm(sc).tolist() == orc.msm_affine(same, sc).tolist()
    finally:
        c.close()


@pytest.mark.parametrize("n,c", [(4096, 11), (4096, 13), (16384, 16), (1000, 13), (16384, 12)])
def test_msm_two_level_sort_equals_one_level(hal, ctx16k, urs4096, n, c):
    """halo_set_sort_mode: the coarse-runs + fine-sort path (automatic from n = 2^18) forced at small sizes, against the
    oracle / the one-pass sort; a fat bucket and a zero stretch included (n = 1000 is not a multiple of 8: falls back)."""
    sc, _ = orc.rng_scalars(777 + n + c, n)
    sc[100:600] = sc[3]
    sc[600:700] = 0
    ctx16k.set_window_bits(c)
    ctx16k.set_small_path(0)  # the sort modes belong to the general pipeline
    try:
        ctx16k.set_sort_mode(0)
        one = ctx16k.msm(sc).tolist()
        ctx16k.set_sort_mode(1)
        two = ctx16k.msm(sc).tolist()
        two_again = ctx16k.msm(sc).tolist()
    finally:
        ctx16k.set_sort_mode(-1); ctx16k.set_window_bits(0); ctx16k.set_small_path(-1)
    assert one == two == two_again
    if n <= 4096:
        assert one == orc.msm_affine(urs4096[:n], sc).tolist()
    with pytest.raises(hal.HaloError):
        ctx16k.set_sort_mode(2)


def test_msm_degenerate_inputs(ctx16k):
    n = 4096
    gs = ctx16k.read_bases(0, n)
    G = None
    zero = np.zeros((n, 4), dtype=np.uint64)
    assert canon(ctx16k.msm(zero)) is None
    for val in (1, pm.R_ORDER - 1):
        sc = np.ascontiguousarray(np.tile(orc.fr_to_mont(val), (n, 1)))
        assert ctx16k.msm(sc).tolist() == orc.msm_affine(gs, sc).tolist()
    sc, _ = orc.rng_scalars(5, n)
    sc[::2] = 0  # 50 % zeros
    assert ctx16k.msm(sc).tolist() == orc.msm_affine(gs, sc).tolist()
    # pcdl::commit of a linear polynomial: n - 2 zero scalars (acc.rs:195)
    lin = np.zeros((n, 4), dtype=np.uint64); lin[:2] = sc[1:3]
    assert ctx16k.msm(lin).tolist() == orc.msm_affine(gs, lin).tolist()


def test_msm_all_same_base_and_cancellation(hal, urs4096):
    n = 1024
    same = np.ascontiguousarray(np.tile(urs4096[5], (n, 1)))
    c = hal.Context(same)
    try:
        sc, _ = orc.rng_scalars(9, n)
        assert c.msm(sc).tolist() == orc.msm_affine(same, sc).tolist()   # buckets full of P + P
        sc[1] = orc.fr_to_mont(pm.R_ORDER - orc.fr_from_mont(sc[0]))
        assert canon(c.msm(sc[:2])) is None                              # P + (-P)
        one = np.ascontiguousarray(np.tile(orc.fr_to_mont(1), (n, 1)))
        assert c.msm(one).tolist() == orc.msm_affine(same, one).tolist()
    finally:
        c.close()


def test_msm_points_matches_point_dot(ctx16k, urs4096):
    """group.rs:18-21: arbitrary Jacobian inputs, zip-to-min semantics."""
    m = 300
    pts = np.zeros((m, 12), dtype=np.uint64)
    ks, s = orc.rng_scalars(31, m)
    lams, _ = orc.rng_scalars(s + 7, m)
    lams[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
    for i in range(m):
        j = orc.z(12); orc.lib().orc_affine_to_jac(orc.ptr(urs4096[i]), orc.ptr(j))
        pts[i] = j if i % 5 == 0 else unnormalise(j, lams[i])  # Z != 1 for four in five
    pts[7] = np.array(list(pts[7][:8]) + [0, 0, 0, 0], dtype=np.uint64)  # infinity
    pts[9] = pts[8]  # a repeated (differently scaled below) point
    pts[9] = unnormalise(pts[9], lams[9])
    sc, _ = orc.rng_scalars(s, m + 5)
    want = orc.msm_jac(pts, sc[:m])
    assert ctx16k.msm_points(pts, sc).tolist() == want.tolist()


# ------------------------------------------------------------------ full size (BASELINE config 2)
@pytest.fixture(scope="module")
def ctx1m(hal):
    c = hal.Context(urs_n=1 << 20)
    yield c
    c.close()


def test_urs_extension_prefix_is_consts_table(ctx1m, kat):
    gs = ctx1m.read_bases(0, kat["GS_count"])
    assert hashlib.sha256(gs.tobytes()).hexdigest() == kat["GS_mont_limbs_sha256"]


def test_msm_2_20_vs_oracle(ctx1m):
    n = 1 << 20
    sc, _ = orc.rng_scalars(0x48414C4F00000002, n)
    gs = ctx1m.read_bases()
    got = ctx1m.msm(sc)
    want = orc.msm_affine(gs, sc)  # ~6 s of single-thread CPU
    assert got.tolist() == want.tolist()


def test_host_scalar_msm_in_stretches(hal, ctx1m):
    """halo_msm on scalars in pageable host memory (what integration/ffi.rs point_dot_affine calls) runs a large MSM over the
    c = 20 table as index stretches on several slots, each behind the copy of its own scalars (abi.hip msm_host_pieces).  Against
    the oracle, against halo_msm_dev, with adversarial scalars, on a stretch of the key that does not start at 0, and at sizes
    that do not take the path (not a multiple of 64; below 2^19); other splits in child processes (the split is read once)."""
    import subprocess, sys, torch
    n = 1 << 20
    sc, _ = orc.rng_scalars(0x48414C4F00000002, n)
    gs = ctx1m.read_bases()
    ctx1m.msm(sc)  # (the table exists from here on)
    assert ctx1m.info(0) > 0
    want = orc.msm_affine(gs, sc)
    for _ in range(3):  # plain launches, graph capture, replay
        assert ctx1m.msm(sc).tolist() == want.tolist()
    d = torch.from_numpy(sc.view(np.int64).copy()).cuda()
    rm1 = np.tile(np.array(orc.scalars_to_mont([pm.R_ORDER - 1])[0]), (n, 1))
    half = sc.copy(); half[::2] = 0
    for name, scal in (("all r-1", rm1), ("half zero", half), ("all zero", np.zeros_like(sc))):
        dd = torch.from_numpy(np.ascontiguousarray(scal).view(np.int64).copy()).cuda()
        assert ctx1m.msm(np.ascontiguousarray(scal)).tolist() == ctx1m.msm_dev(dd.data_ptr(), n).tolist(), name
    # stretches of the key: off != 0, lengths that are / are not multiples of 64, below the threshold
    for off, m in ((4096, (1 << 19) + 640), (64, (1 << 20) - 64), (12, (1 << 19) + 100), (0, (1 << 19) - 64)):
        assert ctx1m.msm(np.ascontiguousarray(sc[:m]), off=off).tolist() == ctx1m.msm_dev(d.data_ptr(), m, off=off).tolist(), (off, m)
    assert ctx1m.msm(sc).tolist() == want.tolist()
    # a slot busy with the caller's own asynchronous MSM: the synchronous call takes the single-launch form on slot 0 ... which is
    # busy too -> the library refuses instead of overlapping two launches on one slot
    ctx1m.msm_dev_begin(0, d.data_ptr(), n)
    with pytest.raises(hal.HaloError):
        ctx1m.msm(sc)
    assert ctx1m.msm_dev_end(0).tolist() == want.tolist()
    ctx1m.msm_dev_begin(1, d.data_ptr(), n)   # slot 1 busy: one copy + one launch sequence on slot 0
    assert ctx1m.msm(sc).tolist() == want.tolist()
    assert ctx1m.msm_dev_end(1).tolist() == want.tolist()
    # from 2^21 points on the default is four stretches (2, 4, 4, 6 sixteenths), each of one or more table pieces
    n2 = 1 << 21
    big = hal.Context(urs_n=n2)
    try:
        d2 = torch.empty(n2 * 4, dtype=torch.int64, device="cuda")
        big.rng_scalars_dev(0x48414C4F00000005, n2, d2.data_ptr())
        sc2 = np.ascontiguousarray(d2.cpu().numpy().view(np.uint64).reshape(n2, 4))
        w2 = big.msm_dev(d2.data_ptr(), n2)          # builds the table
        for _ in range(3):
            assert big.msm(sc2).tolist() == w2.tolist()
        assert big.msm(np.ascontiguousarray(sc2[: n2 - 64]), off=64).tolist() == big.msm_dev(d2.data_ptr(), n2 - 64, off=64).tolist()
    finally:
        big.close()
    for split in ("16", "8,8", "2,3,4,7"):
        env = dict(os.environ, HALO_HOST_SPLIT=split)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "host_msm.py"), "20", "3"], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0 and ("HALO_HOST_SPLIT=%s" % split) in out.stdout, out.stderr[-400:]


def test_msm_table_pipeline_equals_general(hal):
    """Fixed-base tables (T[w][i] = 2^(20 w) G_i, one set of 2^19 buckets) against the general pipeline on the same
    context: random scalars, a stretch of the key that does not start at 0, all-equal / all-(r-1) / half-zero scalars, and
    the oracle on the prefix both agree on."""
    import torch
    n = (1 << 20) + (1 << 18)
    c = hal.Context(urs_n=n)
    try:
        d = torch.empty(n * 4, dtype=torch.int64, device="cuda")
        c.rng_scalars_dev(0x48414C4F00000002, n, d.data_ptr())
        cases = [(0, 1 << 20), (12345 * 4, 1 << 20), (0, n)]
        for off, m in cases:
            c.set_table_mode(-1)
            a = c.msm_dev(d.data_ptr() + off * 32, m, off=off)
            a2 = c.msm_dev(d.data_ptr() + off * 32, m, off=off)
            a3 = c.msm_dev(d.data_ptr() + off * 32, m, off=off)   # graph replay
            c.set_table_mode(0)
            b = c.msm_dev(d.data_ptr() + off * 32, m, off=off)
            assert a.tolist() == b.tolist() == a2.tolist() == a3.tolist(), (off, m)
        m = 1 << 20
        for val in (1, pm.R_ORDER - 1, 3):
            sc = torch.from_numpy(np.ascontiguousarray(np.tile(orc.fr_to_mont(val), (m, 1))).view(np.int64)).cuda()
            if val == 3:
                sc.view(m, 4)[::2] = 0
            c.set_table_mode(-1); a = c.msm_dev(sc.data_ptr(), m)
            c.set_table_mode(0); b = c.msm_dev(sc.data_ptr(), m)
            assert a.tolist() == b.tolist(), val
        # below 2^20 points the table is not used: same call, same answer
        c.set_table_mode(-1)
        small = c.msm_dev(d.data_ptr(), 1 << 19)
        c.set_table_mode(0)
        assert small.tolist() == c.msm_dev(d.data_ptr(), 1 << 19).tolist()
    finally:
        c.close()


def test_table_memory_failure_and_release(hal, monkeypatch):
    """(a) a context whose fixed-base table cannot be built (no memory) carries on through the general pipeline --
    same point, no error, no retry per MSM; (b) halo_set_table_mode(ctx, 0) gives a table that was already built back to the
    device (the header says "no table memory")."""
    import torch
    n = 1 << 17
    sc, _ = orc.rng_scalars(0x7AB1E, n)
    good = hal.Context(urs_n=n)
    try:
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        want = good.msm(sc)                      # builds the c = 17 table: 15 x 128 B per point
        assert want.tolist() == orc.msm_affine(good.read_bases(), sc).tolist()
        free1 = torch.cuda.mem_get_info()[0]
        table_bytes = 15 * 128 * n
        assert free0 - free1 >= table_bytes      # (the first MSM may also allocate slot buffers)
        good.set_table_mode(0)
        assert torch.cuda.mem_get_info()[0] - free1 >= table_bytes * 9 // 10, "set_table_mode(0) must release the table"
        assert good.msm(sc).tolist() == want.tolist()
        good.set_table_mode(-1)                  # and it comes back on demand
        assert good.msm(sc).tolist() == want.tolist()
    finally:
        good.close()
    hal.dev_hook("table_fail", 1)   # the allocation of the table fails (development library's hook)
    bad = hal.Context(urs_n=n)
    try:
        for _ in range(3):
            assert bad.msm(sc).tolist() == want.tolist()
        assert bad.info(0) == 0 and bad.info(6) == 4
        # no latch: with the memory back, the table is built at the attempt after the back-off (64 eligible MSMs)
        hal.dev_hook("table_fail", 0)
        built_after = None
        for k in range(70):
            assert bad.msm(sc).tolist() == want.tolist()
            if bad.info(0):
                built_after = k + 1
                break
        assert built_after is not None and 55 <= built_after <= 64, built_after
        assert bad.info(6) == 2 and bad.msm(sc).tolist() == want.tolist()
    finally:
        bad.close()


@pytest.mark.parametrize("n", [1 << 20, 1 << 17, (1 << 18) + 4096, 1 << 19, (1 << 20) + (1 << 18) + 4100])
def test_msm_table_top_window_edges(hal, n):
    """The table pipeline's plans: keys of >= 2^20 points recode s + (i mod 31) r when the scalar's top window (bits 240..)
    is in 1..2^14 -- the same point, since every base has order r; keys of 2^17 .. 2^19 points use 15 windows of 17 bits
    and recode a scalar >= 2^254 as r - s with flipped signs.  Scalars around each edge of both rules, given as plain
    256-bit integers (scalars_are_mont = 0; values in [r, 2^255) included: the window walk takes them as they are), must
    give what the general pipeline gives, whose digits are untouched.  (The last size runs as two pieces of unequal
    length: more than 1,310,720 points.)"""
    import torch
    c = hal.Context(urs_n=n)
    try:
        r = pm.R_ORDER
        edge = [0, 1, (1 << 240) - 1, 1 << 240, (1 << 240) + 1, (16384 << 240) - 1, 16384 << 240, (16384 << 240) + 12345, r - 1, r, r + 1,
                16385 << 240, (1 << 255) - 19, (1 << 20) - 1, 1 << 19, (1 << 19) + 1, (1 << 239) + (1 << 19),
                (1 << 254) - 1, 1 << 254, (1 << 254) + 1, r - (1 << 16), r - (1 << 16) - 1, r + (1 << 100), (1 << 17) - 1, 1 << 16, (1 << 16) + 1,
                (1 << 238) - 1, 1 << 238, ((1 << 16) - 1) << 238, (1 << 254) - (1 << 237)]
        rng = np.random.default_rng(7)
        table = np.array([[(v >> (64 * limb)) & 0xFFFFFFFFFFFFFFFF for limb in range(4)] for v in edge], dtype=np.uint64)
        raw = table[rng.integers(0, len(edge), size=n)]
        raw[::2] = rng.integers(0, 1 << 64, size=(n // 2, 4), dtype=np.uint64)  # every other scalar uniform below 2^254:
        raw[::2, 3] &= np.uint64((1 << 62) - 1)                                    # buckets of every size
        raw = np.ascontiguousarray(raw)
        vals = [sum(int(raw[k, limb]) << (64 * limb) for limb in range(4)) for k in range(1 << 12)]
        d = torch.from_numpy(raw.view(np.int64)).cuda()
        c.set_table_mode(-1); a = c.msm_dev(d.data_ptr(), n, mont=False)
        a2 = c.msm_dev(d.data_ptr(), n, mont=False)                # graph replay
        c.prof_enable(True); c.prof_reset()
        a3 = c.msm_dev(d.data_ptr(), n, mont=False)
        assert "k_tmsm_recode" in c.prof(), "the table pipeline did not run"
        c.prof_enable(False)
        c.set_table_mode(0); b = c.msm_dev(d.data_ptr(), n, mont=False)
        assert a.tolist() == b.tolist() == a2.tolist() == a3.tolist()
        # a stretch that does not start at 0 (at least half the key: still the table)
        off, m = 1028, n - 4096
        c.set_table_mode(-1); a = c.msm_dev(d.data_ptr() + off * 32, m, off=off, mont=False)
        c.set_table_mode(0); b = c.msm_dev(d.data_ptr() + off * 32, m, off=off, mont=False)
        assert a.tolist() == b.tolist()
        # and a slice small enough for the oracle: the same values reduced mod r, in Montgomery form
        m = 1 << 12
        sc = np.ascontiguousarray(np.stack([orc.fr_to_mont(v % r) for v in vals[:m]]))
        gs = c.read_bases()[:m]
        c.set_table_mode(0)
        small = c.msm_dev(d.data_ptr(), m, mont=False)
        assert small.tolist() == orc.msm_affine(gs, sc).tolist()
    finally:
        c.close()


@pytest.mark.parametrize("lg,batch", [(17, 2), (17, 3), (17, 4), (17, 8), (18, 4), (19, 2)])
def test_msm_table_batch_small_keys(hal, lg, batch):
    """A batch of MSMs over a key of 2^17 .. 2^19 points goes through the small-key table plan as ONE launch (the members'
    bucket sets side by side, count * ranges <= 512): every member equals its own single MSM through the general pipeline;
    one member is all zero, one has scalars >= 2^254."""
    import torch
    n = 1 << lg
    c = hal.Context(urs_n=n)
    try:
        sets = []
        for j in range(batch):
            d = torch.empty(n * 4, dtype=torch.int64, device="cuda")
            c.rng_scalars_dev(100 + j, n, d.data_ptr())
            if j == 1: d.zero_()
            sets.append(d)
        if batch > 2:
            rm1 = torch.from_numpy(np.ascontiguousarray(np.tile(orc.fr_to_mont(pm.R_ORDER - 1), (n // 2, 1))).view(np.int64)).cuda()
            sets[2].view(n, 4)[::2] = rm1
        ptrs = [d.data_ptr() for d in sets]
        c.set_table_mode(0)
        want = [c.msm_dev(p, n).tolist() for p in ptrs]
        c.set_table_mode(-1)
        c.msm_dev_batch_begin(0, ptrs, n); got = c.msm_dev_batch_end(0, batch)
        c.msm_dev_batch_begin(0, ptrs, n); got2 = c.msm_dev_batch_end(0, batch)   # graph capture
        c.msm_dev_batch_begin(0, ptrs, n); got3 = c.msm_dev_batch_end(0, batch)   # graph replay
        c.prof_enable(True); c.prof_reset()
        c.msm_dev_batch_begin(1, ptrs, n); got4 = c.msm_dev_batch_end(1, batch)
        assert "k_tmsm_recode" in c.prof(), "the table pipeline did not run"
        c.prof_enable(False)
        for g in (got, got2, got3, got4):
            assert [g[j].tolist() for j in range(batch)] == want
        # a stretch that does not start at 0
        off, m = 2052, n - 8192
        c.set_table_mode(0); w2 = [c.msm_dev(p + off * 32, m, off=off).tolist() for p in ptrs]
        c.set_table_mode(-1)
        c.msm_dev_batch_begin(0, [p + off * 32 for p in ptrs], m, off=off); g5 = c.msm_dev_batch_end(0, batch)
        assert [g5[j].tolist() for j in range(batch)] == w2
    finally:
        c.close()


def test_msm_2_20_linearity(ctx1m):
    """Size-independent property: msm(a) + msm(b) == msm(a + b); msm(k a) == k msm(a)."""
    n = 1 << 20
    a, s = orc.rng_scalars(123, n)
    b, _ = orc.rng_scalars(s, n)
    ai = a.view(np.uint64)
    # a + b in Fr via the GPU's own field op would be circular: add on the host with numpy big-int-free trick:
    # use the oracle for the elementwise sum (fast C loop through ctypes is too slow for 2^20) -> use k = 2: a + a
    two_a = ctx1m.field_op(1, 1, a[: 1 << 14], a[: 1 << 14])  # checked against the model in test_field_ops
    pa = ctx1m.msm(a[: 1 << 14])
    p2a = ctx1m.msm(two_a)
    want = orc.z(12); orc.lib().orc_point_add(orc.ptr(pa), orc.ptr(pa), orc.ptr(want))
    assert p2a.tolist() == want.tolist()
    # split linearity at full size: msm over [0, n) == msm over [0, n/2) + msm over [n/2, n)
    full = ctx1m.msm(a)
    lo = ctx1m.msm(a[: n // 2])
    hi = ctx1m.msm(a[n // 2:], off=n // 2)
    want = orc.z(12); orc.lib().orc_point_add(orc.ptr(lo), orc.ptr(hi), orc.ptr(want))
    assert full.tolist() == want.tolist()


def test_msm_2_20_adversarial_scalars_stay_fast_and_exact(ctx1m):
    """All-equal scalars put every point of a window into ONE bucket (n entries): the task split
    must keep that bounded.  all-one against the oracle; k * all-one by linearity; wall-clock bound."""
    import time
    n = 1 << 20
    gs = ctx1m.read_bases()
    one = np.ascontiguousarray(np.tile(orc.fr_to_mont(1), (n, 1)))
    t = time.time(); got = ctx1m.msm(one); dt_one = time.time() - t
    assert got.tolist() == orc.msm_affine(gs, one).tolist()
    k, _ = orc.rng_scalars(99, 1)
    same = np.ascontiguousarray(np.tile(k[0], (n, 1)))
    t = time.time(); got_k = ctx1m.msm(same); dt_same = time.time() - t
    want = orc.z(12); orc.lib().orc_point_mul(orc.ptr(got), orc.ptr(k[0]), orc.ptr(want))
    assert got_k.tolist() == want.tolist()
    assert dt_one < 0.5 and dt_same < 0.5, (dt_one, dt_same)


def test_msm_host_scalars_async_halves(hal, ctx16k, urs4096):
    """halo_msm_begin / halo_msm_end: host scalars, several slots in flight, results equal halo_msm's and the oracle's;
    a busy slot is refused"""
    sets = [orc.rng_scalars(0xA5 + k, 4096)[0] for k in range(4)]
    for k in range(4):
        ctx16k.msm_begin(k, sets[k], off=k * 8)
    with pytest.raises(hal.HaloError):
        ctx16k.msm_begin(2, sets[0])
    got = [ctx16k.msm_end(k) for k in range(4)]
    gs = ctx16k.read_bases()
    for k in range(4):
        assert got[k].tolist() == ctx16k.msm(sets[k], off=k * 8).tolist() == orc.msm_affine(gs[k * 8:k * 8 + 4096], sets[k]).tolist()


def test_msm_pipelined_slots_agree(ctx1m):
    """halo_msm_dev_begin/_end on all four slots, interleaved, equals the synchronous call."""
    import torch
    n = 1 << 18
    scs = [orc.rng_scalars(1000 + i, n)[0] for i in range(4)]
    want = [ctx1m.msm(s).tolist() for s in scs]
    ds = [torch.from_numpy(s.view(np.int64)).cuda() for s in scs]
    for rep in range(2):
        for slot in range(4):
            ctx1m.msm_dev_begin(slot, ds[slot].data_ptr(), n)
        for slot in (2, 0, 3, 1):
            assert ctx1m.msm_dev_end(slot).tolist() == want[slot]
    with pytest.raises(Exception):
        ctx1m.msm_dev_end(0)  # nothing in flight


@pytest.mark.parametrize("n,batch", [(1, 2), (63, 3), (1000, 8), (4096, 2), (16384, 4), (16384, 8)])
def test_msm_batch_members_equal_single_msms(hal, ctx16k, urs4096, n, batch):
    """halo_msm_dev_batch_*: every member of a batched launch equals the oracle / the single MSM on its scalars."""
    import torch
    scs = [orc.rng_scalars(7000 + 31 * n + i, n)[0] for i in range(batch)]
    if batch > 2:  # adversarial members: all r-1, all zero
        scs[1] = np.tile(orc.fr_to_mont(0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001 - 1), (n, 1))
        scs[2] = np.zeros((n, 4), dtype=np.uint64)
    ds = [torch.from_numpy(np.ascontiguousarray(s).view(np.int64)).cuda() for s in scs]
    if n <= 4096:
        gs = urs4096[:n]
        want = [orc.msm_affine(gs, s).tolist() for s in scs]
    else:
        want = [ctx16k.msm_dev(d.data_ptr(), n).tolist() for d in ds]
    for rep in range(3):  # third repetition replays the captured graph
        ctx16k.msm_dev_batch_begin(2, [d.data_ptr() for d in ds], n)
        got = ctx16k.msm_dev_batch_end(2, batch)
        assert got.tolist() == want
    # the slot goes back to single MSMs afterwards, and a non-zero base offset works
    ctx16k.msm_dev_begin(2, ds[0].data_ptr(), n)
    assert ctx16k.msm_dev_end(2).tolist() == want[0]
    if n == 1000:
        ctx16k.msm_dev_batch_begin(1, [d.data_ptr() for d in ds[:2]], n, off=3000)
        got = ctx16k.msm_dev_batch_end(1, 2)
        gs = urs4096[3000:4000]
        assert got.tolist() == [orc.msm_affine(gs, s).tolist() for s in scs[:2]]


@pytest.mark.parametrize("n,parts", [(1000, 2), (1000, 5), (4096, 8), (16384, 3), (16384, 64)])
def test_msm_window_shards_sum_to_the_msm(hal, ctx16k, urs4096, n, parts):
    """halo_msm_dev_begin_part: the window shards' partial points add up to the full MSM (more shards than
    windows: the surplus shards return infinity)."""
    import torch
    sc = orc.rng_scalars(9100 + n + parts, n)[0]
    sc[0] = orc.fr_to_mont(0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001 - 1)
    d = torch.from_numpy(np.ascontiguousarray(sc).view(np.int64)).cuda()
    want = orc.msm_affine(urs4096[:n], sc).tolist() if n <= 4096 else ctx16k.msm_dev(d.data_ptr(), n).tolist()
    partials = []
    for part in range(parts):
        slot = part % 4
        ctx16k.msm_dev_begin(slot, d.data_ptr(), n, part=part, parts=parts)
        partials.append(ctx16k.msm_dev_end(slot))
    assert hal.point_sum(np.stack(partials)).tolist() == want
    with pytest.raises(hal.HaloError):
        ctx16k.msm_dev_begin(0, d.data_ptr(), n, part=parts, parts=parts)


def test_msm_window_shards_2_20(hal, ctx1m):
    import torch
    n = 1 << 20
    d = torch.empty(n * 4, dtype=torch.int64, device="cuda")
    ctx1m.rng_scalars_dev(0x48414C4F00000002, n, d.data_ptr())
    want = ctx1m.msm_dev(d.data_ptr(), n).tolist()
    for parts in (2, 8):
        partials = []
        for part in range(parts):
            ctx1m.msm_dev_begin(part % 4, d.data_ptr(), n, part=part, parts=parts)
            if part % 4 == 3:
                partials += [ctx1m.msm_dev_end(s) for s in range(4)]
        partials += [ctx1m.msm_dev_end(s) for s in range(parts % 4)]
        assert hal.point_sum(np.stack(partials)).tolist() == want


def test_msm_randomised_configurations(hal, ctx16k, urs4096):
    """Seeded sweep over sizes (incl. non-multiples of 8 and chunk-boundary sizes), window bits, sort modes, batch sizes,
    window shards and scalar shapes (zeros, repeats, tiny values): every combination equals the oracle."""
    import torch
    rnd = np.random.RandomState(20260101)
    sizes = [1, 7, 8, 9, 255, 256, 1023, 1024, 1025, 2040, 2048, 3333, 4088, 4096]
    try:
        for trial in range(36):
            n = int(sizes[rnd.randint(len(sizes))])
            c = int(rnd.choice([0, 8, 10, 11, 13, 14, 16]))
            mode = int(rnd.randint(0, 2))
            batch = int(rnd.randint(1, 5))
            parts = int(rnd.choice([1, 1, 2, 3, 5]))
            scs = []
            for b in range(batch):
                sc = orc.rng_scalars(31337 + 97 * trial + b, n)[0]
                shape = rnd.randint(4)
                if shape == 1:
                    sc[rnd.rand(n) < 0.5] = 0
                elif shape == 2:
                    sc[:] = sc[rnd.randint(0, n, size=n) % max(1, n // 16)]  # few distinct values
                elif shape == 3:
                    sc[: n // 2] = orc.scalars_to_mont([int(v) for v in rnd.randint(0, 70000, size=3)] * (n // 6 + 1))[: n // 2]
                scs.append(np.ascontiguousarray(sc))
            want = [orc.msm_affine(urs4096[:n], sc).tolist() for sc in scs]
            ds = [torch.from_numpy(sc.view(np.int64)).cuda() for sc in scs]
            ctx16k.set_window_bits(c); ctx16k.set_sort_mode(mode)
            got = []
            for part in range(parts):
                ctx16k.msm_dev_batch_begin(trial % 4, [d.data_ptr() for d in ds], n, part=part, parts=parts)
                got.append(ctx16k.msm_dev_batch_end(trial % 4, batch))
            for b in range(batch):
                total = hal.point_sum(np.stack([g[b] for g in got]))
                assert total.tolist() == want[b], (trial, n, c, mode, batch, parts, b)
    finally:
        ctx16k.set_window_bits(0); ctx16k.set_sort_mode(-1)


def test_msm_batched_window_shards_2_20(hal, ctx1m):
    """What one rank of an 8-rank window-sharded run launches (4 MSMs x 2 of the 16 windows), for every rank: the
    partials of each MSM add up to the unsharded result."""
    import torch
    n = 1 << 20
    ds = []
    for i in range(4):
        d = torch.empty(n * 4, dtype=torch.int64, device="cuda")
        ctx1m.rng_scalars_dev(0x48414C4F00000002 + 4 * i * n * 0x9E3779B97F4A7C15 & (2**64 - 1), n, d.data_ptr())
        ds.append(d)
    want = [ctx1m.msm_dev(d.data_ptr(), n).tolist() for d in ds]
    ptrs = [d.data_ptr() for d in ds]
    partials = []
    for rank in range(8):
        ctx1m.msm_dev_batch_begin(rank % 4, ptrs, n, part=rank, parts=8)
        partials.append(ctx1m.msm_dev_batch_end(rank % 4, 4))
    for b in range(4):
        assert hal.point_sum(np.stack([p[b] for p in partials])).tolist() == want[b]


def test_msm_2_22_split_linearity():
    """n = 2^22 > 2^21: the two-level sort can no longer pack index, sign and bucket bits into one word and looks the
    digits up instead; checked by the size-independent split property (BASELINE config 5 scale on one rank: 2^21)."""
    import torch
    import halo_accumulation_amd as h
    n = 1 << 22
    c = h._lib.Context(urs_n=n)
    try:
        d = torch.empty(n * 4, dtype=torch.int64, device="cuda")
        c.rng_scalars_dev(0x48414C4F00000005, n, d.data_ptr())
        full = c.msm_dev(d.data_ptr(), n)
        lo = c.msm_dev(d.data_ptr(), n // 2)
        hi = c.msm_dev(d.data_ptr() + (n // 2) * 32, n // 2, off=n // 2)
        assert h._lib.point_sum(np.stack([lo, hi])).tolist() == full.tolist()
        q1 = c.msm_dev(d.data_ptr(), n // 4)   # 2^20: the packed form on the same context
        assert h._lib.point_sum(np.stack([q1, c.msm_dev(d.data_ptr() + (n // 4) * 32, n // 4, off=n // 4)])).tolist() == lo.tolist()
    finally:
        c.close()


def test_msm_2_24_shards_and_oracle_slice(hal):
    """BASELINE config 5 (n = 2^24), everything one GPU can say about it: the key comes from the main.rs:18-45
    derivation on the device, the scalars from seed ...05; split-linearity; 8 window shards and 8 index shards
    (the per-rank shares of an 8-GPU run) each sum to the unsharded point; an index shard computed by a context
    of its own (as a rank would hold it) equals the same block of the big context; and a 2^18-point slice is
    compared with the CPU oracle."""
    import torch
    n = 1 << 24
    c = hal.Context(urs_n=n)
    try:
        d = torch.empty(n * 4, dtype=torch.int64, device="cuda")
        c.rng_scalars_dev(0x48414C4F00000005, n, d.data_ptr())
        base = d.data_ptr()
        full = c.msm_dev(base, n)
        assert canon(full) is not None
        lo = c.msm_dev(base, n // 2)
        hi = c.msm_dev(base + (n // 2) * 32, n // 2, off=n // 2)
        assert hal.point_sum(np.stack([lo, hi])).tolist() == full.tolist()
        # 8 window shards (halo_msm_dev_begin_part): what bench.py --gpus 8 --shard window gives each rank
        parts = []
        for r in range(8):
            c.msm_dev_begin(r % 4, base, n, part=r, parts=8)
            parts.append(c.msm_dev_end(r % 4))
        assert hal.point_sum(np.stack(parts)).tolist() == full.tolist()
        # 8 index shards: blocks of 2^21 bases and scalars
        from halo_accumulation_amd.sharded import shard_range
        parts = []
        for r in range(8):
            a, b = shard_range(n, r, 8)
            parts.append(c.msm_dev(base + a * 32, b - a, off=a))
        assert hal.point_sum(np.stack(parts)).tolist() == full.tolist()
        # rank 5's shard from a context of its own: G_i = hash(i + 2) starts at its block (main.rs:35-45)
        a, b = shard_range(n, 5, 8)
        own = hal.Context(urs_n=b - a, first_index=2 + a)
        try:
            assert own.msm_dev(base + a * 32, b - a).tolist() == parts[5].tolist()
        finally:
            own.close()
        # oracle on a slice that crosses nothing special: 2^18 points from offset 5 * 2^21 + 12345
        off, m = a + 12345, 1 << 18
        gs = c.read_bases(off, m)
        sc = np.ascontiguousarray(d[off * 4:(off + m) * 4].cpu().numpy().view(np.uint64).reshape(m, 4))
        assert gs[:2].tolist() == orc.urs_affine(2 + off, 2).tolist()
        assert c.msm_dev(base + off * 32, m, off=off).tolist() == orc.msm_affine(gs, sc).tolist()
    finally:
        c.close()


def test_msm_batch_2_18_and_misuse(hal, ctx1m):
    import torch
    n = 1 << 18
    ds = []
    for i in range(4):
        d = torch.empty(n * 4, dtype=torch.int64, device="cuda")
        ctx1m.rng_scalars_dev(555 + i, n, d.data_ptr())
        ds.append(d)
    want = [ctx1m.msm_dev(d.data_ptr(), n).tolist() for d in ds]
    ctx1m.msm_dev_batch_begin(0, [d.data_ptr() for d in ds], n)
    ctx1m.msm_dev_batch_begin(3, [d.data_ptr() for d in ds[::-1]], n)
    assert ctx1m.msm_dev_batch_end(3, 4).tolist() == want[::-1]
    with pytest.raises(hal.HaloError):
        ctx1m.msm_dev_batch_end(0, 2)      # wrong batch size: reported, the batch stays in flight
    assert ctx1m.msm_dev_batch_end(0, 4).tolist() == want
    with pytest.raises(hal.HaloError):
        ctx1m.msm_dev_batch_begin(0, [d.data_ptr() for d in ds] * 3, n)   # 12 members
    with pytest.raises(hal.HaloError):
        ctx1m.msm_dev_batch_begin(0, [ds[0].data_ptr(), 0], n)            # null member


def test_device_rng_matches_stream(ctx16k):
    """halo_rng_scalars_dev == the sequential SplitMix64 stream the oracle and the tests use."""
    import torch
    n = 1000
    d = torch.empty(n * 4, dtype=torch.int64, device="cuda")
    st = ctx16k.rng_scalars_dev(0x48414C4F00000002, n, d.data_ptr())
    want, st_ref = orc.rng_scalars(0x48414C4F00000002, n)
    assert d.cpu().numpy().view(np.uint64).reshape(n, 4).tolist() == want.tolist() and st == st_ref


def test_api_misuse_is_reported_not_crashed(hal, ctx16k):
    """Bad sizes / slots / states come back as HaloError with a message; nothing is dereferenced."""
    import torch
    n = 16384
    sc = np.zeros((n + 1, 4), dtype=np.uint64)
    with pytest.raises(hal.HaloError):
        ctx16k.msm(sc)                                  # more scalars than bases
    with pytest.raises(hal.HaloError):
        ctx16k.msm(sc[:8], off=n - 4)                   # range past the key
    d = torch.zeros(64 * 4, dtype=torch.int64, device="cuda")
    with pytest.raises(hal.HaloError):
        ctx16k.msm_dev_begin(7, d.data_ptr(), 64)       # slot out of range
    ctx16k.msm_dev_begin(1, d.data_ptr(), 64)
    with pytest.raises(hal.HaloError):
        ctx16k.msm_dev_begin(1, d.data_ptr(), 64)       # slot busy
    assert canon(ctx16k.msm_dev_end(1)) is None         # all-zero scalars
    with pytest.raises(hal.HaloError):
        ctx16k.set_window_bits(3)
    ipa = hal.Ipa(ctx16k, 4, sc[:4], sc[0])
    with pytest.raises(hal.HaloError):
        ipa.finish()                                    # rounds remaining
    with pytest.raises(AssertionError):
        hal.Ipa(ctx16k, 6, sc[:4], sc[0])               # not a power of two (pcdl.rs:130)
    with pytest.raises(AssertionError):
        hal.Ipa(ctx16k, 1 << 15, sc[:4], sc[0])         # larger than the key (pcdl.rs:132)
