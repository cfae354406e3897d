"""Wire format: ark-serialize-compatible compressed encoding of EvalProof / Instance / Accumulator.
Host-only code of libhalo_hip.so: runs without a GPU.  Known answers are computed here with Python integers."""
import numpy as np
import pytest

import orc
import pallas_model as pm


@pytest.fixture(scope="module")
def hal():
    import halo_accumulation_amd as h
    return h._lib


@pytest.fixture(scope="module")
def pp(urs4096):
    return orc.make_pp(urs4096)


def fq_mont(v):
    v = v * pm.MONT_R % pm.P
    return [(v >> (64 * k)) & (2**64 - 1) for k in range(4)]


def jac(x, y):
    return np.array(fq_mont(x) + fq_mont(y) + fq_mont(1), dtype=np.uint64)


INF = np.array(fq_mont(1) + fq_mont(1) + [0, 0, 0, 0], dtype=np.uint64)


def point_bytes(x, y):
    """33 bytes: x LE, bit 7 of byte 32 set iff y > -y (ark-ec to_flags), bit 6 = infinity"""
    b = bytearray(x.to_bytes(32, "little") + b"\x00")
    if y > (pm.P - y) % pm.P:
        b[32] |= 0x80
    return bytes(b)


def make_proof(lg, hiding, pts, c, wprime):
    pf = np.zeros(2 + 24 * lg + 32, dtype=np.uint64)
    pf[0], pf[1] = int(hiding), lg
    for i in range(2 * lg + 1):
        pf[2 + 12 * i: 14 + 12 * i] = pts[i]
    o = 2 + 24 * lg
    pf[o + 12: o + 16] = orc.fr_to_mont(c)
    pf[o + 16: o + 28] = pts[2 * lg + 1] if hiding else INF
    if hiding:
        pf[o + 28: o + 32] = orc.fr_to_mont(wprime)
    return pf


def test_known_answer_bytes(hal):
    """generator (-1, 2), its negative, infinity, edge scalars: the exact bytes"""
    g, ng = jac(pm.P - 1, 2), jac(pm.P - 1, pm.P - 2)
    lg = 1
    pf = make_proof(lg, True, [g, ng, INF, g], pm.R_ORDER - 1, 1)
    data = hal.proof_encode(pf)
    want = (lg.to_bytes(8, "little") + point_bytes(pm.P - 1, 2) + lg.to_bytes(8, "little") + point_bytes(pm.P - 1, pm.P - 2)
            + bytes(32) + b"\x40" + (pm.R_ORDER - 1).to_bytes(32, "little") + b"\x01" + point_bytes(pm.P - 1, 2)
            + b"\x01" + (1).to_bytes(32, "little"))
    assert data == want
    assert point_bytes(pm.P - 1, 2)[32] == 0 and point_bytes(pm.P - 1, pm.P - 2)[32] == 0x80
    assert len(data) == hal.load().halo_proof_encoded_size(lg, 1)
    assert hal.proof_decode(data).tolist() == pf.tolist()
    # non-hiding: the two Option tags are 0 and nothing follows them
    pf0 = make_proof(lg, False, [g, ng, g], 7, 0)
    d0 = hal.proof_encode(pf0)
    assert d0[-2:] == b"\x00\x00" and len(d0) == hal.load().halo_proof_encoded_size(lg, 0)
    assert hal.proof_decode(d0).tolist() == pf0.tolist()


def test_roundtrip_real_proofs_instances_accumulators(hal, pp, urs4096):
    for n, hiding in ((8, True), (8, False), (64, True)):
        d = n - 1
        coeffs, s = orc.rng_scalars(1000 + n, n)
        zw, _ = orc.rng_scalars(s, 2)
        w = zw[1] if hiding else None
        Cm = orc.pcdl_commit(pp, coeffs, d, w)
        pf, _ = orc.pcdl_open(pp, 5, coeffs, Cm, d, zw[0], w)
        data = hal.proof_encode(pf)
        assert len(data) == hal.load().halo_proof_encoded_size(n.bit_length() - 1, int(hiding))
        back = hal.proof_decode(data)
        assert back.tolist() == pf.tolist()
        orc.pcdl_check(pp, Cm, d, zw[0], orc.poly_eval(coeffs, zw[0]), back)
    seed = 77
    q, seed = orc.random_instance(pp, seed, 15)
    assert hal.instance_decode(hal.instance_encode(q)).tolist() == q.tolist()
    acc, seed = orc.acc_prover(pp, seed, 15, [q])
    data = hal.accumulator_encode(acc)
    back = hal.accumulator_decode(data)
    assert back.tolist() == acc.tolist()
    orc.acc_verifier(pp, 15, [q], back)
    orc.acc_decider(pp, back)
    # an accumulator is an instance followed by pi_V: the instance bytes are a prefix
    assert data.startswith(hal.instance_encode(acc[: orc.instance_words(4)]))


def test_decode_rejects_malformed_input(hal, pp):
    coeffs, s = orc.rng_scalars(3, 8)
    zw, _ = orc.rng_scalars(s, 2)
    Cm = orc.pcdl_commit(pp, coeffs, 7, zw[1])
    pf, _ = orc.pcdl_open(pp, 5, coeffs, Cm, 7, zw[0], zw[1])
    good = hal.proof_encode(pf)
    bad = []
    bad.append(good[:-1])                                     # truncated
    bad.append(good + b"\x00")                                # trailing byte
    bad.append(b"\xff" * 8 + good[8:])                        # absurd vector length
    b = bytearray(good); b[8 + 32] |= 0xC0; bad.append(bytes(b))          # both flag bits
    b = bytearray(good); b[8 + 32] |= 0x40; bad.append(bytes(b))          # infinity flag with x != 0
    b = bytearray(good); b[8:8 + 32] = (pm.P).to_bytes(32, "little"); bad.append(bytes(b))  # x = p: not canonical
    # an x with no point on the curve: x^3 + 5 a non-residue
    x = 1
    while pow((x**3 + 5) % pm.P, (pm.P - 1) // 2, pm.P) == 1:
        x += 1
    b = bytearray(good); b[8:8 + 33] = x.to_bytes(32, "little") + b"\x00"; bad.append(bytes(b))
    o = 8 + 33 * 3 + 8 + 33 * 3 + 33
    b = bytearray(good); b[o:o + 32] = pm.R_ORDER.to_bytes(32, "little"); bad.append(bytes(b))  # c = r: not canonical
    b = bytearray(good); b[o + 32] = 2; bad.append(bytes(b))              # Option tag 2
    b = bytearray(good); b[8 + 33 * 3] = 2; bad.append(bytes(b))          # Rs shorter/longer than Ls
    for k, data in enumerate(bad):
        with pytest.raises(ValueError):
            hal.proof_decode(data)
    # every x that IS on the curve decodes to the root the flag names
    for flag in (0, 0x80):
        y = pm.sqrt_mod(((pm.P - 1) ** 3 + 5) % pm.P, pm.P) if hasattr(pm, "sqrt_mod") else 2
        b = bytearray(good); b[8:8 + 33] = (pm.P - 1).to_bytes(32, "little") + bytes([flag])
        got = hal.proof_decode(bytes(b))
        xy = orc.point_canonical(got[2:14])
        assert xy[0] == pm.P - 1 and (xy[1] > pm.P - xy[1]) == (flag == 0x80) and xy[1] in (2, pm.P - 2)


def test_encode_rejects_blobs_with_a_corrupt_header(hal, pp):
    """The encoders take lg and the hiding flag from the blob itself: a corrupted or foreign blob must be refused, not
    read out of bounds (the decoders cap lg at 40 the same way)."""
    coeffs, s = orc.rng_scalars(3, 8)
    zw, _ = orc.rng_scalars(s, 2)
    Cm = orc.pcdl_commit(pp, coeffs, 7, zw[1])
    pf, _ = orc.pcdl_open(pp, 5, coeffs, Cm, 7, zw[0], zw[1])
    lib = hal.load()
    buf = hal.C.create_string_buffer(1 << 16)
    n = hal.C.c_size_t()
    for word, value in ((1, 41), (1, 2**40), (0, 2)):
        bad = pf.copy()
        bad[word] = value
        assert lib.halo_proof_encode(hal.ptr(bad), buf, len(buf), hal.C.byref(n)) == hal.HALO_E_ARG
        inst = np.concatenate([np.zeros(21, dtype=np.uint64), bad])
        assert lib.halo_instance_encode(hal.ptr(inst), buf, len(buf), hal.C.byref(n)) == hal.HALO_E_ARG
        acc = np.concatenate([inst, np.zeros(24, dtype=np.uint64)])
        assert lib.halo_accumulator_encode(hal.ptr(acc), buf, len(buf), hal.C.byref(n)) == hal.HALO_E_ARG
