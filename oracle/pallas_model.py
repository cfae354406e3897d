"""Python big-integer model of the halo-accumulation MSM / IPA hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker.

This is the slow, independent model (Python ``int`` arithmetic, affine group
law with modular inverses).  It restates, for small sizes:

* the Pallas curve and its two fields as used by the reference
  (``code/src/group.rs:7-10``: ark_pallas Projective / Affine / Fr);
* the public-parameter derivation of ``code/src/main.rs:18-45``;
* ``group.rs:13-37`` (scalar_dot, point_dot, construct_powers),
  ``group.rs:41-89`` (rho_0!/rho_1! Fiat-Shamir hashes),
  ``pedersen.rs:6-20``, ``pcdl.rs:49-92,99-110,120-242,252-342`` and
  ``acc.rs:61-107,135-255``.

The field/curve arithmetic itself lives in un-vendored crates (ark-ec / ark-ff /
ark-pallas / ark-poly / ark-serialize 0.5.0, sha3 0.10.8 per ``code/Cargo.lock``);
their published semantics are restated here.

Pinning: ``tests/test_oracle_golden.py`` checks this model against the
reference's only hard-coded numeric truth, the 16,386-point table in
``code/src/consts.rs`` (S, H, GS[0..16384)), via ``tests/golden/urs_kat.json``.
The Fiat-Shamir transcript *bytes* (ark-serialize compressed encoding) are
"parity unpinned": the reference holds no known-answer vector for them.
"""
from __future__ import annotations

import hashlib
from typing import List, Optional, Sequence, Tuple

# --- fields ----------------------------------------------------------------
# ark_pallas::Fq (base field, coordinates) and ark_pallas::Fr (scalars).
P = 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001
R_ORDER = 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001
CURVE_B = 5
MONT_R = 1 << 256  # arkworks in-memory Montgomery radix for 4x64-bit limbs

GENESIS = b"To understand recursion, one must first understand recursion"  # main.rs:19


def fq(x: int) -> int:
    return x % P


def fr(x: int) -> int:
    return x % R_ORDER


def inv_mod(x: int, m: int) -> int:
    if x % m == 0:
        raise ZeroDivisionError("inverse of zero")
    return pow(x, -1, m)


# --- Montgomery limb encoding (consts.rs:4-21 / main.rs:47-53) ---------------
def to_mont_limbs(x: int, modulus: int) -> List[int]:
    v = (x * MONT_R) % modulus
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]


def from_mont_limbs(limbs: Sequence[int], modulus: int) -> int:
    v = sum(int(l) << (64 * i) for i, l in enumerate(limbs))
    return (v * inv_mod(MONT_R, modulus)) % modulus


# --- group: affine points, None = point at infinity -------------------------
Point = Optional[Tuple[int, int]]
GENERATOR: Point = (P - 1, 2)  # ark-pallas generator (-1, 2)


def is_on_curve(pt: Point) -> bool:
    if pt is None:
        return True
    x, y = pt
    return (y * y - x * x * x - CURVE_B) % P == 0


def neg(pt: Point) -> Point:
    if pt is None:
        return None
    return (pt[0], (-pt[1]) % P)


def add(a: Point, b: Point) -> Point:
    if a is None:
        return b
    if b is None:
        return a
    x1, y1 = a
    x2, y2 = b
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return None
        lam = (3 * x1 * x1) * inv_mod(2 * y1, P) % P
    else:
        lam = (y2 - y1) * inv_mod(x2 - x1, P) % P
    x3 = (lam * lam - x1 - x2) % P
    y3 = (lam * (x1 - x3) - y1) % P
    return (x3, y3)


def mul(pt: Point, k: int) -> Point:
    k %= R_ORDER
    acc: Point = None
    addend = pt
    while k:
        if k & 1:
            acc = add(acc, addend)
        addend = add(addend, addend)
        k >>= 1
    return acc


def jacobian_to_affine(X: int, Y: int, Z: int) -> Point:
    if Z % P == 0:
        return None
    zi = inv_mod(Z, P)
    zi2 = zi * zi % P
    return (X * zi2 % P, Y * zi2 * zi % P)


# --- main.rs:18-45 : public parameters ---------------------------------------
def from_le_bytes_mod_order(b: bytes, modulus: int = R_ORDER) -> int:
    return int.from_bytes(b, "little") % modulus


def generator_hash_scalar(i: int) -> int:
    """Scalar of main.rs:18-32 (genesis string first, then LE64 index)."""
    h = hashlib.sha3_256()
    h.update(GENESIS)
    h.update(int(i).to_bytes(8, "little"))
    return from_le_bytes_mod_order(h.digest())


def get_generator_hash(i: int) -> Point:
    return mul(GENERATOR, generator_hash_scalar(i))


def get_pp(n: int) -> Tuple[Point, Point, List[Point]]:
    """main.rs:35-45: S = hash(0), H = hash(1), G_i = hash(i + 2)."""
    return get_generator_hash(0), get_generator_hash(1), [get_generator_hash(i) for i in range(2, n + 2)]


# --- group.rs:13-37 ----------------------------------------------------------
def scalar_dot(xs: Sequence[int], ys: Sequence[int]) -> int:
    return sum(x * y for x, y in zip(xs, ys)) % R_ORDER


def point_dot(xs: Sequence[int], gs: Sequence[Point]) -> Point:
    """Naive MSM, zip-to-min like ark-ec's msm_unchecked (group.rs:18-26)."""
    acc: Point = None
    for x, g in zip(xs, gs):
        acc = add(acc, mul(g, x))
    return acc


def construct_powers(z: int, n: int) -> List[int]:
    out, cur = [], 1
    for _ in range(n):
        out.append(cur)
        cur = cur * z % R_ORDER
    return out


# --- ark-serialize 0.5 compressed encodings (UNPINNED, see module docstring) --
def ser_scalar(x: int) -> bytes:
    return int(x % R_ORDER).to_bytes(32, "little")


def ser_point(pt: Point) -> bytes:
    """33 bytes: x LE in 0..31, flags in byte 32 (bit7: y > -y, bit6: infinity)."""
    if pt is None:
        return bytes(32) + bytes([0x40])
    x, y = pt
    flag = 0x80 if y > (P - y) % P else 0
    return int(x).to_bytes(32, "little") + bytes([flag])


def _rho(tag: int, data: bytes) -> int:
    h = hashlib.sha3_256()
    h.update(data)
    h.update(int(tag).to_bytes(4, "little"))
    return from_le_bytes_mod_order(h.digest())


def rho_0(*items) -> int:
    """group.rs:41-64.  items: ('s', scalar) or ('p', point) tuples."""
    return _rho(0, b"".join(_ser_item(it) for it in items))


def rho_1(*items) -> int:
    """group.rs:66-89."""
    return _rho(1, b"".join(_ser_item(it) for it in items))


def _ser_item(it) -> bytes:
    kind, val = it
    if kind == "s":
        return ser_scalar(val)
    if kind == "p":
        return ser_point(val)
    if kind == "raw":
        return val
    raise ValueError(kind)


# --- pedersen.rs:6-20 --------------------------------------------------------
def pedersen_commit(w: Optional[int], gs: Sequence[Point], ms: Sequence[int], S: Point) -> Point:
    if len(gs) != len(ms):
        raise AssertionError("Length did not match for pedersen commitment: %d, %d" % (len(gs), len(ms)))
    acc = point_dot(ms, gs)
    if w is not None:
        acc = add(mul(S, w), acc)
    return acc


# --- pcdl.rs:49-92 : h(X) ----------------------------------------------------
def h_coeffs(xis: Sequence[int]) -> List[int]:
    """Dense coefficients of h (pcdl.rs:56-77): coeff[k] = prod_{bit i of k} xi_{lg n - i}."""
    lg_n = len(xis) - 1
    coeffs = [1]
    for i in range(lg_n):
        x = xis[lg_n - i]
        coeffs = coeffs + [c * x % R_ORDER for c in coeffs]
    return coeffs


def h_eval(xis: Sequence[int], z: int) -> int:
    """pcdl.rs:79-91."""
    lg_n = len(xis) - 1
    v = (1 + xis[lg_n] * z) % R_ORDER
    z_i = z % R_ORDER
    for i in range(1, lg_n):
        z_i = z_i * z_i % R_ORDER
        v = v * (1 + xis[lg_n - i] * z_i) % R_ORDER
    return v


def poly_eval(coeffs: Sequence[int], z: int) -> int:
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * z + c) % R_ORDER
    return acc


def poly_degree(coeffs: Sequence[int]) -> int:
    d = 0
    for i, c in enumerate(coeffs):
        if c % R_ORDER:
            d = i
    return d


# --- pcdl.rs:99-342 ----------------------------------------------------------
class PublicParams:
    def __init__(self, S: Point, H: Point, GS: Sequence[Point]):
        self.S, self.H, self.GS = S, H, list(GS)
        self.N = len(self.GS)
        self.D = self.N - 1


def pcdl_commit(pp: PublicParams, coeffs: Sequence[int], d: int, w: Optional[int]) -> Point:
    n = d + 1
    assert n & (n - 1) == 0 and n > 0
    assert poly_degree(coeffs) <= d
    assert d <= pp.D
    cs = [c % R_ORDER for c in coeffs][:n] + [0] * max(0, n - len(coeffs))
    return pedersen_commit(w, pp.GS[:n], cs, pp.S)


def pcdl_open(pp: PublicParams, coeffs: Sequence[int], C: Point, d: int, z: int,
              w: Optional[int], q_coeffs: Optional[Sequence[int]] = None, w_bar: Optional[int] = None):
    """pcdl.rs:120-242.  Hiding randomness (q, w_bar) is passed explicitly.

    Returns dict(Ls, Rs, U, c, C_bar, w_prime)."""
    n = d + 1
    lg_n = n.bit_length() - 1
    assert n & (n - 1) == 0
    deg = poly_degree(coeffs)
    assert deg <= d <= pp.D
    p = [c % R_ORDER for c in coeffs]
    v = poly_eval(p, z)
    if w is not None:
        assert q_coeffs is not None and w_bar is not None
        q = list(q_coeffs)  # degree deg-1
        # p_bar = q * (X - z)
        p_bar = [0] * (len(q) + 1)
        for i, qc in enumerate(q):
            p_bar[i] = (p_bar[i] - z * qc) % R_ORDER
            p_bar[i + 1] = (p_bar[i + 1] + qc) % R_ORDER
        assert poly_eval(p_bar, z) == 0
        C_bar = pcdl_commit(pp, p_bar, d, w_bar)
        a = rho_0(("p", C), ("s", z), ("s", v), ("p", C_bar))
        ln = max(len(p), len(p_bar))
        pp_ = [((p[i] if i < len(p) else 0) + a * (p_bar[i] if i < len(p_bar) else 0)) % R_ORDER for i in range(ln)]
        w_prime = (w_bar * a + w) % R_ORDER
        C_prime = add(add(C, mul(C_bar, a)), neg(mul(pp.S, w_prime)))
        p_prime = pp_
    else:
        p_prime, C_prime, w_prime, C_bar = p, C, None, None
    xi = rho_0(("p", C_prime), ("s", z), ("s", v))
    H_prime = mul(pp.H, xi)
    cs = list(p_prime)[:n] + [0] * max(0, n - len(p_prime))
    gs = list(pp.GS[:n])
    zs = construct_powers(z, n)
    Ls, Rs = [], []
    m = n // 2
    for _ in range(lg_n):
        g_l, g_r = gs[:m], gs[m:2 * m]
        c_l, c_r = cs[:m], cs[m:2 * m]
        z_l, z_r = zs[:m], zs[m:2 * m]
        L = add(point_dot(c_r, g_l), mul(H_prime, scalar_dot(c_r, z_l)))
        R = add(point_dot(c_l, g_r), mul(H_prime, scalar_dot(c_l, z_r)))
        Ls.append(L)
        Rs.append(R)
        xi = rho_0(("s", xi), ("p", L), ("p", R))
        xi_inv = inv_mod(xi, R_ORDER)
        gs = [add(g_l[j], mul(g_r[j], xi)) for j in range(m)]
        cs = [(c_l[j] + c_r[j] * xi_inv) % R_ORDER for j in range(m)]
        zs = [(z_l[j] + z_r[j] * xi) % R_ORDER for j in range(m)]
        m //= 2
    return dict(Ls=Ls, Rs=Rs, U=gs[0], c=cs[0], C_bar=C_bar, w_prime=w_prime)


def pcdl_succinct_check(pp: PublicParams, C: Point, d: int, z: int, v: int, pi: dict):
    """pcdl.rs:252-314.  Returns (xis, U) or raises ValueError."""
    n = d + 1
    lg_n = n.bit_length() - 1
    if n & (n - 1):
        raise ValueError("d+1 is not a power of 2!")
    if d > pp.D:
        raise ValueError("d was larger than D!")
    if pi["C_bar"] is not None:
        a = rho_0(("p", C), ("s", z), ("s", v), ("p", pi["C_bar"]))
        C_prime = add(add(C, mul(pi["C_bar"], a)), neg(mul(pp.S, pi["w_prime"])))
    else:
        C_prime = C
    xi0 = rho_0(("p", C_prime), ("s", z), ("s", v))
    xis = [xi0]
    H_prime = mul(pp.H, xi0)
    C_i = add(C_prime, mul(H_prime, v))
    for i in range(lg_n):
        xi_next = rho_0(("s", xis[i]), ("p", pi["Ls"][i]), ("p", pi["Rs"][i]))
        xis.append(xi_next)
        C_i = add(C_i, add(mul(pi["Ls"][i], inv_mod(xi_next, R_ORDER)), mul(pi["Rs"][i], xi_next)))
    v_prime = pi["c"] * h_eval(xis, z) % R_ORDER
    if C_i != add(mul(pi["U"], pi["c"]), mul(H_prime, v_prime)):
        raise ValueError("C_(log_n) != CM.Commit_Sigma(c || v')")
    return xis, pi["U"]


def pcdl_check(pp: PublicParams, C: Point, d: int, z: int, v: int, pi: dict) -> None:
    """pcdl.rs:323-342."""
    xis, U = pcdl_succinct_check(pp, C, d, z, v, pi)
    comm = pedersen_commit(None, pp.GS[: d + 1], h_coeffs(xis), pp.S)
    if U != comm:
        raise ValueError("U != CM.Commit(ck, h_vec)")


# --- deterministic input generator shared by oracle, product and tests -------
class SplitMix64:
    """Counter-based generator used for all synthetic inputs."""

    GAMMA = 0x9E3779B97F4A7C15
    MASK = (1 << 64) - 1

    def __init__(self, seed: int):
        self.state = seed & self.MASK

    def next_u64(self) -> int:
        self.state = (self.state + self.GAMMA) & self.MASK
        z = self.state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & self.MASK
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & self.MASK
        return z ^ (z >> 31)

    def next_scalar(self) -> int:
        """4 x u64 little-endian, reduced mod r."""
        v = 0
        for i in range(4):
            v |= self.next_u64() << (64 * i)
        return v % R_ORDER
