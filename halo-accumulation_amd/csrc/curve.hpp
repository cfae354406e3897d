// Pallas group law on gfx950 (y^2 = x^3 + 5 over Fq, prime order, cofactor 1; group.rs:7-8)
// over the lazy radix-2^29 field of fq29.hpp.  The integer template argument of Fq<K> is the
// proven bound "value < K*p"; the invariants of the point types below are what makes every
// product satisfy Ka*Kb <= 120 (checked at compile time):
//
//  AffN  : affine (x, y), x, y < 2p.  All-zero limbs = point at infinity.  This is also the
//          layout of the base tables in HBM: 20 words per point (9 limbs + 1 pad per coordinate).
//  XyzzN : extended Jacobian (X, Y, ZZ, ZZZ), x = X/ZZ, y = Y/ZZZ; X, Y < 8p, ZZ, ZZZ < 2p.
//          ZZ with all-zero limbs = infinity.  Bucket accumulators: mixed add 8M + 2S, no
//          reduction step anywhere in it.
//  JacN  : Jacobian (X:Y:Z); X, Y < 8p, Z < 4p; Z all-zero limbs = infinity.  Uniform-scalar
//          ladders (doubling 3M + 4S).
//
// Every routine is complete: infinity operands, P + P and P + (-P) are handled, because the
// reference's results must be reproduced for adversarial inputs too (all-equal bases, +s/-s
// pairs, zero scalars).  The P = +-Q tests are exact: k*p = k (mod 2^29), so a cheap test on
// limb 0 filters all but ~K/2^29 of the cases before the full reduction.
#pragma once
#include <type_traits>

#include "fq29.hpp"

namespace halo {

using Q = FqCfg;

struct AffN { Fq<2> x, y; };
struct XyzzN { Fq<8> x, y; Fq<2> zz, zzz; };
struct JacN { Fq<8> x, y; Fq<4> z; };

constexpr int AFF_WORDS = 20;   // words of a native affine point (9 limbs + pad, twice)
constexpr int AFF_STRIDE = 32;  // table stride in words: 128 B, so a gather touches exactly one 128-byte line
constexpr int XYZZ_WORDS = 40;  // native XYZZ point in memory

// ---------------------------------------------------------------- affine
HALO_DEV bool aff_is_inf(const AffN &p) { return fq_limbs_zero(p.x) && fq_limbs_zero(p.y); }
HALO_DEV AffN aff_inf() { AffN r; r.x = fq_zero<2>(); r.y = fq_zero<2>(); return r; }
HALO_DEV AffN aff_cneg(const AffN &p, bool negate) {
    AffN r; r.x = p.x;
    Fq<2> ny = fq_neg<2>(p.y);
    negate = negate && !fq_limbs_zero(p.y);  // the (0,0) infinity marker must stay (0,0)
#pragma unroll
    for (int i = 0; i < 9; i++) r.y.v[i] = negate ? ny.v[i] : p.y.v[i];
    return r;
}
HALO_DEV AffN aff_load(const uint32_t *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    uint4 a = q[0], b = q[1], c = q[2], d = q[3], e = q[4];
    AffN r;
    r.x.v[0] = a.x; r.x.v[1] = a.y; r.x.v[2] = a.z; r.x.v[3] = a.w;
    r.x.v[4] = b.x; r.x.v[5] = b.y; r.x.v[6] = b.z; r.x.v[7] = b.w;
    r.x.v[8] = c.x;
    r.y.v[0] = c.z; r.y.v[1] = c.w;
    r.y.v[2] = d.x; r.y.v[3] = d.y; r.y.v[4] = d.z; r.y.v[5] = d.w;
    r.y.v[6] = e.x; r.y.v[7] = e.y; r.y.v[8] = e.z;
    return r;
}
// A native affine point owns a whole 128-byte line: x at words 0..8, y at 10..18 and -y at 20..28 (the signed-digit bucket
// kernels pick y or -y by ADDRESS instead of negating in registers: ~45 of the ~1970 instructions of an addition)
HALO_DEV void aff_store(uint32_t *p, const AffN &a) {
    uint4 *q = reinterpret_cast<uint4 *>(p);
    q[0] = make_uint4(a.x.v[0], a.x.v[1], a.x.v[2], a.x.v[3]);
    q[1] = make_uint4(a.x.v[4], a.x.v[5], a.x.v[6], a.x.v[7]);
    q[2] = make_uint4(a.x.v[8], 0u, a.y.v[0], a.y.v[1]);
    q[3] = make_uint4(a.y.v[2], a.y.v[3], a.y.v[4], a.y.v[5]);
    q[4] = make_uint4(a.y.v[6], a.y.v[7], a.y.v[8], 0u);
    Fq<2> ny = fq_neg<2>(a.y);
    if (fq_limbs_zero(a.y)) ny = fq_zero<2>();  // the (0, 0) infinity marker reads (0, 0) with either sign
    q[5] = make_uint4(ny.v[0], ny.v[1], ny.v[2], ny.v[3]);
    q[6] = make_uint4(ny.v[4], ny.v[5], ny.v[6], ny.v[7]);
    q[7] = make_uint4(ny.v[8], 0u, 0u, 0u);
}
// (x, y) or (x, -y): the sign picks which stored copy of y is read (both 8-byte aligned: words 10.. and 20..)
HALO_DEV AffN aff_load_signed(const uint32_t *p, bool negate) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    uint4 a = q[0], b = q[1];
    uint32_t x8 = p[8];
    const uint2 *py = reinterpret_cast<const uint2 *>(p + (negate ? 20 : 10));
    uint2 y0 = py[0], y1 = py[1], y2 = py[2], y3 = py[3];
    uint32_t y8 = reinterpret_cast<const uint32_t *>(py)[8];
    AffN r;
    r.x.v[0] = a.x; r.x.v[1] = a.y; r.x.v[2] = a.z; r.x.v[3] = a.w;
    r.x.v[4] = b.x; r.x.v[5] = b.y; r.x.v[6] = b.z; r.x.v[7] = b.w;
    r.x.v[8] = x8;
    r.y.v[0] = y0.x; r.y.v[1] = y0.y; r.y.v[2] = y1.x; r.y.v[3] = y1.y; r.y.v[4] = y2.x; r.y.v[5] = y2.y;
    r.y.v[6] = y3.x; r.y.v[7] = y3.y; r.y.v[8] = y8;
    return r;
}
// arkworks affine words (x | y, 8 u64, (0,0) = infinity) <-> native
HALO_DEV AffN aff_from_words(const uint64_t *w) {
    Fe x = fe_load(w), y = fe_load(w + 4);
    if (fe_is_zero(x) && fe_is_zero(y)) return aff_inf();
    AffN r; r.x = fq_from_words(x); r.y = fq_from_words(y); return r;
}
HALO_DEV void aff_to_words(uint64_t *w, const AffN &a) {
    if (aff_is_inf(a)) { fe_store(w, fe_zero()); fe_store(w + 4, fe_zero()); return; }
    fe_store(w, fq_to_words(a.x));
    fe_store(w + 4, fq_to_words(a.y));
}

// ---------------------------------------------------------------- XYZZ
HALO_DEV bool xyzz_is_inf(const XyzzN &p) { return fq_limbs_zero(p.zz); }
HALO_DEV XyzzN xyzz_inf() {
    XyzzN r; r.x = fq_zero<8>(); r.y = fq_zero<8>(); r.zz = fq_zero<2>(); r.zzz = fq_zero<2>(); return r;
}
HALO_DEV XyzzN xyzz_from_aff(const AffN &a) {
    if (aff_is_inf(a)) return xyzz_inf();
    XyzzN r; r.x = fq_widen<8>(a.x); r.y = fq_widen<8>(a.y);
    r.zz = fq_widen<2>(fq_one()); r.zzz = r.zz; return r;
}
// dbl-2008-s-1, a = 0 (V = 4Y^2, W = 2Y*V, S = X*V, M = 3X^2).  Infinity in -> infinity out.
HALO_DEV XyzzN xyzz_dbl(const XyzzN &p) {
    Fq<8> V = fq_muls<4>(fq_sqr(p.y));
    Fq<4> W = fq_muls<2>(fq_mul(p.y, V));
    Fq<2> S = fq_mul(p.x, V);
    Fq<6> M = fq_muls<3>(fq_sqr(p.x));
    XyzzN r;
    Fq<6> x3 = fq_sub<4>(fq_sqr(M), fq_muls<2>(S));
    r.x = fq_widen<8>(x3);
    r.y = fq_widen<8>(fq_sub<2>(fq_mul(M, fq_sub<8>(S, x3)), fq_mul(W, p.y)));
    r.zz = fq_mul(V, p.zz);
    r.zzz = fq_mul(W, p.zzz);
    if (xyzz_is_inf(p)) return xyzz_inf();
    return r;
}
// madd-2008-s: acc += q (affine).  8M + 2S, five carry passes, no reduction.
HALO_DEV void xyzz_madd(XyzzN &acc, const AffN &q) {
    if (aff_is_inf(q)) return;
    if (xyzz_is_inf(acc)) { acc = xyzz_from_aff(q); return; }
    Fq<2> U2 = fq_mul(q.x, acc.zz);
    Fq<2> S2 = fq_mul(q.y, acc.zzz);
    Fq<10> Pd = fq_sub<8>(U2, acc.x);
    Fq<10> Rd = fq_sub<8>(S2, acc.y);
    if (fq_is_zero_modp(Pd)) {
        if (fq_is_zero_modp(Rd)) acc = xyzz_dbl(xyzz_from_aff(q));
        else acc = xyzz_inf();
        return;
    }
    Fq<2> PP = fq_sqr(Pd);
    Fq<2> PPP = fq_mul(Pd, PP);
    Fq<2> Qv = fq_mul(acc.x, PP);
    Fq<8> x3 = fq_sub_sub2(fq_sqr(Rd), PPP, Qv);
    // y3 = R (Q - x3) - Y1 PPP as ONE reduction: R (Q - x3) + (8p - Y1) PPP
    Fq<2> y3 = fq_mul_add_mul(Rd, fq_sub<8>(Qv, x3), fq_neg<8>(acc.y), PPP);
    acc.x = x3;
    acc.y = fq_widen<8>(y3);
    acc.zz = fq_mul(acc.zz, PP);
    acc.zzz = fq_mul(acc.zzz, PPP);
}
// add-2008-s: acc += q.  12M + 2S.
HALO_DEV void xyzz_add(XyzzN &acc, const XyzzN &q) {
    if (xyzz_is_inf(q)) return;
    if (xyzz_is_inf(acc)) { acc = q; return; }
    Fq<2> U1 = fq_mul(acc.x, q.zz);
    Fq<2> U2 = fq_mul(q.x, acc.zz);
    Fq<2> S1 = fq_mul(acc.y, q.zzz);
    Fq<2> S2 = fq_mul(q.y, acc.zzz);
    Fq<4> Pd = fq_sub<2>(U2, U1);
    Fq<4> Rd = fq_sub<2>(S2, S1);
    if (fq_is_zero_modp(Pd)) {
        if (fq_is_zero_modp(Rd)) acc = xyzz_dbl(acc);
        else acc = xyzz_inf();
        return;
    }
    Fq<2> PP = fq_sqr(Pd);
    Fq<2> PPP = fq_mul(Pd, PP);
    Fq<2> Qv = fq_mul(U1, PP);
    Fq<8> x3 = fq_sub_sub2(fq_sqr(Rd), PPP, Qv);
    Fq<2> y3 = fq_mul_add_mul(Rd, fq_sub<8>(Qv, x3), fq_neg<2>(S1), PPP);
    acc.x = x3;
    acc.y = fq_widen<8>(y3);
    acc.zz = fq_mul(fq_mul(acc.zz, q.zz), PP);
    acc.zzz = fq_mul(fq_mul(acc.zzz, q.zzz), PPP);
}
// native XYZZ in memory: x | y | zz | zzz, 10 words each
HALO_DEV void xyzz_store(uint32_t *o, const XyzzN &p) {
    fq_store_native(o, p.x); fq_store_native(o + 10, p.y); fq_store_native(o + 20, p.zz); fq_store_native(o + 30, p.zzz);
}
template <int K>
HALO_DEV Fq<K> fq_load_native(const uint32_t *p) {
    Fq<K> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = p[i];
    return r;
}
HALO_DEV XyzzN xyzz_load(const uint32_t *o) {
    XyzzN p;
    p.x = fq_load_native<8>(o); p.y = fq_load_native<8>(o + 10); p.zz = fq_load_native<2>(o + 20); p.zzz = fq_load_native<2>(o + 30);
    return p;
}
// (X, Y, ZZ, ZZZ) -> arkworks Jacobian words with Z = ZZZ: X*ZZ^2, Y*ZZZ^2, ZZZ   (ZZ^3 = ZZZ^2)
HALO_DEV void xyzz_store_jac_words(uint64_t *o, const XyzzN &p) {
    if (xyzz_is_inf(p)) {
        fe_store(o, fe_one<Q>()); fe_store(o + 4, fe_one<Q>()); fe_store(o + 8, fe_zero());
        return;
    }
    fe_store(o, fq_to_words(fq_mul(p.x, fq_sqr(p.zz))));
    fe_store(o + 4, fq_to_words(fq_mul(p.y, fq_sqr(p.zzz))));
    fe_store(o + 8, fq_to_words(p.zzz));
}

// ---------------------------------------------------------------- Jacobian
HALO_DEV bool jac_is_inf(const JacN &p) { return fq_limbs_zero(p.z); }
HALO_DEV JacN jac_inf() { JacN r; r.x = fq_zero<8>(); r.y = fq_zero<8>(); r.z = fq_zero<4>(); return r; }
HALO_DEV JacN jac_from_aff(const AffN &a) {
    if (aff_is_inf(a)) return jac_inf();
    JacN r; r.x = fq_widen<8>(a.x); r.y = fq_widen<8>(a.y); r.z = fq_widen<4>(fq_one()); return r;
}
// dbl-2009-l with D = 4*X*Y^2 computed as a product: 3M + 4S.  Infinity in -> infinity out (Z3 = 2*Y*0).
HALO_DEV JacN jac_dbl(const JacN &p) {
    Fq<2> A = fq_sqr(p.x);
    Fq<2> B = fq_sqr(p.y);
    Fq<2> C = fq_sqr(B);
    Fq<8> D = fq_muls<4>(fq_mul(p.x, B));
    Fq<6> E = fq_muls<3>(A);
    Fq<2> F = fq_sqr(E);
    JacN r;
    Fq<2> x3 = fq_tighten(fq_sub<16>(F, fq_muls<2>(D)));
    Fq<2> c8 = fq_tighten(fq_muls<8>(C));
    r.x = fq_widen<8>(x3);
    r.y = fq_widen<8>(fq_sub<2>(fq_mul(E, fq_sub<2>(D, x3)), c8));
    r.z = fq_muls<2>(fq_mul(p.y, p.z));
    return r;
}
// madd-2007-bl with Z3 = 2*Z1*H and r = 2*r0 factored: 8M + 3S
HALO_DEV JacN jac_madd(const JacN &p, const AffN &q) {
    if (aff_is_inf(q)) return p;
    if (jac_is_inf(p)) return jac_from_aff(q);
    Fq<2> Z1Z1 = fq_sqr(p.z);
    Fq<2> U2 = fq_mul(q.x, Z1Z1);
    Fq<2> S2 = fq_mul(fq_mul(q.y, p.z), Z1Z1);
    Fq<10> H = fq_sub<8>(U2, p.x);
    Fq<10> r0 = fq_sub<8>(S2, p.y);
    if (fq_is_zero_modp(H)) {
        if (fq_is_zero_modp(r0)) return jac_dbl(p);
        return jac_inf();
    }
    Fq<8> I = fq_muls<4>(fq_sqr(H));
    Fq<2> J = fq_mul(H, I);
    Fq<2> V = fq_mul(p.x, I);
    Fq<2> x3 = fq_tighten(fq_sub_sub2(fq_muls<4>(fq_sqr(r0)), J, V));
    JacN r;
    r.x = fq_widen<8>(x3);
    r.y = fq_widen<8>(fq_muls<2>(fq_mul_add_mul(r0, fq_sub<2>(V, x3), fq_neg<8>(p.y), J)));  // 2 (r0 (V - x3) - Y1 J), one reduction
    r.z = fq_muls<2>(fq_mul(p.z, H));
    return r;
}
// Jacobian -> affine with one Fermat inversion
HALO_DEV AffN jac_to_aff(const JacN &p) {
    if (jac_is_inf(p)) return aff_inf();
    Fq<2> zi = fq_inv(p.z);
    Fq<2> zi2 = fq_sqr(zi);
    AffN r;
    r.x = fq_mul(p.x, zi2);
    r.y = fq_mul(p.y, fq_mul(zi2, zi));
    return r;
}
// compile-time loop: f(integral_constant<int, I>) for I = 0 .. E-1 (up) or E-1 .. 0 (down).  The field products contain
// inline asm, which the loop unroller will not duplicate under a condition: arrays of points indexed by a loop variable
// then stay in scratch memory, which the build gate refuses (check_resources.py).
template <int I, int E, class F>
HALO_DEV void static_for(F &&f) {
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, E>(f);
    }
}
template <int I, class F>
HALO_DEV void static_for_down(F &&f) {
    if constexpr (I >= 0) {
        f(std::integral_constant<int, I>{});
        static_for_down<I - 1>(f);
    }
}
// E Jacobian points -> affine with ONE Fermat inversion (Montgomery's trick: prefix products of the Z's, one inverse,
// then back down): 3 (E - 1) products instead of E - 1 more inversions of ~320 products each.  Infinity stays infinity
// (its Z is replaced by 1 in the running product).
template <int E>
HALO_DEV void jac_batch_to_aff(const JacN (&p)[E], AffN (&out)[E]) {
    Fq<2> pre[E];  // pre[i] = z_0 ... z_i (infinities skipped)
    static_for<0, E>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        Fq<2> z = jac_is_inf(p[i]) ? fq_widen<2>(fq_one()) : fq_tighten(p[i].z);
        if constexpr (i == 0) pre[0] = z;
        else pre[i] = fq_mul(pre[i - 1], z);
    });
    Fq<2> inv = fq_inv(pre[E - 1]);
    static_for_down<E - 1>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        Fq<2> z = jac_is_inf(p[i]) ? fq_widen<2>(fq_one()) : fq_tighten(p[i].z);
        Fq<2> zi = inv;
        if constexpr (i > 0) { zi = fq_mul(inv, pre[i - 1]); inv = fq_mul(inv, z); }
        Fq<2> zi2 = fq_sqr(zi);
        out[i].x = fq_mul(p[i].x, zi2);
        out[i].y = fq_mul(p[i].y, fq_mul(zi2, zi));
        if (jac_is_inf(p[i])) out[i] = aff_inf();
    });
}
// arkworks Jacobian words <-> native
HALO_DEV JacN jac_from_words(const uint64_t *o) {
    Fe z = fe_load(o + 8);
    if (fe_is_zero(z)) return jac_inf();
    JacN r;
    r.x = fq_widen<8>(fq_from_words(fe_load(o)));
    r.y = fq_widen<8>(fq_from_words(fe_load(o + 4)));
    r.z = fq_widen<4>(fq_from_words(z));
    return r;
}
HALO_DEV void jac_store_words(uint64_t *o, const JacN &p) {
    if (jac_is_inf(p)) {
        fe_store(o, fe_one<Q>()); fe_store(o + 4, fe_one<Q>()); fe_store(o + 8, fe_zero());
        return;
    }
    fe_store(o, fq_to_words(p.x)); fe_store(o + 4, fq_to_words(p.y)); fe_store(o + 8, fq_to_words(p.z));
}
// (X, Y, ZZ, ZZZ) -> Jacobian with Z = ZZZ: (X ZZ^2, Y ZZZ^2, ZZZ)   (ZZ^3 = ZZZ^2), as xyzz_store_jac_words
HALO_DEV JacN xyzz_to_jac(const XyzzN &p) {
    if (xyzz_is_inf(p)) return jac_inf();
    JacN r;
    r.x = fq_widen<8>(fq_mul(p.x, fq_sqr(p.zz)));
    r.y = fq_widen<8>(fq_mul(p.y, fq_sqr(p.zzz)));
    r.z = fq_widen<4>(p.zzz);
    return r;
}
HALO_DEV XyzzN jac_to_xyzz(const JacN &p) {
    if (jac_is_inf(p)) return xyzz_inf();
    XyzzN r; r.x = p.x; r.y = p.y; r.zz = fq_sqr(p.z); r.zzz = fq_mul(r.zz, p.z); return r;
}

// ---------------------------------------------------------------- cross-lane moves (wave64)
template <int K>
HALO_DEV Fq<K> fq_shfl(const Fq<K> &a, int src_lane) {
    Fq<K> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = (uint32_t)__shfl((int)a.v[i], src_lane, 64);
    return r;
}
HALO_DEV XyzzN xyzz_shfl(const XyzzN &p, int src_lane) {
    XyzzN r;
    r.x = fq_shfl(p.x, src_lane); r.y = fq_shfl(p.y, src_lane);
    r.zz = fq_shfl(p.zz, src_lane); r.zzz = fq_shfl(p.zzz, src_lane);
    return r;
}
HALO_DEV Fe fe_shfl(const Fe &a, int src_lane) {
    Fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = (uint32_t)__shfl((int)a.v[i], src_lane, 64);
    return r;
}

}  // namespace halo
