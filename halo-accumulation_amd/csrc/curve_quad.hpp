// Quad-parallel XYZZ arithmetic for the latency-bound tails of an MSM (window sums of small MSMs).
//
// A lane-per-point xyzz_add is a serial string of 14 field products (~3500 instructions, ~7.7 us when the
// wave has the SIMD to itself: the issue rate of ONE wave is the bound, not the CU).  The window-sum step
// of a small MSM is a chain of ~40 dependent point operations run by a few hundred waves, so it is the
// latency of one operation that counts.  Here the 4 lanes of a quad hold the SAME point (replicated) and
// share one operation: the products of an addition form 4 dependency levels of <= 4 independent products
// each (3 levels for a doubling), so every lane runs one product per level on operands picked by its
// position in the quad (v_cndmask), and the four results are broadcast back with DPP quad_perm moves
// (full rate, no LDS).  An addition costs 4 products + ~400 cheap instructions instead of 14 products.
//
// All routines must be called in wave-uniform control flow (DPP reads lanes regardless of their position in
// the quad, so every lane of a quad has to be executing); per-quad special cases (infinity operands,
// P + P, P + (-P)) are resolved with selects, and the rare doubling inside an addition is taken by the
// whole wave when any quad needs it.  Same formulas, same value bounds (Fq<K>) as curve.hpp.
#pragma once
#include "curve.hpp"

namespace halo {

// The empty asm keeps the broadcast a v_mov_b32_dpp of its own.  Without it the DPP-combine pass of this compiler
// (ROCm 7.2) folds the move into the consuming VALU instruction, and where BOTH operands of a subtraction were
// broadcasts of the same register with different lane selectors (y3 = bcast<0>(r) - bcast<1>(r)) it dropped one of the
// two permutations: lanes 0, 2, 3 of every quad computed r - r.  Found with tools/quad_test.hip.
template <int SRC>
HALO_DEV uint32_t quad_bcast_u32(uint32_t v) {
    uint32_t r = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, SRC * 0x55, 0xf, 0xf, true);  // quad_perm:[SRC,SRC,SRC,SRC]
    asm volatile("" : "+v"(r));
    return r;
}
template <int SRC, int K>
HALO_DEV Fq<K> quad_bcast(const Fq<K> &a) {
    Fq<K> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = quad_bcast_u32<SRC>(a.v[i]);
    return r;
}
// lane position ql in 0..3 picks a0..a3
template <int K>
HALO_DEV Fq<K> quad_sel(int ql, const Fq<K> &a0, const Fq<K> &a1, const Fq<K> &a2, const Fq<K> &a3) {
    Fq<K> r;
    bool lo = (ql & 2) == 0, even = (ql & 1) == 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        uint32_t x = even ? a0.v[i] : a1.v[i], y = even ? a2.v[i] : a3.v[i];
        r.v[i] = lo ? x : y;
    }
    return r;
}
HALO_DEV XyzzN xyzz_select(bool take_a, const XyzzN &a, const XyzzN &b) {
    XyzzN r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        r.x.v[i] = take_a ? a.x.v[i] : b.x.v[i];
        r.y.v[i] = take_a ? a.y.v[i] : b.y.v[i];
        r.zz.v[i] = take_a ? a.zz.v[i] : b.zz.v[i];
        r.zzz.v[i] = take_a ? a.zzz.v[i] : b.zzz.v[i];
    }
    return r;
}

// dbl-2008-s-1 (a = 0) over a quad: levels {Y^2, X^2} -> {Y V, X V, M^2, V ZZ} -> {M (S - X3), W Y, W ZZZ}
HALO_DEV XyzzN xyzz_dbl_quad(const XyzzN &p, int ql) {
    Fq<8> a1 = quad_sel(ql, p.y, p.x, p.y, p.x);
    Fq<2> r1 = fq_mul(a1, a1);
    Fq<8> V = fq_muls<4>(quad_bcast<0>(r1));
    Fq<6> M = fq_muls<3>(quad_bcast<1>(r1));
    Fq<8> a2 = quad_sel(ql, p.y, p.x, fq_widen<8>(M), V);
    Fq<8> b2 = quad_sel(ql, V, V, fq_widen<8>(M), fq_widen<8>(p.zz));
    Fq<2> r2 = fq_mul(a2, b2);
    Fq<4> W = fq_muls<2>(quad_bcast<0>(r2));
    Fq<2> S = quad_bcast<1>(r2), MM = quad_bcast<2>(r2), ZZ3 = quad_bcast<3>(r2);
    Fq<6> x3 = fq_sub<4>(MM, fq_muls<2>(S));
    Fq<6> a3 = quad_sel(ql, M, fq_widen<6>(W), fq_widen<6>(W), fq_widen<6>(W));
    Fq<10> b3 = quad_sel(ql, fq_sub<8>(S, x3), fq_widen<10>(p.y), fq_widen<10>(p.zzz), fq_widen<10>(p.zzz));
    Fq<2> r3 = fq_mul(a3, b3);
    XyzzN r;
    r.x = fq_widen<8>(x3);
    r.y = fq_widen<8>(fq_sub<2>(quad_bcast<0>(r3), quad_bcast<1>(r3)));
    r.zz = ZZ3;
    r.zzz = quad_bcast<2>(r3);
    return xyzz_select(xyzz_is_inf(p), xyzz_inf(), r);
}

// add-2008-s over a quad: acc += q.  Levels {U1, U2, S1, S2} -> {P^2, R^2, ZZ1 ZZ2, ZZZ1 ZZZ2} -> {P PP, U1 PP,
// ZZ12 PP, ZZZ12 PP} -> {R (Q - X3), S1 PPP, (ZZZ12 PP) P}.
HALO_DEV void xyzz_add_quad(XyzzN &acc, const XyzzN &q, int ql) {
    bool acc_inf = xyzz_is_inf(acc), q_inf = xyzz_is_inf(q);
    Fq<8> a1 = quad_sel(ql, acc.x, q.x, acc.y, q.y);
    Fq<2> b1 = quad_sel(ql, q.zz, acc.zz, q.zzz, acc.zzz);
    Fq<2> r1 = fq_mul(a1, b1);
    Fq<2> U1 = quad_bcast<0>(r1), U2 = quad_bcast<1>(r1), S1 = quad_bcast<2>(r1), S2 = quad_bcast<3>(r1);
    Fq<4> Pd = fq_sub<2>(U2, U1), Rd = fq_sub<2>(S2, S1);
    bool both = !acc_inf && !q_inf;
    bool p_zero = both && fq_is_zero_modp(Pd);
    bool r_zero = p_zero && fq_is_zero_modp(Rd);
    Fq<4> a2 = quad_sel(ql, Pd, Rd, fq_widen<4>(acc.zz), fq_widen<4>(acc.zzz));
    Fq<4> b2 = quad_sel(ql, Pd, Rd, fq_widen<4>(q.zz), fq_widen<4>(q.zzz));
    Fq<2> r2 = fq_mul(a2, b2);
    Fq<2> PP = quad_bcast<0>(r2), RR = quad_bcast<1>(r2), ZZ12 = quad_bcast<2>(r2), ZZZ12 = quad_bcast<3>(r2);
    Fq<4> a3 = quad_sel(ql, Pd, fq_widen<4>(U1), fq_widen<4>(ZZ12), fq_widen<4>(ZZZ12));
    Fq<2> r3 = fq_mul(a3, PP);
    Fq<2> PPP = quad_bcast<0>(r3), Qv = quad_bcast<1>(r3), ZZ3 = quad_bcast<2>(r3), Wv = quad_bcast<3>(r3);
    Fq<8> x3 = fq_sub_sub2(RR, PPP, Qv);
    Fq<4> a4 = quad_sel(ql, Rd, fq_widen<4>(S1), fq_widen<4>(Wv), fq_widen<4>(Wv));
    Fq<10> b4 = quad_sel(ql, fq_sub<8>(Qv, x3), fq_widen<10>(PPP), fq_widen<10>(Pd), fq_widen<10>(Pd));
    Fq<2> r4 = fq_mul(a4, b4);
    XyzzN r;
    r.x = x3;
    r.y = fq_widen<8>(fq_sub<2>(quad_bcast<0>(r4), quad_bcast<1>(r4)));
    r.zz = ZZ3;
    r.zzz = quad_bcast<2>(r4);
    if (__any(r_zero ? 1 : 0)) {  // P + P somewhere in the wave (equal bucket values: adversarial inputs only)
        XyzzN d = xyzz_dbl_quad(q, ql);
        r = xyzz_select(r_zero, d, r);
    }
    r = xyzz_select(p_zero && !r_zero, xyzz_inf(), r);  // P + (-P)
    r = xyzz_select(q_inf, acc, r);
    r = xyzz_select(acc_inf, q, r);
    acc = r;
}

// ---------------------------------------------------------------- Jacobian ladders over a quad (uniform-scalar folds of small keys)
HALO_DEV JacN jac_select(bool take_a, const JacN &a, const JacN &b) {
    JacN r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        r.x.v[i] = take_a ? a.x.v[i] : b.x.v[i];
        r.y.v[i] = take_a ? a.y.v[i] : b.y.v[i];
        r.z.v[i] = take_a ? a.z.v[i] : b.z.v[i];
    }
    return r;
}
// dbl-2009-l (a = 0) as jac_dbl: levels {X^2, Y^2, Y Z} -> {B^2, X B, E^2} -> {E (D - X3)}.  Infinity in -> infinity out.
HALO_DEV JacN jac_dbl_quad(const JacN &p, int ql) {
    Fq<8> a1 = quad_sel(ql, p.x, p.y, p.y, p.y);
    Fq<8> b1 = quad_sel(ql, p.x, p.y, fq_widen<8>(p.z), fq_widen<8>(p.z));
    Fq<2> r1 = fq_mul(a1, b1);
    Fq<2> A = quad_bcast<0>(r1), B = quad_bcast<1>(r1), YZ = quad_bcast<2>(r1);
    Fq<6> E = fq_muls<3>(A);
    Fq<8> a2 = quad_sel(ql, fq_widen<8>(B), p.x, fq_widen<8>(E), fq_widen<8>(E));
    Fq<6> b2 = quad_sel(ql, fq_widen<6>(B), fq_widen<6>(B), E, E);
    Fq<2> r2 = fq_mul(a2, b2);
    Fq<2> C = quad_bcast<0>(r2), F = quad_bcast<2>(r2);
    Fq<8> D = fq_muls<4>(quad_bcast<1>(r2));
    Fq<2> x3 = fq_tighten(fq_sub<16>(F, fq_muls<2>(D)));
    Fq<2> c8 = fq_tighten(fq_muls<8>(C));
    Fq<2> r3 = fq_mul(E, fq_sub<2>(D, x3));  // every lane computes the same product: no selection needed
    JacN r;
    r.x = fq_widen<8>(x3);
    r.y = fq_widen<8>(fq_sub<2>(r3, c8));
    r.z = fq_muls<2>(YZ);
    return r;
}
// madd-2007-bl as jac_madd, q = (beta^e x, +-y) given as (x, y, bconst): levels {Z^2, y Z, x beta^e} -> {U2, S2} ->
// {H^2, r^2, Z H} -> {H I, X1 I} -> {r (V - X3), Y1 J}.  `live` false (q at infinity) leaves p unchanged.
HALO_DEV JacN jac_madd_quad(const JacN &p, const Fq<2> &qx, const Fq<2> &qy, const Fq<2> &bconst, bool live, int ql) {
    bool p_inf = jac_is_inf(p);
    Fq<4> a1 = quad_sel(ql, p.z, fq_widen<4>(qy), fq_widen<4>(qx), fq_widen<4>(qx));
    Fq<4> b1 = quad_sel(ql, p.z, p.z, fq_widen<4>(bconst), fq_widen<4>(bconst));
    Fq<2> r1 = fq_mul(a1, b1);
    Fq<2> Z1Z1 = quad_bcast<0>(r1), YZ = quad_bcast<1>(r1), X2 = quad_bcast<2>(r1);
    Fq<2> a2 = quad_sel(ql, X2, YZ, X2, YZ);
    Fq<2> r2 = fq_mul(a2, Z1Z1);
    Fq<2> U2 = quad_bcast<0>(r2), S2 = quad_bcast<1>(r2);
    Fq<10> H = fq_sub<8>(U2, p.x), r0 = fq_sub<8>(S2, p.y);
    bool h_zero = live && !p_inf && fq_is_zero_modp(H);
    bool r_zero = h_zero && fq_is_zero_modp(r0);
    Fq<10> a3 = quad_sel(ql, H, r0, fq_widen<10>(p.z), fq_widen<10>(p.z));
    Fq<10> b3 = quad_sel(ql, H, r0, H, H);
    Fq<2> r3 = fq_mul(a3, b3);
    Fq<8> I = fq_muls<4>(quad_bcast<0>(r3));
    Fq<2> RR = quad_bcast<1>(r3), ZH = quad_bcast<2>(r3);
    Fq<10> a4 = quad_sel(ql, H, fq_widen<10>(p.x), H, fq_widen<10>(p.x));
    Fq<2> r4 = fq_mul(a4, I);
    Fq<2> J = quad_bcast<0>(r4), V = quad_bcast<1>(r4);
    Fq<2> x3 = fq_tighten(fq_sub_sub2(fq_muls<4>(RR), J, V));
    Fq<10> a5 = quad_sel(ql, r0, fq_widen<10>(p.y), r0, fq_widen<10>(p.y));
    Fq<4> b5 = quad_sel(ql, fq_sub<2>(V, x3), fq_widen<4>(J), fq_sub<2>(V, x3), fq_widen<4>(J));
    Fq<2> r5 = fq_mul(a5, b5);
    JacN r;
    r.x = fq_widen<8>(x3);
    r.y = fq_muls<2>(fq_sub<2>(quad_bcast<0>(r5), quad_bcast<1>(r5)));
    r.z = fq_muls<2>(ZH);
    if (__any(r_zero ? 1 : 0)) {  // P + P: only where a folded key repeats a point
        JacN d = jac_dbl_quad(p, ql);
        r = jac_select(r_zero, d, r);
    }
    r = jac_select(h_zero && !r_zero, jac_inf(), r);  // P + (-P)
    JacN from_q;                                       // infinity + q
    from_q.x = fq_widen<8>(X2); from_q.y = fq_widen<8>(qy); from_q.z = fq_widen<4>(fq_one());
    r = jac_select(p_inf, from_q, r);
    return jac_select(live, r, p);
}

// memory: every lane of the quad reads the whole point (same addresses: one transaction); lane ql stores coordinate ql
HALO_DEV void xyzz_store_quad(uint32_t *o, const XyzzN &p, int ql) {
    Fq<8> c = quad_sel(ql, p.x, p.y, fq_widen<8>(p.zz), fq_widen<8>(p.zzz));
    fq_store_native(o + 10 * ql, c);
}

}  // namespace halo
