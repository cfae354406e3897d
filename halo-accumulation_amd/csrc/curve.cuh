// Pallas group law on gfx950 (y^2 = x^3 + 5 over Fq, prime order, cofactor 1; group.rs:7-8).
//
//  Aff  : affine (x, y), (0,0) encodes the point at infinity ((0,0) is not on the curve)
//  Jac  : Jacobian (X:Y:Z), x = X/Z^2, y = Y/Z^3, Z = 0 infinity  -- what ark-ec's Projective holds;
//         used for the uniform-scalar ladders (doubling is 2M+5S)
//  Xyzz : extended Jacobian (X, Y, ZZ, ZZZ), x = X/ZZ, y = Y/ZZZ, ZZ = 0 infinity -- bucket
//         accumulators (mixed add 8M+2S, full add 12M+2S)
//
// Every routine is complete: infinity operands, P + P and P + (-P) are handled, because the
// reference's results must be reproduced for adversarial inputs too (all-equal bases, +s/-s
// pairs, zero scalars; SURVEY.md section 7 "edge cases").
#pragma once
#include "field.cuh"

namespace halo {

using Q = FqCfg;

struct Aff { Fe x, y; };
struct Jac { Fe x, y, z; };
struct Xyzz { Fe x, y, zz, zzz; };

HALO_DEV bool aff_is_inf(const Aff &p) { return fe_is_zero(p.x) && fe_is_zero(p.y); }
HALO_DEV Aff aff_inf() { Aff r; r.x = fe_zero(); r.y = fe_zero(); return r; }
HALO_DEV Aff aff_cneg(const Aff &p, bool negate) {
    Aff r; r.x = p.x;
    Fe ny = fe_neg<Q>(p.y);
#pragma unroll
    for (int i = 0; i < 8; i++) r.y.v[i] = negate ? ny.v[i] : p.y.v[i];
    return r;
}
HALO_DEV Aff aff_load(const uint64_t *p) { Aff r; r.x = fe_load(p); r.y = fe_load(p + 4); return r; }
HALO_DEV void aff_store(uint64_t *p, const Aff &a) { fe_store(p, a.x); fe_store(p + 4, a.y); }

// ---------------------------------------------------------------- XYZZ
HALO_DEV bool xyzz_is_inf(const Xyzz &p) { return fe_is_zero(p.zz); }
HALO_DEV Xyzz xyzz_inf() {
    Xyzz r; r.x = fe_one<Q>(); r.y = fe_one<Q>(); r.zz = fe_zero(); r.zzz = fe_zero(); return r;
}
HALO_DEV Xyzz xyzz_from_aff(const Aff &a) {
    if (aff_is_inf(a)) return xyzz_inf();
    Xyzz r; r.x = a.x; r.y = a.y; r.zz = fe_one<Q>(); r.zzz = fe_one<Q>(); return r;
}
// dbl-2008-s-1, a = 0
HALO_DEV Xyzz xyzz_dbl(const Xyzz &p) {
    if (xyzz_is_inf(p)) return p;
    Fe U = fe_dbl<Q>(p.y);
    Fe V = fe_sqr<Q>(U);
    Fe W = fe_mul<Q>(U, V);
    Fe S = fe_mul<Q>(p.x, V);
    Fe xx = fe_sqr<Q>(p.x);
    Fe M = fe_add<Q>(fe_dbl<Q>(xx), xx);
    Xyzz r;
    r.x = fe_sub<Q>(fe_sub<Q>(fe_sqr<Q>(M), S), S);
    r.y = fe_sub<Q>(fe_mul<Q>(M, fe_sub<Q>(S, r.x)), fe_mul<Q>(W, p.y));
    r.zz = fe_mul<Q>(V, p.zz);
    r.zzz = fe_mul<Q>(W, p.zzz);
    return r;
}
// madd-2008-s: acc += q (affine)
HALO_DEV void xyzz_madd(Xyzz &acc, const Aff &q) {
    if (aff_is_inf(q)) return;
    if (xyzz_is_inf(acc)) { acc = xyzz_from_aff(q); return; }
    Fe U2 = fe_mul<Q>(q.x, acc.zz);
    Fe S2 = fe_mul<Q>(q.y, acc.zzz);
    Fe Pd = fe_sub<Q>(U2, acc.x);
    Fe Rd = fe_sub<Q>(S2, acc.y);
    if (fe_is_zero(Pd)) {
        if (fe_is_zero(Rd)) acc = xyzz_dbl(xyzz_from_aff(q));
        else acc = xyzz_inf();
        return;
    }
    Fe PP = fe_sqr<Q>(Pd);
    Fe PPP = fe_mul<Q>(Pd, PP);
    Fe Qv = fe_mul<Q>(acc.x, PP);
    Fe x3 = fe_sub<Q>(fe_sub<Q>(fe_sub<Q>(fe_sqr<Q>(Rd), PPP), Qv), Qv);
    Fe y3 = fe_sub<Q>(fe_mul<Q>(Rd, fe_sub<Q>(Qv, x3)), fe_mul<Q>(acc.y, PPP));
    acc.x = x3;
    acc.y = y3;
    acc.zz = fe_mul<Q>(acc.zz, PP);
    acc.zzz = fe_mul<Q>(acc.zzz, PPP);
}
// add-2008-s: acc += q
HALO_DEV void xyzz_add(Xyzz &acc, const Xyzz &q) {
    if (xyzz_is_inf(q)) return;
    if (xyzz_is_inf(acc)) { acc = q; return; }
    Fe U1 = fe_mul<Q>(acc.x, q.zz);
    Fe U2 = fe_mul<Q>(q.x, acc.zz);
    Fe S1 = fe_mul<Q>(acc.y, q.zzz);
    Fe S2 = fe_mul<Q>(q.y, acc.zzz);
    Fe Pd = fe_sub<Q>(U2, U1);
    Fe Rd = fe_sub<Q>(S2, S1);
    if (fe_is_zero(Pd)) {
        if (fe_is_zero(Rd)) acc = xyzz_dbl(acc);
        else acc = xyzz_inf();
        return;
    }
    Fe PP = fe_sqr<Q>(Pd);
    Fe PPP = fe_mul<Q>(Pd, PP);
    Fe Qv = fe_mul<Q>(U1, PP);
    Fe x3 = fe_sub<Q>(fe_sub<Q>(fe_sub<Q>(fe_sqr<Q>(Rd), PPP), Qv), Qv);
    Fe y3 = fe_sub<Q>(fe_mul<Q>(Rd, fe_sub<Q>(Qv, x3)), fe_mul<Q>(S1, PPP));
    acc.x = x3;
    acc.y = y3;
    acc.zz = fe_mul<Q>(fe_mul<Q>(acc.zz, q.zz), PP);
    acc.zzz = fe_mul<Q>(fe_mul<Q>(acc.zzz, q.zzz), PPP);
}
// (X, Y, ZZ, ZZZ) -> Jacobian with Z = ZZZ: X*ZZ^2, Y*ZZZ^2, ZZZ   (ZZ^3 = ZZZ^2)
HALO_DEV Jac xyzz_to_jac(const Xyzz &p) {
    Jac r;
    if (xyzz_is_inf(p)) { r.x = fe_one<Q>(); r.y = fe_one<Q>(); r.z = fe_zero(); return r; }
    r.x = fe_mul<Q>(p.x, fe_sqr<Q>(p.zz));
    r.y = fe_mul<Q>(p.y, fe_sqr<Q>(p.zzz));
    r.z = p.zzz;
    return r;
}
HALO_DEV void jac_store(uint64_t *o, const Jac &p) { fe_store(o, p.x); fe_store(o + 4, p.y); fe_store(o + 8, p.z); }
HALO_DEV Jac jac_load(const uint64_t *o) { Jac p; p.x = fe_load(o); p.y = fe_load(o + 4); p.z = fe_load(o + 8); return p; }
HALO_DEV void xyzz_store(uint64_t *o, const Xyzz &p) { fe_store(o, p.x); fe_store(o + 4, p.y); fe_store(o + 8, p.zz); fe_store(o + 12, p.zzz); }
HALO_DEV Xyzz xyzz_load(const uint64_t *o) { Xyzz p; p.x = fe_load(o); p.y = fe_load(o + 4); p.zz = fe_load(o + 8); p.zzz = fe_load(o + 12); return p; }

// ---------------------------------------------------------------- Jacobian
HALO_DEV bool jac_is_inf(const Jac &p) { return fe_is_zero(p.z); }
HALO_DEV Jac jac_inf() { Jac r; r.x = fe_one<Q>(); r.y = fe_one<Q>(); r.z = fe_zero(); return r; }
HALO_DEV Jac jac_from_aff(const Aff &a) {
    if (aff_is_inf(a)) return jac_inf();
    Jac r; r.x = a.x; r.y = a.y; r.z = fe_one<Q>(); return r;
}
// dbl-2009-l (a = 0): 2M + 5S.  Infinity in -> infinity out (Z3 = 2*Y*0).
HALO_DEV Jac jac_dbl(const Jac &p) {
    Fe A = fe_sqr<Q>(p.x);
    Fe B = fe_sqr<Q>(p.y);
    Fe C = fe_sqr<Q>(B);
    Fe t = fe_sub<Q>(fe_sub<Q>(fe_sqr<Q>(fe_add<Q>(p.x, B)), A), C);
    Fe D = fe_dbl<Q>(t);
    Fe E = fe_add<Q>(fe_dbl<Q>(A), A);
    Fe F = fe_sqr<Q>(E);
    Jac r;
    r.z = fe_dbl<Q>(fe_mul<Q>(p.y, p.z));
    r.x = fe_sub<Q>(fe_sub<Q>(F, D), D);
    Fe c8 = fe_dbl<Q>(fe_dbl<Q>(fe_dbl<Q>(C)));
    r.y = fe_sub<Q>(fe_mul<Q>(E, fe_sub<Q>(D, r.x)), c8);
    return r;
}
// madd-2007-bl: p + q (affine): 7M + 4S
HALO_DEV Jac jac_madd(const Jac &p, const Aff &q) {
    if (aff_is_inf(q)) return p;
    if (jac_is_inf(p)) return jac_from_aff(q);
    Fe Z1Z1 = fe_sqr<Q>(p.z);
    Fe U2 = fe_mul<Q>(q.x, Z1Z1);
    Fe S2 = fe_mul<Q>(fe_mul<Q>(q.y, p.z), Z1Z1);
    Fe H = fe_sub<Q>(U2, p.x);
    Fe rr = fe_sub<Q>(S2, p.y);
    if (fe_is_zero(H)) {
        if (fe_is_zero(rr)) return jac_dbl(p);
        return jac_inf();
    }
    rr = fe_dbl<Q>(rr);
    Fe HH = fe_sqr<Q>(H);
    Fe I = fe_dbl<Q>(fe_dbl<Q>(HH));
    Fe J = fe_mul<Q>(H, I);
    Fe V = fe_mul<Q>(p.x, I);
    Jac r;
    r.x = fe_sub<Q>(fe_sub<Q>(fe_sub<Q>(fe_sqr<Q>(rr), J), V), V);
    r.y = fe_sub<Q>(fe_mul<Q>(rr, fe_sub<Q>(V, r.x)), fe_dbl<Q>(fe_mul<Q>(p.y, J)));
    r.z = fe_sub<Q>(fe_sub<Q>(fe_sqr<Q>(fe_add<Q>(p.z, H)), Z1Z1), HH);
    return r;
}
// Jacobian -> affine with one Fermat inversion
HALO_DEV Aff jac_to_aff(const Jac &p) {
    if (jac_is_inf(p)) return aff_inf();
    Fe zi = fe_inv<Q>(p.z);
    Fe zi2 = fe_sqr<Q>(zi);
    Aff r;
    r.x = fe_mul<Q>(p.x, zi2);
    r.y = fe_mul<Q>(p.y, fe_mul<Q>(zi2, zi));
    return r;
}

// ---------------------------------------------------------------- cross-lane moves (wave64)
HALO_DEV Fe fe_shfl(const Fe &a, int src_lane) {
    Fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = (uint32_t)__shfl((int)a.v[i], src_lane, 64);
    return r;
}
HALO_DEV Xyzz xyzz_shfl(const Xyzz &p, int src_lane) {
    Xyzz r;
    r.x = fe_shfl(p.x, src_lane); r.y = fe_shfl(p.y, src_lane);
    r.zz = fe_shfl(p.zz, src_lane); r.zzz = fe_shfl(p.zzz, src_lane);
    return r;
}

}  // namespace halo
