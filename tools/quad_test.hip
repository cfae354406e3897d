// development aid: quad-parallel XYZZ ops against the lane-serial ones, on the device
#include "../halo-accumulation_amd/csrc/curve_quad.hpp"
#include <cstdio>
using namespace halo;
__device__ bool same(const XyzzN &a, const XyzzN &b) {
    // compare as affine: x = X/ZZ, y = Y/ZZZ  <=>  X1 ZZ2 == X2 ZZ1, Y1 ZZZ2 == Y2 ZZZ1
    if (xyzz_is_inf(a) || xyzz_is_inf(b)) return xyzz_is_inf(a) && xyzz_is_inf(b);
    return fq_eq_modp(fq_mul(a.x, b.zz), fq_mul(b.x, a.zz)) && fq_eq_modp(fq_mul(a.y, b.zzz), fq_mul(b.y, a.zzz));
}
__device__ unsigned dbg_add(const XyzzN &acc, const XyzzN &q, int ql) {
    unsigned ok = 0;
    Fq<2> sU1 = fq_mul(acc.x, q.zz), sU2 = fq_mul(q.x, acc.zz), sS1 = fq_mul(acc.y, q.zzz), sS2 = fq_mul(q.y, acc.zzz);
    Fq<8> a1 = quad_sel(ql, acc.x, q.x, acc.y, q.y);
    Fq<2> b1 = quad_sel(ql, q.zz, acc.zz, q.zzz, acc.zzz);
    Fq<2> r1 = fq_mul(a1, b1);
    Fq<2> U1 = quad_bcast<0>(r1), U2 = quad_bcast<1>(r1), S1 = quad_bcast<2>(r1), S2 = quad_bcast<3>(r1);
    if (fq_eq_modp(U1, sU1) && fq_eq_modp(U2, sU2) && fq_eq_modp(S1, sS1) && fq_eq_modp(S2, sS2)) ok |= 1;
    Fq<4> Pd = fq_sub<2>(U2, U1), Rd = fq_sub<2>(S2, S1);
    Fq<4> sPd = fq_sub<2>(sU2, sU1), sRd = fq_sub<2>(sS2, sS1);
    if (fq_eq_modp(Pd, sPd) && fq_eq_modp(Rd, sRd)) ok |= 2;
    Fq<4> a2 = quad_sel(ql, Pd, Rd, fq_widen<4>(acc.zz), fq_widen<4>(acc.zzz));
    Fq<4> b2 = quad_sel(ql, Pd, Rd, fq_widen<4>(q.zz), fq_widen<4>(q.zzz));
    Fq<2> r2 = fq_mul(a2, b2);
    Fq<2> PP = quad_bcast<0>(r2), RR = quad_bcast<1>(r2), ZZ12 = quad_bcast<2>(r2), ZZZ12 = quad_bcast<3>(r2);
    Fq<2> sPP = fq_sqr(sPd), sRR = fq_sqr(sRd);
    if (fq_eq_modp(PP, sPP) && fq_eq_modp(RR, sRR) && fq_eq_modp(ZZ12, fq_mul(acc.zz, q.zz)) && fq_eq_modp(ZZZ12, fq_mul(acc.zzz, q.zzz))) ok |= 4;
    Fq<4> a3 = quad_sel(ql, Pd, fq_widen<4>(U1), fq_widen<4>(ZZ12), fq_widen<4>(ZZZ12));
    Fq<2> r3 = fq_mul(a3, PP);
    Fq<2> PPP = quad_bcast<0>(r3), Qv = quad_bcast<1>(r3), ZZ3 = quad_bcast<2>(r3), Wv = quad_bcast<3>(r3);
    Fq<2> sPPP = fq_mul(sPd, sPP), sQ = fq_mul(sU1, sPP);
    if (fq_eq_modp(PPP, sPPP) && fq_eq_modp(Qv, sQ)) ok |= 8;
    Fq<8> x3 = fq_sub_sub2(RR, PPP, Qv);
    Fq<8> sx3 = fq_sub_sub2(sRR, sPPP, sQ);
    if (fq_eq_modp(x3, sx3)) ok |= 16;
    Fq<4> a4 = quad_sel(ql, Rd, fq_widen<4>(S1), fq_widen<4>(Wv), fq_widen<4>(Wv));
    Fq<10> b4 = quad_sel(ql, fq_sub<8>(Qv, x3), fq_widen<10>(PPP), fq_widen<10>(Pd), fq_widen<10>(Pd));
    Fq<2> r4 = fq_mul(a4, b4);
    Fq<2> A = quad_bcast<0>(r4), Bm = quad_bcast<1>(r4);
    if (fq_eq_modp(A, fq_mul(sRd, fq_sub<8>(sQ, sx3))) && fq_eq_modp(Bm, fq_mul(sS1, sPPP))) ok |= 32;
    return ok;
}
__global__ void k(unsigned *out) {
    unsigned t = threadIdx.x + blockIdx.x * blockDim.x, i = t >> 2; int ql = t & 3;
    // point i: (i + 2) * G by repeated addition of G = (-1, 2) in native form
    AffN g; g.x = fq_neg<2>(fq_one()); g.y = fq_widen<2>(fq_tighten(fq_muls<2>(fq_one())));
    XyzzN p = xyzz_from_aff(g), q = xyzz_from_aff(g);
    for (unsigned k = 0; k < i + 1; k++) xyzz_madd(p, g);
    q = xyzz_dbl(p); xyzz_madd(q, g);  // q = (2i + 5) G
    XyzzN d1 = xyzz_dbl(p), d2 = xyzz_dbl_quad(p, ql);
    XyzzN s1 = p; xyzz_add(s1, q); XyzzN s2 = p; xyzz_add_quad(s2, q, ql);
    XyzzN e2 = p; xyzz_add_quad(e2, p, ql);
    unsigned r = (same(d1, d2) ? 1 : 0) | (same(s1, s2) ? 2 : 0) | (same(d1, e2) ? 4 : 0);
    unsigned c = (fq_eq_modp(s1.x, s2.x) ? 1 : 0) | (fq_eq_modp(s1.y, s2.y) ? 2 : 0) | (fq_eq_modp(s1.zz, s2.zz) ? 4 : 0) | (fq_eq_modp(s1.zzz, s2.zzz) ? 8 : 0);
    out[t] = r | (dbg_add(p, q, ql) << 8) | (c << 16);
}
int main() {
    unsigned *d, h[256];
    (void)hipMalloc(&d, 1024);
    k<<<1, 256>>>(d);
    (void)hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    for (int i = 0; i < 16; i++) printf("%x ", h[i]); printf("\n");
}
