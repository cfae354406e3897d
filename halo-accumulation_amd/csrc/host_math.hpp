// Host-side Pallas arithmetic for the glue the reference keeps on the CPU between kernels:
// Fiat-Shamir hashing (group.rs:41-89), challenge inversion (pcdl.rs:213), the O(lg n)
// succinct check (pcdl.rs:252-314), the final window combine of an MSM and the
// normalisation of points that leave the library.  4 x 64-bit limb Montgomery (R = 2^256),
// the same in-memory form arkworks uses, so limbs cross the C ABI untouched.
//
// This is product code (linked into libhalo_hip.so); it shares nothing with oracle/.
#pragma once
#if defined(__x86_64__)
#include <x86intrin.h>
#endif
#include <array>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace halo {
namespace host {
inline bool g_inv_fermat = false;  // set from tuning() when a context is created (development switch)

using u64 = uint64_t;
using u128 = unsigned __int128;

// add / subtract with carry, and the spin-wait hint: the x86 intrinsics where the host is one (adc / sbb chains without a
// branch), plain 128-bit arithmetic anywhere else -- the library builds on any 64-bit host an MI355X can sit in
#if defined(__x86_64__)
inline unsigned char addc64(unsigned char c, u64 a, u64 b, unsigned long long *out) { return _addcarry_u64(c, a, b, out); }
inline unsigned char subb64(unsigned char b, u64 a, u64 c, unsigned long long *out) { return _subborrow_u64(b, a, c, out); }
inline void cpu_relax() { __builtin_ia32_pause(); }
#else
inline unsigned char addc64(unsigned char c, u64 a, u64 b, unsigned long long *out) { u128 s = (u128)a + b + c; *out = (u64)s; return (unsigned char)(s >> 64); }
inline unsigned char subb64(unsigned char b, u64 a, u64 c, unsigned long long *out) { u128 d = (u128)a - c - b; *out = (u64)d; return (unsigned char)((d >> 64) & 1); }
inline void cpu_relax() {
#if defined(__aarch64__)
    __asm__ __volatile__("yield");
#endif
}
#endif

struct FqP {
    static constexpr u64 M[4] = {0x992d30ed00000001ULL, 0x224698fc094cf91bULL, 0x0ULL, 0x4000000000000000ULL};
    static constexpr u64 ONE[4] = {0x34786d38fffffffdULL, 0x992c350be41914adULL, 0xffffffffffffffffULL, 0x3fffffffffffffffULL};
    static constexpr u64 R2[4] = {0x8c78ecb30000000fULL, 0xd7d30dbd8b0de0e7ULL, 0x7797a99bc3c95d18ULL, 0x096d41af7b9cb714ULL};
    static constexpr u64 INV = 0x992d30ecffffffffULL;
};
struct FrP {
    static constexpr u64 M[4] = {0x8c46eb2100000001ULL, 0x224698fc0994a8ddULL, 0x0ULL, 0x4000000000000000ULL};
    static constexpr u64 ONE[4] = {0x5b2b3e9cfffffffdULL, 0x992c350be3420567ULL, 0xffffffffffffffffULL, 0x3fffffffffffffffULL};
    static constexpr u64 R2[4] = {0xfc9678ff0000000fULL, 0x67bb433d891a16e3ULL, 0x7fae231004ccf590ULL, 0x096d41af7ccfdaa9ULL};
    static constexpr u64 INV = 0x8c46eb20ffffffffULL;
};

// ---- modular inverse by 62 division steps at a time (Bernstein-Yang "safegcd", in the variable-time form published with
// libsecp256k1's modinv64: transition matrices from the low 64 bits of f and g, applied to f, g and to the cofactors d, e in
// signed 62-bit limbs).  Every value inverted on the host is public (proof points, challenges), so variable time is fine.
// ~1.4 us for a 255-bit value against ~9.5 us for the Fermat power: an open has two of them on its critical path per round
// (the normalisation of L / R for the transcript, then xi^-1).  tests/native/host_math_sanitize.cpp compares the two.
namespace modinv {
typedef __int128 i128;
struct S62 { int64_t v[5]; };  // sum v[i] 2^(62 i); limbs 0..3 in [0, 2^62), limb 4 signed
struct T22 { int64_t u, v, q, r; };  // 2^62 x the transition matrix of 62 division steps
static constexpr int64_t M62 = (int64_t)((~(uint64_t)0) >> 2);
inline int64_t divsteps_62_var(int64_t eta, uint64_t f0, uint64_t g0, T22 *t) {
    uint64_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0, m;
    uint32_t w;
    int i = 62, limit, zeros;
    for (;;) {
        zeros = __builtin_ctzll(g | (~(uint64_t)0 << i));  // (the sentinel bit stops the count at i)
        g >>= zeros; u <<= zeros; v <<= zeros; eta -= zeros; i -= zeros;
        if (i == 0) break;
        if (eta < 0) {  // swap: (f, g) <- (g, -f), and cancel up to 6 low bits of g at once
            uint64_t tmp;
            eta = -eta;
            tmp = f; f = g; g = (uint64_t)0 - tmp;
            tmp = u; u = q; q = (uint64_t)0 - tmp;
            tmp = v; v = r; r = (uint64_t)0 - tmp;
            limit = ((int)eta + 1) > i ? i : ((int)eta + 1);
            m = (~(uint64_t)0 >> (64 - limit)) & 63u;
            w = (uint32_t)((f * g * (f * f - 2)) & m);
        } else {        // up to 4 bits
            limit = ((int)eta + 1) > i ? i : ((int)eta + 1);
            m = (~(uint64_t)0 >> (64 - limit)) & 15u;
            w = (uint32_t)(f + (((f + 1) & 4) << 1));
            w = (uint32_t)((((uint64_t)0 - (uint64_t)w) * g) & m);
        }
        g += f * w; q += u * w; r += v * w;
    }
    t->u = (int64_t)u; t->v = (int64_t)v; t->q = (int64_t)q; t->r = (int64_t)r;
    return eta;
}
// (d, e) <- t (d, e) / 2^62 mod M, both kept in (-2M, M)
inline void update_de_62(S62 *d, S62 *e, const T22 *t, const S62 &mod, uint64_t modinv62) {
    const int64_t d0 = d->v[0], d1 = d->v[1], d2 = d->v[2], d3 = d->v[3], d4 = d->v[4];
    const int64_t e0 = e->v[0], e1 = e->v[1], e2 = e->v[2], e3 = e->v[3], e4 = e->v[4];
    const int64_t u = t->u, v = t->v, q = t->q, r = t->r;
    const int64_t sd = d4 >> 63, se = e4 >> 63;
    int64_t md = (u & sd) + (v & se), me = (q & sd) + (r & se);
    i128 cd = (i128)u * d0 + (i128)v * e0, ce = (i128)q * d0 + (i128)r * e0;
    md -= (int64_t)((modinv62 * (uint64_t)cd + (uint64_t)md) & (uint64_t)M62);  // so that the low 62 bits below vanish
    me -= (int64_t)((modinv62 * (uint64_t)ce + (uint64_t)me) & (uint64_t)M62);
    cd += (i128)mod.v[0] * md; ce += (i128)mod.v[0] * me;
    cd >>= 62; ce >>= 62;
    cd += (i128)u * d1 + (i128)v * e1 + (i128)mod.v[1] * md; ce += (i128)q * d1 + (i128)r * e1 + (i128)mod.v[1] * me;
    d->v[0] = (int64_t)cd & M62; cd >>= 62; e->v[0] = (int64_t)ce & M62; ce >>= 62;
    cd += (i128)u * d2 + (i128)v * e2 + (i128)mod.v[2] * md; ce += (i128)q * d2 + (i128)r * e2 + (i128)mod.v[2] * me;
    d->v[1] = (int64_t)cd & M62; cd >>= 62; e->v[1] = (int64_t)ce & M62; ce >>= 62;
    cd += (i128)u * d3 + (i128)v * e3 + (i128)mod.v[3] * md; ce += (i128)q * d3 + (i128)r * e3 + (i128)mod.v[3] * me;
    d->v[2] = (int64_t)cd & M62; cd >>= 62; e->v[2] = (int64_t)ce & M62; ce >>= 62;
    cd += (i128)u * d4 + (i128)v * e4 + (i128)mod.v[4] * md; ce += (i128)q * d4 + (i128)r * e4 + (i128)mod.v[4] * me;
    d->v[3] = (int64_t)cd & M62; cd >>= 62; e->v[3] = (int64_t)ce & M62; ce >>= 62;
    d->v[4] = (int64_t)cd; e->v[4] = (int64_t)ce;
}
// (f, g) <- t (f, g) / 2^62 over the `len` limbs still in use
inline void update_fg_62_var(int len, S62 *f, S62 *g, const T22 *t) {
    const int64_t u = t->u, v = t->v, q = t->q, r = t->r;
    int64_t fi = f->v[0], gi = g->v[0];
    i128 cf = (i128)u * fi + (i128)v * gi, cg = (i128)q * fi + (i128)r * gi;
    cf >>= 62; cg >>= 62;
    for (int i = 1; i < len; ++i) {
        fi = f->v[i]; gi = g->v[i];
        cf += (i128)u * fi + (i128)v * gi; cg += (i128)q * fi + (i128)r * gi;
        f->v[i - 1] = (int64_t)cf & M62; cf >>= 62;
        g->v[i - 1] = (int64_t)cg & M62; cg >>= 62;
    }
    f->v[len - 1] = (int64_t)cf; g->v[len - 1] = (int64_t)cg;
}
inline S62 to_s62(const uint64_t x[4]) {
    S62 r;
    r.v[0] = (int64_t)(x[0] & (uint64_t)M62); r.v[1] = (int64_t)(((x[0] >> 62) | (x[1] << 2)) & (uint64_t)M62);
    r.v[2] = (int64_t)(((x[1] >> 60) | (x[2] << 4)) & (uint64_t)M62); r.v[3] = (int64_t)(((x[2] >> 58) | (x[3] << 6)) & (uint64_t)M62);
    r.v[4] = (int64_t)(x[3] >> 56);
    return r;
}
// x^-1 mod M as plain integers (x in [1, M), M odd and below 2^255); out in [0, M)
inline void inverse(const uint64_t x[4], const uint64_t M[4], uint64_t out[4]) {
    const S62 mod = to_s62(M);
    uint64_t minv = 1;
    for (int i = 0; i < 6; i++) minv *= 2 - M[0] * minv;  // M^-1 mod 2^64 (Newton)
    const uint64_t modinv62 = minv & (uint64_t)M62;
    S62 d = {{0, 0, 0, 0, 0}}, e = {{1, 0, 0, 0, 0}}, f = mod, g = to_s62(x);
    int len = 5;
    int64_t eta = -1;  // eta = -delta, delta starts at 1
    for (;;) {
        T22 t;
        eta = divsteps_62_var(eta, (uint64_t)f.v[0], (uint64_t)g.v[0], &t);
        update_de_62(&d, &e, &t, mod, modinv62);
        update_fg_62_var(len, &f, &g, &t);
        if (g.v[0] == 0) {
            int64_t any = 0;
            for (int j = 1; j < len; ++j) any |= g.v[j];
            if (any == 0) break;  // g = 0: f = +-1 and d = +-x^-1
        }
        const int64_t fn = f.v[len - 1], gn = g.v[len - 1];
        int64_t cond = ((int64_t)len - 2) >> 63;
        cond |= fn ^ (fn >> 63);
        cond |= gn ^ (gn >> 63);
        if (cond == 0) {  // the top limbs of f and g are 0 or -1: fold their sign into the limb below
            f.v[len - 2] |= (int64_t)((uint64_t)fn << 62);
            g.v[len - 2] |= (int64_t)((uint64_t)gn << 62);
            --len;
        }
    }
    // d in (-2M, M), to be negated if f = -1: into [0, M)
    int64_t r[5] = {d.v[0], d.v[1], d.v[2], d.v[3], d.v[4]};
    auto add_mod_if = [&](int64_t mask) { for (int i = 0; i < 5; ++i) r[i] += mod.v[i] & mask; };
    auto carry = [&]() { for (int i = 0; i < 4; ++i) { r[i + 1] += r[i] >> 62; r[i] &= M62; } };
    add_mod_if(r[4] >> 63);
    const int64_t neg = f.v[len - 1] >> 63;
    for (int i = 0; i < 5; ++i) r[i] = (r[i] ^ neg) - neg;
    carry();
    add_mod_if(r[4] >> 63);
    carry();
    out[0] = (uint64_t)r[0] | ((uint64_t)r[1] << 62); out[1] = ((uint64_t)r[1] >> 2) | ((uint64_t)r[2] << 60);
    out[2] = ((uint64_t)r[2] >> 4) | ((uint64_t)r[3] << 58); out[3] = ((uint64_t)r[3] >> 6) | ((uint64_t)r[4] << 56);
}
}  // namespace modinv

template <class P>
struct Fp {
    u64 l[4];

    static Fp zero() { return Fp{{0, 0, 0, 0}}; }
    static Fp one() { return Fp{{P::ONE[0], P::ONE[1], P::ONE[2], P::ONE[3]}}; }
    static Fp load(const u64 *p) { Fp r; std::memcpy(r.l, p, 32); return r; }
    void store(u64 *p) const { std::memcpy(p, l, 32); }
    bool is_zero() const { return (l[0] | l[1] | l[2] | l[3]) == 0; }
    bool operator==(const Fp &o) const { return std::memcmp(l, o.l, 32) == 0; }
    bool operator!=(const Fp &o) const { return !(*this == o); }

    static bool geq(const u64 a[4], const u64 b[4]) {
        for (int i = 3; i >= 0; --i)
            if (a[i] != b[i]) return a[i] > b[i];
        return true;
    }
    static u64 sub_limbs(u64 r[4], const u64 a[4], const u64 b[4]) {
        u64 borrow = 0;
        for (int i = 0; i < 4; ++i) {
            u128 d = (u128)a[i] - b[i] - borrow;
            r[i] = (u64)d;
            borrow = (u64)(d >> 64) & 1;
        }
        return borrow;
    }
    static u64 add_limbs(u64 r[4], const u64 a[4], const u64 b[4]) {
        u64 carry = 0;
        for (int i = 0; i < 4; ++i) {
            u128 s = (u128)a[i] + b[i] + carry;
            r[i] = (u64)s;
            carry = (u64)(s >> 64);
        }
        return carry;
    }
    // Branch-free (the window combine of an MSM is ~250 point doublings in a row on the host, between two rounds of an open, and
    // half of a doubling's time went into the compare-and-branch form of these two): a + b < 2M < 2^256 never carries out, and
    // whether M has to come off again is the borrow of the trial subtraction.
    Fp operator+(const Fp &o) const {
        unsigned long long s0, s1, s2, s3, d0, d1, d2, d3;
        unsigned char c = addc64(0, l[0], o.l[0], &s0);
        c = addc64(c, l[1], o.l[1], &s1);
        c = addc64(c, l[2], o.l[2], &s2);
        (void)addc64(c, l[3], o.l[3], &s3);
        unsigned char b = subb64(0, s0, P::M[0], &d0);
        b = subb64(b, s1, P::M[1], &d1);
        b = subb64(b, s2, P::M[2], &d2);
        b = subb64(b, s3, P::M[3], &d3);
        const u64 keep = (u64)0 - (u64)b;  // all ones: the sum was below M
        return Fp{{(s0 & keep) | (d0 & ~keep), (s1 & keep) | (d1 & ~keep), (s2 & keep) | (d2 & ~keep), (s3 & keep) | (d3 & ~keep)}};
    }
    Fp operator-(const Fp &o) const {
        unsigned long long d0, d1, d2, d3, r0, r1, r2, r3;
        unsigned char b = subb64(0, l[0], o.l[0], &d0);
        b = subb64(b, l[1], o.l[1], &d1);
        b = subb64(b, l[2], o.l[2], &d2);
        b = subb64(b, l[3], o.l[3], &d3);
        const u64 back = (u64)0 - (u64)b;  // all ones: the difference went below zero, M goes back on
        unsigned char c = addc64(0, d0, P::M[0] & back, &r0);
        c = addc64(c, d1, P::M[1] & back, &r1);
        c = addc64(c, d2, P::M[2] & back, &r2);
        (void)addc64(c, d3, P::M[3] & back, &r3);
        return Fp{{r0, r1, r2, r3}};
    }
    Fp operator-() const {
        if (is_zero()) return *this;
        Fp r;
        sub_limbs(r.l, P::M, l);
        return r;
    }
    // Montgomery product, coarsely integrated operand scanning (CIOS), fully unrolled: t stays below 2M
    // after every round, so four limbs plus a carry word hold it
    Fp operator*(const Fp &o) const {
        u64 t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
        for (int i = 0; i < 4; ++i) {
            const u64 a = l[i];
            u128 c = (u128)a * o.l[0] + t0;
            t0 = (u64)c;
            c = (c >> 64) + (u128)a * o.l[1] + t1;
            t1 = (u64)c;
            c = (c >> 64) + (u128)a * o.l[2] + t2;
            t2 = (u64)c;
            c = (c >> 64) + (u128)a * o.l[3] + t3;
            t3 = (u64)c;
            c = (c >> 64) + t4;
            t4 = (u64)c;
            const u64 t5 = (u64)(c >> 64);
            const u64 m = t0 * P::INV;  // t + m*M = 0 mod 2^64
            c = (u128)m * P::M[0] + t0;
            c = (c >> 64) + (u128)m * P::M[1] + t1;
            t0 = (u64)c;
            c = (c >> 64) + (u128)m * P::M[2] + t2;
            t1 = (u64)c;
            c = (c >> 64) + (u128)m * P::M[3] + t3;
            t2 = (u64)c;
            c = (c >> 64) + t4;
            t3 = (u64)c;
            t4 = (u64)(c >> 64) + t5;
        }
        // t < 2M: M comes off once if t >= M -- t4 set, or no borrow from the trial subtraction (branch-free, like operator+)
        unsigned long long d0, d1, d2, d3;
        unsigned char b = subb64(0, t0, P::M[0], &d0);
        b = subb64(b, t1, P::M[1], &d1);
        b = subb64(b, t2, P::M[2], &d2);
        b = subb64(b, t3, P::M[3], &d3);
        const u64 keep = ((u64)0 - (u64)b) & ((u64)0 - (u64)(t4 == 0));  // all ones: t was below M
        return Fp{{(t0 & keep) | (d0 & ~keep), (t1 & keep) | (d1 & ~keep), (t2 & keep) | (d2 & ~keep), (t3 & keep) | (d3 & ~keep)}};
    }
    Fp sqr() const { return *this * *this; }
    Fp dbl() const { return *this + *this; }
    Fp from_mont() const { return *this * Fp{{1, 0, 0, 0}}; }
    Fp to_mont() const { return *this * Fp{{P::R2[0], P::R2[1], P::R2[2], P::R2[3]}}; }
    static Fp from_u64(u64 v) { return Fp{{v, 0, 0, 0}}.to_mont(); }
    // reduce a 256-bit little-endian integer mod M (ark-ff from_le_bytes_mod_order on 32 bytes)
    static Fp from_le_bytes_mod_order(const uint8_t b[32]) {
        Fp v;
        std::memcpy(v.l, b, 32);
        while (geq(v.l, P::M)) sub_limbs(v.l, v.l, P::M);  // 2^256 < 4M
        return v.to_mont();
    }
    Fp pow(const u64 e[4]) const {
        Fp acc = one();
        for (int i = 255; i >= 0; --i) {
            acc = acc.sqr();
            if ((e[i / 64] >> (i % 64)) & 1) acc = acc * *this;
        }
        return acc;
    }
    // Fermat inverse (the slow, obviously right one: kept as the cross-check of inv())
    Fp inv_fermat() const {
        u64 e[4], two[4] = {2, 0, 0, 0};
        sub_limbs(e, P::M, two);
        return pow(e);
    }
    // inverse; caller checks for zero (ark-ff inverse() -> None; zero in gives zero out, as the Fermat power does).
    // The limbs hold a = x R: the plain integer inverse of a is x^-1 R^-1, and one Montgomery product with R^3 makes it x^-1 R.
    Fp inv() const {
        if (is_zero()) return *this;
        if (g_inv_fermat) return inv_fermat();  // development switch for A/B runs (HALO_HOST_INV_FERMAT, csrc/tuning.hpp)
        static const Fp R3 = Fp{{P::R2[0], P::R2[1], P::R2[2], P::R2[3]}} * Fp{{P::R2[0], P::R2[1], P::R2[2], P::R2[3]}};
        Fp r;
        modinv::inverse(l, P::M, r.l);
        return r * R3;
    }
};

using Fq = Fp<FqP>;
using Fr = Fp<FrP>;

// ------------------------------------------------------------------ points
struct Affine {
    Fq x, y;
    bool inf;
};
struct Point {  // Jacobian, Z = 0 infinity
    Fq X, Y, Z;

    static Point infinity() { return Point{Fq::one(), Fq::one(), Fq::zero()}; }
    static Point from_affine(const Fq &x, const Fq &y) { return Point{x, y, Fq::one()}; }
    static Point load(const u64 *w) { return Point{Fq::load(w), Fq::load(w + 4), Fq::load(w + 8)}; }
    static Point load_affine(const u64 *w) {
        Fq x = Fq::load(w), y = Fq::load(w + 4);
        if (x.is_zero() && y.is_zero()) return infinity();
        return from_affine(x, y);
    }
    static Point generator() { return from_affine(-Fq::one(), Fq::one().dbl()); }  // (-1, 2)
    bool is_inf() const { return Z.is_zero(); }
    // Y^2 = X^3 + 5 Z^6 with coordinates below the modulus (what ark-ec's checked deserialisation guarantees for
    // every `Projective` the reference ever holds); infinity is on the curve
    bool on_curve() const {
        if (Fq::geq(X.l, FqP::M) || Fq::geq(Y.l, FqP::M) || Fq::geq(Z.l, FqP::M)) return false;
        if (is_inf()) return true;
        Fq z2 = Z.sqr(), z6 = z2.sqr() * z2;
        Fq five = Fq::from_u64(5);
        return Y.sqr() == X.sqr() * X + five * z6;
    }

    Point dbl() const {  // dbl-2009-l
        if (is_inf()) return *this;
        Fq A = X.sqr(), B = Y.sqr(), C = B.sqr();
        Fq D = ((X + B).sqr() - A - C).dbl();
        Fq E = A.dbl() + A, F = E.sqr();
        Point r;
        r.X = F - D.dbl();
        r.Y = E * (D - r.X) - C.dbl().dbl().dbl();
        r.Z = (Y * Z).dbl();
        return r;
    }
    Point operator+(const Point &q) const {  // add-2007-bl
        if (is_inf()) return q;
        if (q.is_inf()) return *this;
        Fq Z1Z1 = Z.sqr(), Z2Z2 = q.Z.sqr();
        Fq U1 = X * Z2Z2, U2 = q.X * Z1Z1;
        Fq S1 = Y * q.Z * Z2Z2, S2 = q.Y * Z * Z1Z1;
        if (U1 == U2) return S1 == S2 ? dbl() : infinity();
        Fq H = U2 - U1, I = H.dbl().sqr(), J = H * I, rr = (S2 - S1).dbl(), V = U1 * I;
        Point r;
        r.X = rr.sqr() - J - V.dbl();
        r.Y = rr * (V - r.X) - (S1 * J).dbl();
        r.Z = ((Z + q.Z).sqr() - Z1Z1 - Z2Z2) * H;
        return r;
    }
    Point operator-() const { return Point{X, -Y, Z}; }
    Point operator-(const Point &q) const { return *this + (-q); }
    bool operator==(const Point &q) const {  // projective equality, as ark-ec
        if (is_inf() || q.is_inf()) return is_inf() && q.is_inf();
        Fq Z1Z1 = Z.sqr(), Z2Z2 = q.Z.sqr();
        if (X * Z2Z2 != q.X * Z1Z1) return false;
        return Y * Z2Z2 * q.Z == q.Y * Z1Z1 * Z;
    }
    bool operator!=(const Point &q) const { return !(*this == q); }
    // scalar multiple by a Montgomery-form Fr, 4-bit fixed windows
    Point mul(const Fr &k_mont) const {
        Fr k = k_mont.from_mont();
        Point tbl[16];
        tbl[0] = infinity();
        tbl[1] = *this;
        for (int i = 2; i < 16; ++i) tbl[i] = (i & 1) ? tbl[i - 1] + *this : tbl[i / 2].dbl();
        Point acc = infinity();
        for (int w = 63; w >= 0; --w) {
            if (!acc.is_inf()) acc = acc.dbl().dbl().dbl().dbl();
            unsigned nib = (unsigned)(k.l[w / 16] >> (4 * (w % 16))) & 15u;
            if (nib) acc = acc + tbl[nib];
        }
        return acc;
    }
    Affine to_affine() const {
        if (is_inf()) return Affine{Fq::zero(), Fq::zero(), true};
        if (Z == Fq::one()) return Affine{X, Y, false};  // already normalised (everything the library emits)
        Fq zi = Z.inv(), zi2 = zi.sqr();
        return Affine{X * zi2, Y * zi2 * zi, false};
    }
    Point normalized() const {
        if (is_inf()) return infinity();
        Affine a = to_affine();
        return from_affine(a.x, a.y);
    }
    void store(u64 *w) const { X.store(w); Y.store(w + 4); Z.store(w + 8); }
    void store_normalized(u64 *w) const { normalized().store(w); }
};

// k * P for a point used many times (H' of one pcdl::open): 64 windows x 15 multiples, no doublings
class FixedBaseTable {
   public:
    FixedBaseTable() = default;
    explicit FixedBaseTable(const Point &p) : base_(p), tbl_(64 * 16) {
        Point w = p;
        for (int i = 0; i < 64; ++i) {
            tbl_[16 * i] = Point::infinity();
            for (int d = 1; d < 16; ++d) tbl_[16 * i + d] = tbl_[16 * i + d - 1] + w;
            w = tbl_[16 * i + 15] + w;  // 16^(i+1) * P
        }
    }
    bool matches(const Point &p) const { return !tbl_.empty() && base_.X == p.X && base_.Y == p.Y && base_.Z == p.Z; }
    Point mul(const Fr &k_mont) const {
        Fr k = k_mont.from_mont();
        Point acc = Point::infinity();
        for (int i = 0; i < 64; ++i) {
            unsigned nib = (unsigned)(k.l[i / 16] >> (4 * (i % 16))) & 15u;
            if (nib) acc = acc + tbl_[16 * i + nib];
        }
        return acc;
    }

   private:
    Point base_;
    std::vector<Point> tbl_;
};

// sum_i k_i * P_i for a handful of points (succinct check, acc.rs:178): interleaved 4-bit windows
inline Point small_msm(const std::vector<Point> &pts, const std::vector<Fr> &ks_mont) {
    size_t n = pts.size() < ks_mont.size() ? pts.size() : ks_mont.size();
    std::vector<std::array<Point, 16>> tbl(n);
    std::vector<Fr> ks(n);
    for (size_t i = 0; i < n; ++i) {
        ks[i] = ks_mont[i].from_mont();
        tbl[i][0] = Point::infinity();
        tbl[i][1] = pts[i];
        for (int j = 2; j < 16; ++j) tbl[i][j] = (j & 1) ? tbl[i][j - 1] + pts[i] : tbl[i][j / 2].dbl();
    }
    Point acc = Point::infinity();
    for (int w = 63; w >= 0; --w) {
        if (!acc.is_inf()) acc = acc.dbl().dbl().dbl().dbl();
        for (size_t i = 0; i < n; ++i) {
            unsigned nib = (unsigned)(ks[i].l[w / 16] >> (4 * (w % 16))) & 15u;
            if (nib) acc = acc + tbl[i][nib];
        }
    }
    return acc;
}

// ------------------------------------------------------------------ GLV digits for the uniform-scalar fold
// Pallas has the endomorphism phi(x, y) = (beta x, y) = [lambda](x, y), beta^3 = 1 in Fq,
// lambda^3 = 1 in Fr.  xi = k1 + k2 lambda with |k1|, |k2| < 2^128 (Babai rounding on the short
// basis (a1, -b1n), (a2, b2)), i.e. xi is the Eisenstein integer z = k1 + k2 w (w^2 + w + 1 = 0).
// z is then written in base 2 with digits from {0, +-1, +-w, +-w^2}: z = sum_i d_i 2^i.  Every
// non-zero digit is a unit, and unit * P = (beta^e x, +-y) costs nothing, so xi P is a ~129-step
// double-and-add whose additions all take a "free" table entry.  z mod 2 fixes the unit up to its
// sign; the sign is chosen so that the next digit is zero whenever that is possible, which leaves
// ~0.6 additions per doubling (0.75 for the plain joint binary ladder over k1, k2).
struct GlvDigits {
    uint8_t d[144];  // digit codes, least significant first: 0 none, 1..3 = +w^0..+w^2, 4..6 = -w^0..-w^2
    int n;           // number of digits (<= 132)
};
namespace glv_detail {
inline void mul_limbs(const u64 *a, int na, const u64 *b, int nb, u64 *out /* na + nb */) {
    for (int i = 0; i < na + nb; ++i) out[i] = 0;
    for (int i = 0; i < na; ++i) {
        u64 carry = 0;
        for (int j = 0; j < nb; ++j) {
            u128 t = (u128)a[i] * b[j] + out[i + j] + carry;
            out[i + j] = (u64)t;
            carry = (u64)(t >> 64);
        }
        out[i + nb] = carry;
    }
}
// (k * g + 2^383) >> 384 for 4-limb k and 5-limb g -> 3 limbs
inline void round_shift(const u64 k[4], const u64 g[5], u64 c[3]) {
    u64 prod[9];
    mul_limbs(k, 4, g, 5, prod);
    u64 carry = 1ULL << 63;  // + 2^383: bit 63 of limb 5
    for (int i = 5; i < 9 && carry; ++i) {
        u64 t = prod[i] + carry;
        carry = t < prod[i] ? 1 : 0;
        prod[i] = t;
    }
    c[0] = prod[6]; c[1] = prod[7]; c[2] = prod[8];
}
inline void sub256(u64 r[4], const u64 a[4], const u64 b[4]) {
    u64 borrow = 0;
    for (int i = 0; i < 4; ++i) { u128 d = (u128)a[i] - b[i] - borrow; r[i] = (u64)d; borrow = (u64)(d >> 64) & 1; }
}
inline void mul_lo256(const u64 c[3], const u64 v[2], u64 out[4]) {
    u64 full[5];
    mul_limbs(c, 3, v, 2, full);
    for (int i = 0; i < 4; ++i) out[i] = full[i];
}
// 256-bit two's complement helpers
inline bool is_zero256(const u64 v[4]) { return (v[0] | v[1] | v[2] | v[3]) == 0; }
inline void add_small(u64 v[4], int s) {  // v += s, s in {-1, 0, 1}
    if (s > 0) { for (int i = 0; i < 4 && ++v[i] == 0; ++i) {} }
    if (s < 0) { for (int i = 0; i < 4 && v[i]-- == 0; ++i) {} }
}
inline void sar1(u64 v[4]) {  // arithmetic shift right by one
    u64 sign = v[3] & (1ULL << 63);
    for (int i = 0; i < 3; ++i) v[i] = (v[i] >> 1) | (v[i + 1] << 63);
    v[3] = (v[3] >> 1) | sign;
}
}  // namespace glv_detail

// k1, k2 (256-bit two's complement, |.| < 2^128) with xi = k1 + k2 lambda (mod r)
inline void glv_decompose(const Fr &xi_mont, u64 k1[4], u64 k2[4]) {
    using namespace glv_detail;
    static const u64 G1[5] = {0x111f686111afc293ULL, 0xc35fbd4d086862e0ULL, 0x31f0256800000002ULL, 0x4f34e8b2066389a4ULL, 0x2ULL};
    static const u64 G2[5] = {0x4a95a2d972171db4ULL, 0x61afdea68480fa55ULL, 0x32c49e4bffffffffULL, 0x279a745902a2654eULL, 0x1ULL};
    static const u64 A1[2] = {0x7fcae1c700000001ULL, 0x49e69d1640f04915ULL};
    static const u64 B1N[2] = {0x8cb1279300000000ULL, 0x49e69d1640a89953ULL};  // b1 = -B1N
    static const u64 A2[2] = {0x8cb1279300000000ULL, 0x49e69d1640a89953ULL};
    static const u64 B2[2] = {0x0c7c095a00000001ULL, 0x93cd3a2c8198e269ULL};
    Fr k = xi_mont.from_mont();
    u64 c1[3], c2[3], t[4];
    round_shift(k.l, G1, c1);
    round_shift(k.l, G2, c2);
    // k1 = k - c1 a1 - c2 a2 ; k2 = c1 b1n - c2 b2   (mod 2^256, small signed results)
    mul_lo256(c1, A1, t); sub256(k1, k.l, t);
    mul_lo256(c2, A2, t); sub256(k1, k1, t);
    mul_lo256(c1, B1N, k2);
    mul_lo256(c2, B2, t); sub256(k2, k2, t);
}

// The sign choices form a tiny search space: whatever signs were taken, the value left after i digits is
// floor-ish(z / 2^i) plus a bounded offset, so at most a handful (measured: 3) of distinct remainders exist
// per level.  A level-by-level dynamic program over them finds the expansion with the fewest non-zero
// digits (~71 of ~127 for a random xi; the greedy "make the next digit zero" rule gives ~77).
inline GlvDigits glv_digits(const Fr &xi_mont) {
    using namespace glv_detail;
    // coordinates (a, b) of the units +1, +w, +w^2 = -1 - w over the basis (1, w); codes 4..6 are their negatives
    static const int UA[3] = {1, 0, -1}, UB[3] = {0, 1, -1};
    constexpr int LEVELS = 140, WIDTH = 8;
    struct State { u64 a[4], b[4]; int cost, parent; uint8_t code; };
    static thread_local State lv[LEVELS + 1][WIDTH];
    int count[LEVELS + 1];
    glv_decompose(xi_mont, lv[0][0].a, lv[0][0].b);
    lv[0][0].cost = 0; lv[0][0].parent = -1; lv[0][0].code = 0;
    count[0] = 1;
    int best_level = -1, best_idx = -1, best_cost = 1 << 30;
    for (int L = 0; L < LEVELS; ++L) {
        count[L + 1] = 0;
        auto push = [&](const u64 a[4], const u64 b[4], int cost, int parent, uint8_t code) {
            for (int k = 0; k < count[L + 1]; ++k) {
                State &t = lv[L + 1][k];
                if (std::memcmp(t.a, a, 32) == 0 && std::memcmp(t.b, b, 32) == 0) {
                    if (cost < t.cost) { t.cost = cost; t.parent = parent; t.code = code; }
                    return;
                }
            }
            if (count[L + 1] == WIDTH) return;  // cannot happen (<= 3 remainders per level); keeps the table bounded
            State &t = lv[L + 1][count[L + 1]++];
            std::memcpy(t.a, a, 32); std::memcpy(t.b, b, 32);
            t.cost = cost; t.parent = parent; t.code = code;
        };
        bool live = false;
        for (int k = 0; k < count[L]; ++k) {
            const State &st = lv[L][k];
            if (is_zero256(st.a) && is_zero256(st.b)) {
                if (st.cost < best_cost) { best_cost = st.cost; best_level = L; best_idx = k; }
                continue;
            }
            if (st.cost >= best_cost) continue;  // cannot beat a finished expansion any more
            live = true;
            unsigned pa = (unsigned)(st.a[0] & 1), pb = (unsigned)(st.b[0] & 1);
            u64 a[4], b[4];
            if (!(pa | pb)) {
                std::memcpy(a, st.a, 32); std::memcpy(b, st.b, 32);
                sar1(a); sar1(b);
                push(a, b, st.cost, k, 0);
            } else {
                int e = (pa && pb) ? 2 : (pb ? 1 : 0);  // z = 1, w, 1 + w = -w^2 (mod 2): the unit up to its sign
                for (int neg = 0; neg < 2; ++neg) {
                    std::memcpy(a, st.a, 32); std::memcpy(b, st.b, 32);
                    add_small(a, neg ? UA[e] : -UA[e]);
                    add_small(b, neg ? UB[e] : -UB[e]);
                    sar1(a); sar1(b);
                    push(a, b, st.cost + 1, k, (uint8_t)(e + 1 + 3 * neg));
                }
            }
        }
        if (!live) break;
    }
    GlvDigits r;
    r.n = best_level < 0 ? 0 : best_level;
    for (int L = best_level, k = best_idx; L > 0; --L) {
        r.d[L - 1] = lv[L][k].code;
        k = lv[L][k].parent;
    }
    while (r.n > 0 && r.d[r.n - 1] == 0) --r.n;
    return r;
}

// ------------------------------------------------------------------ SHA3-256 (FIPS 202)
class Sha3_256 {
   public:
    Sha3_256() { std::memset(st_, 0, sizeof st_); }
    void update(const void *data, size_t len) {
        const uint8_t *p = static_cast<const uint8_t *>(data);
        while (len--) {
            reinterpret_cast<uint8_t *>(st_)[pos_++] ^= *p++;
            if (pos_ == kRate) { permute(); pos_ = 0; }
        }
    }
    void finalize(uint8_t out[32]) {
        uint8_t *b = reinterpret_cast<uint8_t *>(st_);
        b[pos_] ^= 0x06;
        b[kRate - 1] ^= 0x80;
        permute();
        std::memcpy(out, st_, 32);
    }

   private:
    static constexpr size_t kRate = 136;
    u64 st_[25];
    size_t pos_ = 0;
    static u64 rol(u64 x, int n) { return (x << n) | (x >> (64 - n)); }
    void permute() {
        static const u64 RC[24] = {
            0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
            0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
            0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
            0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
            0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
            0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
        u64 *a = st_;
        for (int r = 0; r < 24; ++r) {
            u64 c[5], d[5];
            for (int x = 0; x < 5; ++x) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
            for (int x = 0; x < 5; ++x) d[x] = c[(x + 4) % 5] ^ rol(c[(x + 1) % 5], 1);
            for (int i = 0; i < 25; ++i) a[i] ^= d[i % 5];
            // rho + pi
            u64 b[25];
            int x = 1, y = 0;
            b[0] = a[0];
            u64 cur = a[1];
            for (int t = 0; t < 24; ++t) {
                int rot = ((t + 1) * (t + 2) / 2) % 64;
                int nx = y, ny = (2 * x + 3 * y) % 5;
                b[nx + 5 * ny] = rot ? rol(cur, rot) : cur;
                x = nx; y = ny;
                cur = a[x + 5 * y];
            }
            for (int yy = 0; yy < 25; yy += 5)
                for (int xx = 0; xx < 5; ++xx) a[yy + xx] = b[yy + xx] ^ (~b[yy + (xx + 1) % 5] & b[yy + (xx + 2) % 5]);
            a[0] ^= RC[r];
        }
    }
};

// ------------------------------------------------------------------ transcript (group.rs:41-89)
// ark-serialize 0.5 compressed encodings.  UNVERIFIED against the reference (no Rust
// toolchain, no known-answer vector in the reference): "parity unpinned", see DESIGN.md.
class Transcript {
   public:
    void scalar(const Fr &s_mont) {
        Fr c = s_mont.from_mont();
        put(c.l, 32);
    }
    void point(const Point &p) {
        uint8_t out[33] = {0};
        if (p.is_inf()) {
            out[32] = 0x40;
        } else {
            Affine a = p.to_affine();
            Fq x = a.x.from_mont(), y = a.y.from_mont(), ny = (-a.y).from_mont();
            std::memcpy(out, x.l, 32);
            bool y_gt = !Fq::geq(ny.l, y.l);  // y > -y  => "negative" flag
            if (y_gt) out[32] |= 0x80;
        }
        put(out, 33);
    }
    void u64le(u64 v) { put(&v, 8); }
    void byte(uint8_t v) { put(&v, 1); }
    Fr finish(uint32_t tag) {  // tag 0 = rho_0!, 1 = rho_1!
        Sha3_256 h;
        h.update(buf_.data(), buf_.size());
        h.update(&tag, 4);
        uint8_t dig[32];
        h.finalize(dig);
        buf_.clear();
        return Fr::from_le_bytes_mod_order(dig);
    }

   private:
    std::vector<uint8_t> buf_;
    void put(const void *p, size_t n) {
        const uint8_t *b = static_cast<const uint8_t *>(p);
        buf_.insert(buf_.end(), b, b + n);
    }
};

// main.rs:18-32: scalar of generator `index`
inline Fr urs_scalar(u64 index) {
    static const char kGenesis[] = "To understand recursion, one must first understand recursion";
    Sha3_256 h;
    h.update(kGenesis, sizeof(kGenesis) - 1);
    h.update(&index, 8);
    uint8_t dig[32];
    h.finalize(dig);
    return Fr::from_le_bytes_mod_order(dig);
}

// SplitMix64 stream shared with the tests' input generator
struct Rng {
    u64 state;
    u64 next() {
        state += 0x9E3779B97F4A7C15ULL;
        u64 z = state;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        return z ^ (z >> 31);
    }
    Fr scalar() {
        Fr v;
        for (int i = 0; i < 4; ++i) v.l[i] = next();
        while (Fr::geq(v.l, FrP::M)) Fr::sub_limbs(v.l, v.l, FrP::M);
        return v.to_mont();
    }
};

}  // namespace host
}  // namespace halo
