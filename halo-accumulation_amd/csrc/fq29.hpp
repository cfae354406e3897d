// Pallas base field Fq for the curve kernels: 9 limbs of 29 bits in 32-bit registers, lazy
// Montgomery arithmetic with R' = 2^261.
//
// Why not the 8 x 32-bit form of field.hpp (kept for Fr and for the C ABI)?  Measured on gfx950
// (profiles/r01_microbench_instr_throughput.txt): v_mad_u64_u32 issues in ~5.3 cycles per wave,
// but every carry instruction (v_add_co/v_addc_co, v_lshl_add_u64) costs ~4.6 and the 32-bit-limb
// product needs one of those per partial product plus register shuffling for the even-aligned
// 64-bit operands: ~540 instructions per Montgomery product.  With 29-bit limbs nine partial
// products (each < 2^58) fit a 64-bit accumulator without any carry, so a column is a plain
// chain of v_mad_u64_u32: ~230 instructions per product, and field additions are 9 plain adds.
//
// Lazy values.  Fq<K> holds a value < K*p (not reduced mod p) with limbs 0..7 < 2^29 and a
// small top limb.  Because R' = 2^261 = 128 * 2^254, a product of operands < Ka*p and < Kb*p
// comes out < (Ka*Kb/128 + 1) p, so with Ka*Kb <= 120 no conditional subtraction is ever needed
// and the result is an Fq<2>.  K is a template parameter: every bound in the group law is
// checked by the compiler (static_assert), not by hand.
//
// Representation of x in Fq: X = x * 2^261 mod p.  The arkworks/C-ABI form is x * 2^256
// (field.hpp); fq_from_words / fq_to_words convert with one multiplication each, and the base
// tables live in HBM in the native form (20 words per affine point) so the conversion is paid
// once per table, not per use.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "field.hpp"

namespace halo {

constexpr uint32_t M29 = (1u << 29) - 1u;

struct Limbs9 {
    uint32_t v[9];
};
// k * p as normalised radix-2^29 limbs (top limb unmasked), k <= 64
constexpr Limbs9 make_kp(uint32_t k) {
    // p = 2^254 + 0x224698fc094cf91b992d30ed * 2^32 + 1
    const uint64_t pl[9] = {0x1, 0x9698768, 0x133e46e6, 0xd31f812, 0x224, 0x0, 0x0, 0x0, 0x400000};
    Limbs9 r{};
    uint64_t carry = 0;
    for (int i = 0; i < 9; i++) {
        uint64_t t = pl[i] * k + carry;
        if (i < 8) { r.v[i] = (uint32_t)(t & M29); carry = t >> 29; }
        else r.v[i] = (uint32_t)t;
    }
    return r;
}
struct P29 {
    static constexpr uint32_t L[9] = {0x1, 0x9698768, 0x133e46e6, 0xd31f812, 0x224, 0x0, 0x0, 0x0, 0x400000};
};
template <uint32_t k>
struct KP {
    static constexpr Limbs9 value = make_kp(k);
};

template <int K>
struct Fq {
    uint32_t v[9];
};

template <int K>
HALO_DEV Fq<K> fq_zero() {
    Fq<K> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = 0;
    return r;
}
template <int Kn, int K>
HALO_DEV Fq<Kn> fq_widen(const Fq<K> &a) {
    static_assert(K <= Kn, "cannot narrow a value bound");
    Fq<Kn> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = a.v[i];
    return r;
}
// exact limb test: only meaningful where "zero" is represented by all-zero limbs (infinity flags)
template <int K>
HALO_DEV bool fq_limbs_zero(const Fq<K> &a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) o |= a.v[i];
    return o == 0;
}

// signed carry pass: t[i] are int32 limb sums of a non-negative total value
HALO_DEV void carry_pass(int32_t (&t)[9], uint32_t (&out)[9]) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
        int32_t c = t[i] >> 29;  // arithmetic shift: floor
        out[i] = (uint32_t)t[i] & M29;
        t[i + 1] += c;
    }
    out[8] = (uint32_t)t[8];
}

// ------------------------------------------------------------------ multiplication
// columns c[0..16] of a*b, then nine 29-bit Montgomery steps with m = -c_i mod 2^29 (p = 1 mod 2^29)
template <int Ka, int Kb>
HALO_DEV Fq<2> fq_mul(const Fq<Ka> &a, const Fq<Kb> &b) {
    static_assert(Ka * Kb <= 120, "Montgomery product bound: Ka*Kb/128 + 1 must stay < 2");
    uint64_t c[18];
    uint64_t k29 = M29;
    asm volatile("" : "+v"(k29));  // a register pair, so it is the free addend of each column's first mad
#pragma unroll
    for (int k = 0; k < 17; k++) {
        uint64_t acc = k < 9 ? k29 : 0;  // 2^29 - 1 in the columns the reduction consumes
        bool first = k < 9;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            int j = k - i;
            if (j >= 0 && j < 9) {
                acc = (uint64_t)a.v[i] * b.v[j] + acc;
                if (first) { asm volatile("" : "+v"(acc)); first = false; }  // keep k29 as this mad's addend (no reassociation)
            }
        }
        c[k] = acc;
    }
    c[17] = 0;
    uint32_t p8 = P29::L[8];
    asm volatile("" : "+v"(p8));  // keep m * 2^22 on v_mad_u64_u32 (5.3 cycles) instead of a 64-bit shift + add (9.2)
    // Step i: t = c[i] + carry, m = -t mod 2^29, carry' = (t + m) >> 29 = ceil(t / 2^29).  The low
    // columns were started at 2^29 - 1 (see above), so here t already holds t + 2^29 - 1: the carry is
    // a plain shift and m = ~t mod 2^29 -- no "c[i] += m" and no zero-extension of m.
    uint64_t carry = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        uint64_t t = c[i] + carry;
        uint32_t m = (~(uint32_t)t) & M29;
        c[i + 1] = (uint64_t)m * P29::L[1] + c[i + 1];
        c[i + 2] = (uint64_t)m * P29::L[2] + c[i + 2];
        c[i + 3] = (uint64_t)m * P29::L[3] + c[i + 3];
        c[i + 4] = (uint64_t)m * P29::L[4] + c[i + 4];
        c[i + 8] = (uint64_t)m * p8 + c[i + 8];
        carry = t >> 29;
    }
    c[9] += carry;
    Fq<2> r;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        r.v[i] = (uint32_t)c[9 + i] & M29;
        c[10 + i] += c[9 + i] >> 29;
    }
    r.v[8] = (uint32_t)c[17];
    return r;
}
// squaring: 45 instead of 81 partial products (cross terms use the doubled limb, < 2^30)
template <int Ka>
HALO_DEV Fq<2> fq_sqr(const Fq<Ka> &a) {
    static_assert(Ka * Ka <= 120, "Montgomery square bound");
    uint32_t d[9];
#pragma unroll
    for (int i = 0; i < 9; i++) d[i] = a.v[i] << 1;
    uint64_t c[18];
    uint64_t k29 = M29;
    asm volatile("" : "+v"(k29));
#pragma unroll
    for (int k = 0; k < 17; k++) {
        uint64_t acc = k < 9 ? k29 : 0;
        bool first = k < 9;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            int j = k - i;
            if (j >= 0 && j < 9 && i < j) {
                acc = (uint64_t)d[i] * a.v[j] + acc;
                if (first) { asm volatile("" : "+v"(acc)); first = false; }
            }
            if (j >= 0 && j < 9 && i == j) {
                acc = (uint64_t)a.v[i] * a.v[i] + acc;
                if (first) { asm volatile("" : "+v"(acc)); first = false; }
            }
        }
        c[k] = acc;
    }
    c[17] = 0;
    uint32_t p8 = P29::L[8];
    asm volatile("" : "+v"(p8));  // keep m * 2^22 on v_mad_u64_u32 (5.3 cycles) instead of a 64-bit shift + add (9.2)
    // Step i: t = c[i] + carry, m = -t mod 2^29, carry' = (t + m) >> 29 = ceil(t / 2^29).  The low
    // columns were started at 2^29 - 1 (see above), so here t already holds t + 2^29 - 1: the carry is
    // a plain shift and m = ~t mod 2^29 -- no "c[i] += m" and no zero-extension of m.
    uint64_t carry = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        uint64_t t = c[i] + carry;
        uint32_t m = (~(uint32_t)t) & M29;
        c[i + 1] = (uint64_t)m * P29::L[1] + c[i + 1];
        c[i + 2] = (uint64_t)m * P29::L[2] + c[i + 2];
        c[i + 3] = (uint64_t)m * P29::L[3] + c[i + 3];
        c[i + 4] = (uint64_t)m * P29::L[4] + c[i + 4];
        c[i + 8] = (uint64_t)m * p8 + c[i + 8];
        carry = t >> 29;
    }
    c[9] += carry;
    Fq<2> r;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        r.v[i] = (uint32_t)c[9 + i] & M29;
        c[10 + i] += c[9 + i] >> 29;
    }
    r.v[8] = (uint32_t)c[17];
    return r;
}

// a*b + c*d with ONE Montgomery reduction: both products accumulate into the same 64-bit columns
// (18 partial products of < 2^58 still fit) before the nine reduction steps.  Differences of
// products are formed by passing a negated (K*p - x) operand.
template <int Ka, int Kb, int Kc, int Kd>
HALO_DEV Fq<2> fq_mul_add_mul(const Fq<Ka> &a, const Fq<Kb> &b, const Fq<Kc> &c2, const Fq<Kd> &d) {
    static_assert(Ka * Kb + Kc * Kd <= 120, "fused product bound: (KaKb + KcKd)/128 + 1 must stay < 2");
    uint64_t c[18];
    uint64_t k29 = M29;
    asm volatile("" : "+v"(k29));
#pragma unroll
    for (int k = 0; k < 17; k++) {
        uint64_t acc = k < 9 ? k29 : 0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            int j = k - i;
            if (j >= 0 && j < 9) {
                acc = (uint64_t)a.v[i] * b.v[j] + acc;
                acc = (uint64_t)c2.v[i] * d.v[j] + acc;
            }
        }
        c[k] = acc;
    }
    c[17] = 0;
    uint32_t p8 = P29::L[8];
    asm volatile("" : "+v"(p8));
    uint64_t carry = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        uint64_t t = c[i] + carry;
        uint32_t m = (~(uint32_t)t) & M29;
        c[i + 1] = (uint64_t)m * P29::L[1] + c[i + 1];
        c[i + 2] = (uint64_t)m * P29::L[2] + c[i + 2];
        c[i + 3] = (uint64_t)m * P29::L[3] + c[i + 3];
        c[i + 4] = (uint64_t)m * P29::L[4] + c[i + 4];
        c[i + 8] = (uint64_t)m * p8 + c[i + 8];
        carry = t >> 29;
    }
    c[9] += carry;
    Fq<2> r;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        r.v[i] = (uint32_t)c[9 + i] & M29;
        c[10 + i] += c[9 + i] >> 29;
    }
    r.v[8] = (uint32_t)c[17];
    return r;
}

// ------------------------------------------------------------------ linear operations (one carry pass each)
template <int Ka, int Kb>
HALO_DEV Fq<Ka + Kb> fq_add(const Fq<Ka> &a, const Fq<Kb> &b) {
    int32_t t[9];
#pragma unroll
    for (int i = 0; i < 9; i++) t[i] = (int32_t)(a.v[i] + b.v[i]);
    Fq<Ka + Kb> r;
    carry_pass(t, r.v);
    return r;
}
// a - b + Kc*p, Kc >= bound of b
template <int Kc, int Ka, int Kb>
HALO_DEV Fq<Ka + Kc> fq_sub(const Fq<Ka> &a, const Fq<Kb> &b) {
    static_assert(Kb <= Kc, "subtrahend may exceed the added multiple of p");
    int32_t t[9];
#pragma unroll
    for (int i = 0; i < 9; i++) t[i] = (int32_t)a.v[i] - (int32_t)b.v[i] + (int32_t)KP<Kc>::value.v[i];
    Fq<Ka + Kc> r;
    carry_pass(t, r.v);
    return r;
}
// a - b - 2c + (Kb + 2Kc)*p     (x3 = R^2 - PPP - 2Q and friends)
template <int Ka, int Kb, int Kc>
HALO_DEV Fq<Ka + Kb + 2 * Kc> fq_sub_sub2(const Fq<Ka> &a, const Fq<Kb> &b, const Fq<Kc> &c) {
    int32_t t[9];
#pragma unroll
    for (int i = 0; i < 9; i++)
        t[i] = (int32_t)a.v[i] - (int32_t)b.v[i] - 2 * (int32_t)c.v[i] + (int32_t)KP<Kb + 2 * Kc>::value.v[i];
    Fq<Ka + Kb + 2 * Kc> r;
    carry_pass(t, r.v);
    return r;
}
// k * a for a small constant k (unsigned carry pass: limbs < 2^29 so k*limb + carry < 2^32 for k <= 8)
template <int k, int Ka>
HALO_DEV Fq<k * Ka> fq_muls(const Fq<Ka> &a) {
    static_assert(k >= 1 && k <= 8, "small multiplier");
    static_assert(k * Ka <= 60, "value bound: the top limb must stay below 2^28");
    Fq<k * Ka> r;
    uint32_t carry = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint32_t t = a.v[i] * (uint32_t)k + carry;
        r.v[i] = t & M29;
        carry = t >> 29;
    }
    r.v[8] = a.v[8] * (uint32_t)k + carry;
    return r;
}
// Kc*p - a
template <int Kc, int Ka>
HALO_DEV Fq<Kc> fq_neg(const Fq<Ka> &a) {
    static_assert(Ka <= Kc, "negation constant too small");
    int32_t t[9];
#pragma unroll
    for (int i = 0; i < 9; i++) t[i] = (int32_t)KP<Kc>::value.v[i] - (int32_t)a.v[i];
    Fq<Kc> r;
    carry_pass(t, r.v);
    return r;
}

// value < K*p (K <= 32) -> value < 2p: subtract (q - 1) p with q = floor(v / 2^254), in 64-bit columns
template <int K>
HALO_DEV Fq<2> fq_tighten(const Fq<K> &a) {
    static_assert(K <= 60, "tighten: bound too large");
    int32_t s = (int32_t)(a.v[8] >> 22) - 1;  // q - 1 in [-1, K]
    int64_t w[9];
#pragma unroll
    for (int i = 0; i < 9; i++) w[i] = (int64_t)a.v[i] - (int64_t)s * (int64_t)P29::L[i];
    Fq<2> r;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        r.v[i] = (uint32_t)w[i] & M29;
        w[i + 1] += w[i] >> 29;
    }
    r.v[8] = (uint32_t)w[8];
    return r;
}

// fully reduced representative in [0, p)
template <int K>
HALO_DEV Fq<1> fq_canonical(const Fq<K> &a) {
    Fq<2> t = fq_tighten(a);
    // t < 2p: subtract p once if t >= p
    int32_t d[9];
#pragma unroll
    for (int i = 0; i < 9; i++) d[i] = (int32_t)t.v[i] - (int32_t)P29::L[i];
    uint32_t o[9];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        int32_t c = d[i] >> 29;
        o[i] = (uint32_t)d[i] & M29;
        d[i + 1] += c;
    }
    o[8] = (uint32_t)d[8];
    bool neg = d[8] < 0;
    Fq<1> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = neg ? t.v[i] : o[i];
    return r;
}
// a == 0 (mod p).  k*p = k (mod 2^29), so limb 0 >= K rules it out without a reduction.
template <int K>
HALO_DEV bool fq_is_zero_modp(const Fq<K> &a) {
    if (a.v[0] >= (uint32_t)K) return false;
    return fq_limbs_zero(fq_canonical(a));
}
template <int Ka, int Kb>
HALO_DEV bool fq_eq_modp(const Fq<Ka> &a, const Fq<Kb> &b) {
    return fq_is_zero_modp(fq_sub<Kb>(a, b));
}

// ------------------------------------------------------------------ constants and conversions
// 1 in native form: 2^261 mod p
HALO_DEV Fq<1> fq_one() {
    Fq<1> r;
    constexpr uint32_t ONE[9] = {0x1fffff81, 0x14a5d367, 0x141ad3c0, 0x1435eec5, 0x1ffeefef, 0x1fffffff, 0x1fffffff, 0x1fffffff, 0x3fffff};
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = ONE[i];
    return r;
}
// radix-2^32 words -> radix-2^29 limbs (no arithmetic)
HALO_DEV void words_to_limbs(const uint32_t (&w)[8], uint32_t (&l)[9]) {
#pragma unroll
    for (int i = 0; i < 9; i++) {
        int bit = 29 * i, wi = bit >> 5, sh = bit & 31;
        uint64_t two = (uint64_t)w[wi] | (wi + 1 < 8 ? ((uint64_t)w[wi + 1] << 32) : 0ull);
        l[i] = (uint32_t)(two >> sh) & (i < 8 ? M29 : 0xffffffffu);
    }
}
HALO_DEV void limbs_to_words(const uint32_t (&l)[9], uint32_t (&w)[8]) {
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        int bit = 29 * i, wi = bit >> 5, sh = bit & 31;
        uint64_t v = (uint64_t)l[i] << sh;
        w[wi] |= (uint32_t)v;
        if (wi + 1 < 8) w[wi + 1] |= (uint32_t)(v >> 32);
    }
}
// arkworks form (x * 2^256, radix 2^32, canonical) -> native: multiply by 2^266 (times 2^-261 from the product)
HALO_DEV Fq<2> fq_from_words(const Fe &x) {
    Fq<4> raw;  // any 256-bit pattern is < 4p
    words_to_limbs(x.v, raw.v);
    constexpr uint32_t C_IN[9] = {0x1ffff001, 0x10f30767, 0xecfe231, 0xdb0ce73, 0x1fddbb8b, 0x1fffffff, 0x1fffffff, 0x1fffffff, 0x3fffff};
    Fq<1> c;
#pragma unroll
    for (int i = 0; i < 9; i++) c.v[i] = C_IN[i];
    return fq_mul(raw, c);
}
// native -> arkworks form, canonical: multiply by 2^256 (times 2^-261) = divide by 32
template <int K>
HALO_DEV Fe fq_to_words(const Fq<K> &a) {
    static_assert(K <= 60, "to_words bound");
    constexpr uint32_t C_OUT[9] = {0x1ffffffd, 0x3c369c7, 0x6452b4d, 0x186a17c8, 0x1ffff992, 0x1fffffff, 0x1fffffff, 0x1fffffff, 0x3fffff};
    Fq<1> c;
#pragma unroll
    for (int i = 0; i < 9; i++) c.v[i] = C_OUT[i];
    Fq<1> r = fq_canonical(fq_mul(fq_widen<60>(a), c));
    Fe o;
    limbs_to_words(r.v, o.v);
    return o;
}

// a^(p-2) for a != 0 mod p
template <int K>
HALO_DEV Fq<2> fq_inv(const Fq<K> &a0) {
    Fq<2> a = fq_tighten(a0);
    Fq<2> acc = fq_widen<2>(fq_one());
#pragma unroll 1
    for (int w = 7; w >= 0; w--) {
        uint32_t e = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) e = (k == w) ? FqCfg::PM2[k] : e;
#pragma unroll 1
        for (int bit = 31; bit >= 0; bit--) {
            acc = fq_sqr(acc);
            if ((e >> bit) & 1) acc = fq_mul(acc, a);
        }
    }
    return acc;
}

// native limbs in memory: 10 words per element (9 limbs + pad), 16-byte aligned pairs
template <int K>
HALO_DEV void fq_store_native(uint32_t *p, const Fq<K> &a) {
#pragma unroll
    for (int i = 0; i < 9; i++) p[i] = a.v[i];
    p[9] = 0;
}

}  // namespace halo
