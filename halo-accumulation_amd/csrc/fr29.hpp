// Pallas scalar field Fr for the bandwidth-side kernels (K4-K9: folds of c and z, dot products, powers, p(z), h(X)):
// 9 limbs of 29 bits, lazy Montgomery products with R' = 2^261 -- the same construction as fq29.hpp uses for the base
// field (both Pasta moduli are 2^254 + t 2^32 + 1: limb 0 is 1, limbs 5..7 are 0, limb 8 is 2^22), here for r.
//
// Why: these kernels move 32-byte elements and do ONE field product per element or per pair of elements.  With the
// 8 x 32-bit product of field.hpp (~540 instructions: a carry instruction per partial product) they were compute-bound
// at 0.7-1.0 TB/s; the carry-free 29-bit columns need ~230.  At ~950 cycles per wave and product the chip does
// 1024 SIMDs x 2.4 GHz / 950 x 64 lanes = 1.65e11 products per second: a kernel with one product per 32 bytes is
// VALU-bound at 5.3 TB/s, one with a product per 64 bytes or more is HBM-bound.
//
// Forms.  Data in memory stays what arkworks keeps: x R mod r with R = 2^256, canonical ("A-form").  A product of two
// limb vectors is a b 2^-261, so
//     A(x) * N(k) = A(x k)        with N(k) = k 2^261 mod r  ("N-form": the multiplier the host prepares, = 32 k in A-form)
//     N(a) * N(b) = N(a b)
//     A(x) * A(y) = A(x y) / 32   (dot products: the sum is fixed once with C266 = 2^266: S * C266 = 32 S)
// so the element-wise kernels need no conversion product at all: load, repack 8 x 32 -> 9 x 29 bits, multiply by an
// N-form constant, add, bring below r, repack, store.
#pragma once
#include "fq29.hpp"

namespace halo {

struct R29 {
    static constexpr uint32_t L[9] = {0x1, 0x2375908, 0x52a3763, 0xd31f813, 0x224, 0x0, 0x0, 0x0, 0x400000};
};
// value < K r, limbs 0..7 < 2^29, small top limb
template <int K>
struct Fs {
    uint32_t v[9];
};

// ---- the product: columns of a b (+ c d), nine Montgomery steps (m = -t mod 2^29 because r = 1 mod 2^29), carry pass.
// The low columns start at 2^29 - 1 so that the carry of step i is a plain shift (see fq29.hpp fq_mul).
HALO_DEV void fs_reduce_columns(uint64_t (&c)[18], uint32_t (&out)[9]) {
    uint32_t p8 = R29::L[8];
    asm volatile("" : "+v"(p8));  // m * 2^22 stays a v_mad_u64_u32
    uint64_t carry = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        uint64_t t = c[i] + carry;
        uint32_t m = (~(uint32_t)t) & M29;
        c[i + 1] = (uint64_t)m * R29::L[1] + c[i + 1];
        c[i + 2] = (uint64_t)m * R29::L[2] + c[i + 2];
        c[i + 3] = (uint64_t)m * R29::L[3] + c[i + 3];
        c[i + 4] = (uint64_t)m * R29::L[4] + c[i + 4];
        c[i + 8] = (uint64_t)m * p8 + c[i + 8];
        carry = t >> 29;
    }
    c[9] += carry;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        out[i] = (uint32_t)c[9 + i] & M29;
        c[10 + i] += c[9 + i] >> 29;
    }
    out[8] = (uint32_t)c[17];
}
template <int Ka, int Kb>
HALO_DEV Fs<2> fs_mul(const Fs<Ka> &a, const Fs<Kb> &b) {
    static_assert(Ka * Kb <= 120, "Montgomery product bound: Ka*Kb/128 + 1 must stay < 2");
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
            if (j >= 0 && j < 9) {
                acc = (uint64_t)a.v[i] * b.v[j] + acc;
                if (first) { asm volatile("" : "+v"(acc)); first = false; }
            }
        }
        c[k] = acc;
    }
    c[17] = 0;
    Fs<2> r;
    fs_reduce_columns(c, r.v);
    return r;
}
// a b + c d with one reduction (18 partial products of < 2^58 per column still fit 64 bits)
template <int Ka, int Kb, int Kc, int Kd>
HALO_DEV Fs<2> fs_mul_add_mul(const Fs<Ka> &a, const Fs<Kb> &b, const Fs<Kc> &c2, const Fs<Kd> &d) {
    static_assert(Ka * Kb + Kc * Kd <= 120, "fused product bound");
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
    Fs<2> r;
    fs_reduce_columns(c, r.v);
    return r;
}

// ---- linear operations
template <int Ka, int Kb>
HALO_DEV Fs<Ka + Kb> fs_add(const Fs<Ka> &a, const Fs<Kb> &b) {
    static_assert(Ka + Kb <= 60, "value bound");
    int32_t t[9];
#pragma unroll
    for (int i = 0; i < 9; i++) t[i] = (int32_t)(a.v[i] + b.v[i]);
    Fs<Ka + Kb> r;
    carry_pass(t, r.v);
    return r;
}
template <int K>
HALO_DEV Fs<K> fs_zero() {
    Fs<K> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = 0;
    return r;
}
template <int Kn, int K>
HALO_DEV Fs<Kn> fs_widen(const Fs<K> &a) {
    static_assert(K <= Kn, "cannot narrow a value bound");
    Fs<Kn> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = a.v[i];
    return r;
}
// value < K r (K <= 60) -> value < 2 r: subtract (q - 1) r with q = floor(v / 2^254)
template <int K>
HALO_DEV Fs<2> fs_tighten(const Fs<K> &a) {
    static_assert(K <= 60, "tighten: bound too large");
    int32_t s = (int32_t)(a.v[8] >> 22) - 1;  // q - 1 in [-1, K]
    int64_t w[9];
#pragma unroll
    for (int i = 0; i < 9; i++) w[i] = (int64_t)a.v[i] - (int64_t)s * (int64_t)R29::L[i];
    Fs<2> r;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        r.v[i] = (uint32_t)w[i] & M29;
        w[i + 1] += w[i] >> 29;
    }
    r.v[8] = (uint32_t)w[8];
    return r;
}

// ---- memory forms
// any 256-bit pattern (canonical data is < r; anything else is still < 4 r): repacking only
HALO_DEV Fs<4> fs_from_fe(const Fe &x) {
    Fs<4> r;
    words_to_limbs(x.v, r.v);
    return r;
}
// the canonical representative in [0, r) as 8 x 32-bit words
template <int K>
HALO_DEV Fs<2> fs_below_2r(const Fs<K> &a) { return fs_tighten(a); }
template <>
HALO_DEV Fs<2> fs_below_2r<2>(const Fs<2> &a) { return a; }  // a product's result: nothing to do
template <>
HALO_DEV Fs<2> fs_below_2r<1>(const Fs<1> &a) { Fs<2> r; for (int i = 0; i < 9; i++) r.v[i] = a.v[i]; return r; }
template <int K>
HALO_DEV Fe fs_to_fe(const Fs<K> &a) {
    Fs<2> t = fs_below_2r(a);
    int32_t d[9];
#pragma unroll
    for (int i = 0; i < 9; i++) d[i] = (int32_t)t.v[i] - (int32_t)R29::L[i];
    uint32_t o[9];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        int32_t c = d[i] >> 29;
        o[i] = (uint32_t)d[i] & M29;
        d[i + 1] += c;
    }
    o[8] = (uint32_t)d[8];
    bool neg = d[8] < 0;  // t < r: keep t
    uint32_t l[9];
#pragma unroll
    for (int i = 0; i < 9; i++) l[i] = neg ? t.v[i] : o[i];
    Fe out;
    limbs_to_words(l, out.v);
    return out;
}
HALO_DEV Fs<4> fs_load(const uint64_t *p) { return fs_from_fe(fe_load(p)); }
template <int K>
HALO_DEV void fs_store(uint64_t *p, const Fs<K> &a) { fe_store(p, fs_to_fe(a)); }

// 2^266 mod r: S * C266 = 32 S (puts a sum of A(x) * A(y) products back into A-form)
HALO_DEV Fs<1> fs_c266() {
    constexpr uint32_t C[9] = {0x1ffff001, 0xca6d907, 0x1b40647, 0xdb0c57e, 0x1fddbb8b, 0x1fffffff, 0x1fffffff, 0x1fffffff, 0x3fffff};
    Fs<1> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = C[i];
    return r;
}
template <int K>
HALO_DEV Fs<K> fs_shfl(const Fs<K> &a, int src_lane) {
    Fs<K> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = (uint32_t)__shfl((int)a.v[i], src_lane, 64);
    return r;
}

}  // namespace halo
