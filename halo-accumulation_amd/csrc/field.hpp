// Pallas base field Fq and scalar field Fr on gfx950: 8 x 32-bit limbs, Montgomery form,
// R = 2^256 -- bit-identical to the arkworks in-memory representation the reference keeps
// (code/src/consts.rs:4-21: BigInt<4> u64 limbs, little-endian), so values cross the C ABI
// without conversion.
//
// Both Pallas moduli are 2^254 + t*2^32 + 1 with a ~94-bit t: limb 0 is 1 (so the
// Montgomery factor is -1 and m = -t0 costs no multiply), limbs 4..6 are 0 and limb 7 is
// 2^30.  With the modulus as constexpr the reduction folds to 3 v_mad_u64_u32 + shifts per
// step instead of 8 multiplies.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace halo {

struct Fe {
    uint32_t v[8];
};

struct FqCfg {  // ark_pallas::Fq (group.rs:7-8 coordinates)
    static constexpr uint32_t P[8] = {0x00000001u, 0x992d30edu, 0x094cf91bu, 0x224698fcu, 0u, 0u, 0u, 0x40000000u};
    static constexpr uint32_t ONE[8] = {0xfffffffdu, 0x34786d38u, 0xe41914adu, 0x992c350bu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0x3fffffffu};
    static constexpr uint32_t R2[8] = {0x0000000fu, 0x8c78ecb3u, 0x8b0de0e7u, 0xd7d30dbdu, 0xc3c95d18u, 0x7797a99bu, 0x7b9cb714u, 0x096d41afu};
    static constexpr uint32_t PM2[8] = {0xffffffffu, 0x992d30ecu, 0x094cf91bu, 0x224698fcu, 0u, 0u, 0u, 0x40000000u};
};
struct FrCfg {  // ark_pallas::Fr (group.rs:9 PallasScalar)
    static constexpr uint32_t P[8] = {0x00000001u, 0x8c46eb21u, 0x0994a8ddu, 0x224698fcu, 0u, 0u, 0u, 0x40000000u};
    static constexpr uint32_t ONE[8] = {0xfffffffdu, 0x5b2b3e9cu, 0xe3420567u, 0x992c350bu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0x3fffffffu};
    static constexpr uint32_t R2[8] = {0x0000000fu, 0xfc9678ffu, 0x891a16e3u, 0x67bb433du, 0x04ccf590u, 0x7fae2310u, 0x7ccfdaa9u, 0x096d41afu};
    static constexpr uint32_t PM2[8] = {0xffffffffu, 0x8c46eb20u, 0x0994a8ddu, 0x224698fcu, 0u, 0u, 0u, 0x40000000u};
};

#define HALO_DEV __device__ __forceinline__

HALO_DEV Fe fe_zero() {
    Fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = 0;
    return r;
}
template <class F>
HALO_DEV Fe fe_one() {
    Fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = F::ONE[i];
    return r;
}
HALO_DEV bool fe_is_zero(const Fe &a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.v[i];
    return o == 0;
}
HALO_DEV bool fe_eq(const Fe &a, const Fe &b) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.v[i] ^ b.v[i];
    return o == 0;
}

// r = a - P if a >= P else a   (a < 2P)
template <class F>
HALO_DEV void fe_cond_sub(Fe &a) {
    uint32_t d[8];
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t s = (uint64_t)a.v[i] - F::P[i] - br;
        d[i] = (uint32_t)s;
        br = (s >> 32) & 1;
    }
    bool keep = br != 0;  // a < P
#pragma unroll
    for (int i = 0; i < 8; i++) a.v[i] = keep ? a.v[i] : d[i];
}

template <class F>
HALO_DEV Fe fe_add(const Fe &a, const Fe &b) {
    Fe r;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t s = (uint64_t)a.v[i] + b.v[i] + c;
        r.v[i] = (uint32_t)s;
        c = s >> 32;
    }
    fe_cond_sub<F>(r);  // a + b < 2P < 2^256
    return r;
}
template <class F>
HALO_DEV Fe fe_sub(const Fe &a, const Fe &b) {
    Fe r;
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t s = (uint64_t)a.v[i] - b.v[i] - br;
        r.v[i] = (uint32_t)s;
        br = (s >> 32) & 1;
    }
    uint32_t mask = (uint32_t)0 - (uint32_t)br;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t s = (uint64_t)r.v[i] + (F::P[i] & mask) + c;
        r.v[i] = (uint32_t)s;
        c = s >> 32;
    }
    return r;
}
template <class F>
HALO_DEV Fe fe_neg(const Fe &a) {
    Fe r;
    uint64_t br = 0;
    bool z = fe_is_zero(a);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t s = (uint64_t)F::P[i] - a.v[i] - br;
        r.v[i] = z ? 0u : (uint32_t)s;
        br = (s >> 32) & 1;
    }
    return r;
}
template <class F>
HALO_DEV Fe fe_dbl(const Fe &a) {
    return fe_add<F>(a, a);
}

// One Montgomery reduction step on t[0..8]: t = (t + m*P) >> 32 with m = -t[0] (P[0] = 1, -P^-1 = -1 mod 2^32).
template <class F>
HALO_DEV void mont_step(uint32_t (&t)[9]) {
    uint32_t m = 0u - t[0];
    uint64_t c = (t[0] != 0) ? 1u : 0u;  // t[0] + m = 2^32 or 0
#pragma unroll
    for (int j = 1; j < 8; j++) {
        uint64_t s = (uint64_t)m * F::P[j] + t[j] + c;
        t[j - 1] = (uint32_t)s;
        c = s >> 32;
    }
    uint64_t s = (uint64_t)t[8] + c;
    t[7] = (uint32_t)s;
    t[8] = (uint32_t)(s >> 32);
}

// Montgomery product a*b*R^-1 mod P (operand scanning, reduction interleaved).
template <class F>
HALO_DEV Fe fe_mul(const Fe &a, const Fe &b) {
    uint32_t t[9];
#pragma unroll
    for (int i = 0; i < 9; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint64_t s = (uint64_t)a.v[j] * b.v[i] + t[j] + c;
            t[j] = (uint32_t)s;
            c = s >> 32;
        }
        // t < 2P + P*2^32 < 2^288: nine limbs hold every intermediate, no tenth limb needed
        t[8] += (uint32_t)c;
        mont_step<F>(t);
    }
    Fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = t[i];
    fe_cond_sub<F>(r);  // t < 2P: t[8] == 0 because 2P < 2^256
    return r;
}
template <class F>
HALO_DEV Fe fe_sqr(const Fe &a) {
    return fe_mul<F>(a, a);
}
// out of Montgomery form: a * 1 * R^-1
template <class F>
HALO_DEV Fe fe_from_mont(const Fe &a) {
    uint32_t t[9];
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = a.v[i];
    t[8] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        mont_step<F>(t);
    }
    Fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = t[i];
    fe_cond_sub<F>(r);
    return r;
}
template <class F>
HALO_DEV Fe fe_to_mont(const Fe &a) {
    Fe r2;
#pragma unroll
    for (int i = 0; i < 8; i++) r2.v[i] = F::R2[i];
    return fe_mul<F>(a, r2);
}
// a^(P-2); a != 0.  ~255 squarings + popcount(P-2) multiplies, uniform control flow.
template <class F>
HALO_DEV Fe fe_inv(const Fe &a) {
    Fe acc = fe_one<F>();
    for (int w = 7; w >= 0; w--) {
        uint32_t e = F::PM2[0];
        // select limb without a runtime-indexed constexpr array
#pragma unroll
        for (int k = 0; k < 8; k++) e = (k == w) ? F::PM2[k] : e;
        for (int bit = 31; bit >= 0; bit--) {
            acc = fe_sqr<F>(acc);
            if ((e >> bit) & 1) acc = fe_mul<F>(acc, a);
        }
    }
    return acc;
}

// 16-byte vector loads/stores of one 32-byte element (two dwordx4 per lane)
HALO_DEV Fe fe_load(const uint64_t *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    uint4 lo = q[0], hi = q[1];
    Fe r;
    r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w;
    r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
    return r;
}
HALO_DEV void fe_store(uint64_t *p, const Fe &a) {
    uint4 *q = reinterpret_cast<uint4 *>(p);
    q[0] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]);
    q[1] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
}

}  // namespace halo
