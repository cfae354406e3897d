// K10 batch_to_affine, K11 URS generation (main.rs:18-45: G_i = [SHA3-256(genesis || LE64(i)) mod r] (-1, 2), hashes and the
// fixed-base comb on the device) and the conversions between the ABI's arkworks limbs and the native 128-byte base entries.
#include <atomic>
#include <cstring>
#include <thread>

#include "msm_kernels.hpp"

namespace halo {

// ------------------------------------------------------------------------------ K10 / K11 / format conversion
// arkworks Jacobian words -> native affine: TBL_E points per lane, one shared Fermat inversion (jac_batch_to_aff)
__global__ __launch_bounds__(256) void k_batch_to_affine(const uint64_t *__restrict__ jac, uint32_t n, uint32_t *__restrict__ out) {
    uint32_t t = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
    if (t >= n) return;
    JacN p[TBL_E];
    static_for<0, TBL_E>([&](auto ic) {
        constexpr int e = decltype(ic)::value;
        uint32_t i = t + (uint32_t)e * stride;
        p[e] = i < n ? jac_from_words(jac + 12 * (size_t)i) : jac_inf();
    });
    AffN a[TBL_E];
    jac_batch_to_aff(p, a);
    static_for<0, TBL_E>([&](auto ic) {
        constexpr int e = decltype(ic)::value;
        uint32_t i = t + (uint32_t)e * stride;
        if (i < n) aff_store(out + AFF_STRIDE * (size_t)i, a[e]);
    });
}
// arkworks affine words (n x 8 u64) -> native table (n x 20 words)
__global__ __launch_bounds__(256) void k_aff_to_native(const uint64_t *__restrict__ in, uint32_t n, uint32_t *__restrict__ out) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    aff_store(out + AFF_STRIDE * (size_t)i, aff_from_words(in + 8 * (size_t)i));
}
__global__ __launch_bounds__(256) void k_native_to_aff(const uint32_t *__restrict__ in, uint32_t n, uint64_t *__restrict__ out) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    aff_to_words(out + 8 * (size_t)i, aff_load(in + AFF_STRIDE * (size_t)i));
}
// main.rs:18-32 on the device: canon[i] = SHA3-256(genesis || LE64(first_index + i * stride)) read as a little-endian
// integer mod r (ark-ff from_le_bytes_mod_order), as 8 plain 32-bit words -- the scalar of generator i.  One lane per
// hash: Keccak-f[1600] with the 25 lanes in registers (the 68-byte message is one block of the 136-byte rate).
// The host spent ~70-120 ms on the 2^20 hashes of a context; here they take tens of microseconds.
HALO_DEV uint64_t rol64(uint64_t x, int n) { return (x << n) | (x >> (64 - n)); }
__global__ __launch_bounds__(256) void k_urs_scalars(uint64_t first_index, uint64_t stride, uint32_t n, uint32_t *__restrict__ canon) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    constexpr uint64_t RC[24] = {0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
                                 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
                                 0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
                                 0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
                                 0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    constexpr int ROT[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
    constexpr int PIL[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
    // "To understand recursion, one must first understand recursion" (60 bytes), then the index, 0x06 ... 0x80 padding
    uint64_t idx = first_index + (uint64_t)i * stride;
    uint64_t a[25] = {0x7265646e75206f54ULL, 0x657220646e617473ULL, 0x2c6e6f6973727563ULL, 0x73756d20656e6f20ULL, 0x2074737269662074ULL,
                      0x6174737265646e75ULL, 0x727563657220646eULL, 0x6e6f6973ULL | (idx << 32), (idx >> 32) | (0x06ULL << 32), 0, 0, 0, 0, 0, 0, 0,
                      0x8000000000000000ULL, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 1
    for (int r = 0; r < 24; r++) {
        uint64_t c[5], t, bc;
#pragma unroll
        for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
#pragma unroll
        for (int x = 0; x < 5; x++) {
            t = c[(x + 4) % 5] ^ rol64(c[(x + 1) % 5], 1);
#pragma unroll
            for (int y = 0; y < 25; y += 5) a[y + x] ^= t;
        }
        t = a[1];
#pragma unroll
        for (int k = 0; k < 24; k++) {
            bc = a[PIL[k]];
            a[PIL[k]] = rol64(t, ROT[k]);
            t = bc;
        }
#pragma unroll
        for (int y = 0; y < 25; y += 5) {
            uint64_t b0 = a[y], b1 = a[y + 1], b2 = a[y + 2], b3 = a[y + 3], b4 = a[y + 4];
            a[y] = b0 ^ (~b1 & b2); a[y + 1] = b1 ^ (~b2 & b3); a[y + 2] = b2 ^ (~b3 & b4); a[y + 3] = b3 ^ (~b4 & b0); a[y + 4] = b4 ^ (~b0 & b1);
        }
        uint64_t rc = 0;
#pragma unroll
        for (int q = 0; q < 24; q++) rc = (q == r) ? RC[q] : rc;  // (no runtime-indexed constant array: that would live in scratch)
        a[0] ^= rc;
    }
    Fe v;
#pragma unroll
    for (int k = 0; k < 4; k++) { v.v[2 * k] = (uint32_t)a[k]; v.v[2 * k + 1] = (uint32_t)(a[k] >> 32); }
    // 2^256 < 4 r: at most three subtractions bring the digest below r
#pragma unroll 1
    for (int q = 0; q < 3; q++) fe_cond_sub<FrCfg>(v);
#pragma unroll
    for (int k = 0; k < 8; k++) canon[8 * (size_t)i + k] = v.v[k];
}
// table[w][d] = d * 16^w * (-1, 2), d in 0..15 (d = 0 stored as infinity): 64 mixed adds, no doublings
// TBL_E generators per lane (i, i + stride, ...): their 64-step ladders run side by side and share ONE inversion
__global__ __launch_bounds__(256) void k_urs(const uint32_t *__restrict__ table, const uint32_t *__restrict__ canon, uint32_t n,
                                             uint32_t *__restrict__ out) {
    uint32_t t0 = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
    if (t0 >= n) return;
    JacN acc[TBL_E];
    static_for<0, TBL_E>([&](auto ic) { acc[decltype(ic)::value] = jac_inf(); });
#pragma unroll 1
    for (int limb = 0; limb < 8; limb++) {
        uint32_t word[TBL_E];
        static_for<0, TBL_E>([&](auto ic) {
            constexpr int e = decltype(ic)::value;
            uint32_t i = t0 + (uint32_t)e * stride;
            word[e] = i < n ? canon[8 * (size_t)i + limb] : 0u;  // (a lane past the end adds table entry 0 = infinity)
        });
#pragma unroll 1
        for (int k = 0; k < 8; k++) {
            static_for<0, TBL_E>([&](auto ic) {
                constexpr int e = decltype(ic)::value;
                uint32_t nib = (word[e] >> (4 * k)) & 15u;
                AffN t = aff_load(table + AFF_STRIDE * (size_t)((limb * 8 + k) * 16 + nib));
                acc[e] = jac_madd(acc[e], t);
            });
        }
    }
    AffN a[TBL_E];
    jac_batch_to_aff(acc, a);
    static_for<0, TBL_E>([&](auto ic) {
        constexpr int e = decltype(ic)::value;
        uint32_t i = t0 + (uint32_t)e * stride;
        if (i < n) aff_store(out + AFF_STRIDE * (size_t)i, a[e]);
    });
}

int batch_to_affine(halo_ctx *ctx, const uint64_t *d_jac, size_t n, uint32_t *d_out) {
    if (n == 0) return HALO_OK;
    dim3 grid((unsigned)(((n + TBL_E - 1) / TBL_E + 255) / 256)), block(256);
    HALO_LAUNCH(ctx, "k_batch_to_affine", k_batch_to_affine, grid, block, 0, d_jac, (uint32_t)n, d_out);
    HALO_HIP(hipGetLastError());
    return HALO_OK;
}
int aff_words_to_native(halo_ctx *ctx, const uint64_t *d_in, size_t n, uint32_t *d_out) {
    if (n == 0) return HALO_OK;
    HALO_LAUNCH(ctx, "k_aff_to_native", k_aff_to_native, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, d_in, (uint32_t)n, d_out);
    HALO_HIP(hipGetLastError());
    return HALO_OK;
}
int aff_native_to_words(halo_ctx *ctx, const uint32_t *d_in, size_t n, uint64_t *d_out) {
    if (n == 0) return HALO_OK;
    HALO_LAUNCH(ctx, "k_native_to_aff", k_native_to_aff, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, d_in, (uint32_t)n, d_out);
    HALO_HIP(hipGetLastError());
    return HALO_OK;
}

// ------------------------------------------------------------------------------ URS
static const std::vector<uint64_t> &urs_table() {
    // (a function-local static initialised by a lambda: thread-safe, two threads may create their first contexts at once)
    static const std::vector<uint64_t> tbl = [] {
        std::vector<uint64_t> t(64 * 16 * 8, 0);
        host::Point base = host::Point::generator();
        for (int w = 0; w < 64; ++w) {
            host::Point acc = host::Point::infinity();
            for (int d = 1; d < 16; ++d) {
                acc = acc + base;
                host::Affine a = acc.to_affine();
                a.x.store(&t[(size_t)(w * 16 + d) * 8]);
                a.y.store(&t[(size_t)(w * 16 + d) * 8 + 4]);
            }
            base = base.dbl().dbl().dbl().dbl();
        }
        return t;
    }();
    return tbl;
}

int urs_generate(halo_ctx *ctx, uint64_t first_index, uint64_t stride, size_t n, uint32_t *d_out) {
    if (n == 0) return HALO_OK;
    const std::vector<uint64_t> &tbl = urs_table();
    // three temporaries; freed on every path out of this function
    struct Tmp {
        uint64_t *d_tbl = nullptr, *d_canon = nullptr;
        uint32_t *d_tbl_native = nullptr;
        ~Tmp() { (void)hipFree(d_tbl); (void)hipFree(d_tbl_native); (void)hipFree(d_canon); }
    } t;
    HALO_HIP(hipMalloc(&t.d_tbl, tbl.size() * 8));
    HALO_HIP(hipMalloc(&t.d_tbl_native, (size_t)1024 * AFF_STRIDE * 4));
    HALO_HIP(hipMalloc(&t.d_canon, n * 32));
    HALO_HIP(hipMemcpyAsync(t.d_tbl, tbl.data(), tbl.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    // main.rs:18-32: the n SHA3-256 hashes and their reduction mod r, on the device
    HALO_LAUNCH(ctx, "k_urs_scalars", k_urs_scalars, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, first_index, stride, (uint32_t)n,
                reinterpret_cast<uint32_t *>(t.d_canon));
    int rc = aff_words_to_native(ctx, t.d_tbl, 1024, t.d_tbl_native);
    if (rc) { (void)hipStreamSynchronize(ctx->stream); return rc; }
    dim3 grid((unsigned)(((n + TBL_E - 1) / TBL_E + 255) / 256)), block(256);
    HALO_LAUNCH(ctx, "k_urs", k_urs, grid, block, 0, t.d_tbl_native, reinterpret_cast<const uint32_t *>(t.d_canon), (uint32_t)n, d_out);
    hipError_t e1 = hipGetLastError(), e2 = hipStreamSynchronize(ctx->stream);  // the temporaries are in use until here
    if (e1 != hipSuccess) return hip_fail(e1, "k_urs launch");
    if (e2 != hipSuccess) return hip_fail(e2, "hipStreamSynchronize");
    return HALO_OK;
}

}  // namespace halo
