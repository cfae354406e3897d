// Fixed-base comb tables for the FIRST fold of pcdl::open (pcdl.rs:216-219 applied twice, ipa.hip k_fold_points4):
//     G''[j] = G[j] + s1 G[j+m] + s2 G[j+2m] + s3 G[j+3m],   m = n / 4,
// where G is the context's own key -- a constant (consts.rs:68), like the bases of the MSM tables of msm.hip.  The generic
// kernel multiplies each of the three points by its scalar with a shared 128-step doubling chain (Straus over GLV digit
// strings: ~128 doublings + ~213 mixed additions, ~3240 field products per output).  With
//     E[w][d][i] = d * 16^w * G_i   (w < 64, d = 1..8, i in [n/4, n), affine, 64 bytes each, 32 KiB per point)
// a scalar multiple is 64 table entries added up (signed base-16 digits, no doublings at all): ~180 mixed additions per
// output, ~2000 products -- the pass over 2^18 outputs goes from 5.3 to ~3.3 ms.  The digits depend on the challenges only,
// i.e. they are the same for every lane of a wave: entry (w, d) of consecutive points is read by consecutive lanes, so the
// layout [w][d][i] makes every gather a fully coalesced 4 KiB read (3 GB per pass at n = 2^20).
//
// Cost: 25.8 GB at n = 2^20 (of 288 GB) and ~0.1 s to build (512 group operations and one share of an inversion per table
// entry) -- forty opens' worth of savings.  So the table is built on the SECOND full-size open of a context (mode -1,
// default; halo_set_fold_table: 1 = at the first, 0 = never), which a prover chain (acc.rs:190-228: two opens per step)
// reaches at once and a single open never does.  No memory, no table: the generic kernel gives the same points.
#include "curve.hpp"
#include "internal.hpp"

namespace halo {

constexpr int FT_WINDOWS = 64, FT_MULT = 8, FT_ENTRIES = FT_WINDOWS * FT_MULT;
constexpr int FT_WORDS = 16;  // x | y, canonical native-form values (x 2^261 mod p) as 8 x 32-bit words each; all zero = infinity

// ---- packed entries
HALO_DEV void ft_store(uint32_t *p, const AffN &a) {
    uint32_t w[16];
    if (aff_is_inf(a)) {
#pragma unroll
        for (int i = 0; i < 16; i++) w[i] = 0;
    } else {
        Fq<1> x = fq_canonical(a.x), y = fq_canonical(a.y);
        uint32_t wx[8], wy[8];
        limbs_to_words(x.v, wx);
        limbs_to_words(y.v, wy);
#pragma unroll
        for (int i = 0; i < 8; i++) { w[i] = wx[i]; w[8 + i] = wy[i]; }
    }
    uint4 *q = reinterpret_cast<uint4 *>(p);
#pragma unroll
    for (int i = 0; i < 4; i++) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}
HALO_DEV AffN ft_load(const uint32_t *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    uint4 a = q[0], b = q[1], c = q[2], d = q[3];
    uint32_t wx[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}, wy[8] = {c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
    AffN r;
    words_to_limbs(wx, r.x.v);
    words_to_limbs(wy, r.y.v);
    return r;  // (all-zero words = all-zero limbs = the infinity marker of AffN)
}

// ---- build: lane s of a slice takes point i = first + s.  Forward: the 512 multiples in XYZZ form, window by window
// (P, 2P, 3P = 2P + P, 4P = 2 (2P), 5P = 4P + P, 6P = 2 (3P), 7P = 6P + P, 8P = 2 (4P); the next window starts at 2 (8P)),
// each written to tmp[e][s] next to the running product of the ZZZ's.  One inversion per point.  Backward: 1 / ZZZ_e from
// the running products, x = X ZZ^2 / ZZZ^2 (ZZ^3 = ZZZ^2), y = Y / ZZZ, packed into E[e][i - lo].
constexpr int FT_TMP_WORDS = 50;  // XYZZ (40) + running product (10)
__global__ __launch_bounds__(256) void k_foldtab_build(const uint32_t *__restrict__ bases, uint32_t first, uint32_t count, uint32_t lo, uint32_t cnt,
                                                       uint32_t *__restrict__ tmp, uint32_t *__restrict__ tab) {
    uint32_t s = blockIdx.x * 256 + threadIdx.x;
    if (s >= count) return;
    auto slot = [&](int e) { return tmp + ((size_t)e * count + s) * FT_TMP_WORDS; };
    XyzzN P1 = xyzz_from_aff(aff_load(bases + AFF_STRIDE * (size_t)(first + s)));
    Fq<2> run = fq_widen<2>(fq_one());
    auto emit = [&](int e, const XyzzN &p) {
        uint32_t *o = slot(e);
        xyzz_store(o, p);
        if (!xyzz_is_inf(p)) run = fq_mul(run, p.zzz);
        fq_store_native(o + 40, run);
    };
#pragma unroll 1
    for (int w = 0; w < FT_WINDOWS; w++) {
        int e0 = w * FT_MULT;
        emit(e0, P1);
        XyzzN t = xyzz_dbl(P1);  // 2P
        emit(e0 + 1, t);
        xyzz_add(t, P1);         // 3P
        emit(e0 + 2, t);
        t = xyzz_dbl(xyzz_load(slot(e0 + 1)));  // 4P
        emit(e0 + 3, t);
        xyzz_add(t, P1);         // 5P
        emit(e0 + 4, t);
        t = xyzz_dbl(xyzz_load(slot(e0 + 2)));  // 6P
        emit(e0 + 5, t);
        xyzz_add(t, P1);         // 7P
        emit(e0 + 6, t);
        t = xyzz_dbl(xyzz_load(slot(e0 + 3)));  // 8P
        emit(e0 + 7, t);
        P1 = xyzz_dbl(t);        // 16 P: the next window's unit
    }
    Fq<2> inv = fq_inv(run);
#pragma unroll 1
    for (int e = FT_ENTRIES - 1; e >= 0; e--) {
        const uint32_t *o = slot(e);
        XyzzN p = xyzz_load(o);
        AffN a = aff_inf();
        if (!xyzz_is_inf(p)) {
            Fq<2> before = e > 0 ? fq_load_native<2>(slot(e - 1) + 40) : fq_widen<2>(fq_one());
            Fq<2> iz = fq_mul(inv, before);  // 1 / ZZZ_e
            inv = fq_mul(inv, p.zzz);
            Fq<2> t = fq_mul(fq_sqr(p.zz), fq_sqr(iz));  // ZZ^2 / ZZZ^2 = 1 / ZZ
            a.x = fq_mul(p.x, t);
            a.y = fq_mul(p.y, iz);
        }
        ft_store(tab + ((size_t)e * cnt + (first + s - lo)) * FT_WORDS, a);
    }
}

// ---- the fold: digits (signed base 16, one byte each, four per word) are kernel arguments: wave-uniform
struct FoldDigits { uint32_t w[3][16]; };
HALO_DEV JacN fold_one_tab(const uint32_t *__restrict__ G, const uint32_t *__restrict__ tab, uint32_t j, uint32_t m, uint32_t lo, uint32_t cnt,
                           const FoldDigits &dg) {
    JacN acc = jac_from_aff(aff_load(G + AFF_STRIDE * (size_t)j));
#pragma unroll 1
    for (int word = 0; word < 16; word++) {
        uint32_t d1 = 0, d2 = 0, d3 = 0;
#pragma unroll
        for (int q = 0; q < 16; q++) {  // (no runtime-indexed argument array: a select chain over scalar registers)
            d1 = (q == word) ? dg.w[0][q] : d1;
            d2 = (q == word) ? dg.w[1][q] : d2;
            d3 = (q == word) ? dg.w[2][q] : d3;
        }
#pragma unroll 1
        for (int k = 0; k < 4; k++) {
            int win = word * 4 + k;
            auto step = [&](uint32_t packed, uint32_t t) {
                int d = (int)(int8_t)((packed >> (8 * k)) & 0xffu);
                if (d == 0) return;  // wave-uniform
                uint32_t mag = (uint32_t)(d < 0 ? -d : d);
                AffN e = ft_load(tab + ((size_t)(win * FT_MULT + (int)mag - 1) * cnt + (j + t * m - lo)) * FT_WORDS);
                acc = jac_madd(acc, aff_cneg(e, d < 0));
            };
            step(d1, 1);
            step(d2, 2);
            step(d3, 3);
        }
    }
    return acc;
}
// as k_fold_points4: a lane folds j and j + half and shares one inversion; src (the key) and dst are different arrays
__global__ __launch_bounds__(256, 2) void k_fold_tab4(const uint32_t *__restrict__ G, const uint32_t *__restrict__ tab, uint32_t *__restrict__ out,
                                                      uint32_t m, uint32_t half, uint32_t lo, uint32_t cnt, FoldDigits dg) {
    uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= half) return;
    bool two = j + half < m;
    JacN p[2];
    p[0] = fold_one_tab(G, tab, j, m, lo, cnt, dg);
    p[1] = two ? fold_one_tab(G, tab, j + half, m, lo, cnt, dg) : jac_inf();
    AffN a[2];
    jac_batch_to_aff(p, a);
    aff_store(out + AFF_STRIDE * (size_t)j, a[0]);
    if (two) aff_store(out + AFF_STRIDE * (size_t)(j + half), a[1]);
}

// ---- host side
// signed base-16 digits of a scalar: 64 digits in [-8, 8] (the value is below 2^255: the top nibble is at most 7, + carry 8)
static void signed_digits16(const host::Fr &s_mont, int8_t out[64]) {
    host::Fr c = s_mont.from_mont();
    int carry = 0;
    for (int i = 0; i < 64; ++i) {
        int v = (int)((c.l[i / 16] >> (4 * (i % 16))) & 15u) + carry;
        if (v > 8) { v -= 16; carry = 1; } else carry = 0;
        out[i] = (int8_t)v;
    }
}

void foldtab_release(halo_ctx *ctx) {
    if (!ctx->d_foldtab) return;
    alloc_epoch_bump(ctx);
    (void)hipFree(ctx->d_foldtab);
    ctx->d_foldtab = nullptr;
    ctx->foldtab_bytes = 0;
}

// the table over [n/4, n) of the context's key, built in slices through a temporary of at most ~4 GiB
static int foldtab_build(halo_ctx *ctx) {
    const size_t N = ctx->n, lo = N / 4, cnt = N - lo;
    const size_t bytes = (size_t)FT_ENTRIES * cnt * FT_WORDS * 4;
    auto t0 = std::chrono::steady_clock::now();
    uint32_t *tab = nullptr, *tmp = nullptr;
    size_t slice = cnt < ((size_t)1 << 15) ? cnt : ((size_t)1 << 15);  // 32768 points x 512 entries x 200 B = 3.4 GB of temporaries
    hipError_t e = getenv("HALO_TEST_TABLE_FAIL") ? hipErrorOutOfMemory : hipMalloc(&tab, bytes);
    if (e == hipSuccess) e = hipMalloc(&tmp, (size_t)FT_ENTRIES * slice * FT_TMP_WORDS * 4);
    for (size_t off = 0; off < cnt && e == hipSuccess; off += slice) {
        size_t count = cnt - off < slice ? cnt - off : slice;
        HALO_LAUNCH(ctx, "k_foldtab_build", k_foldtab_build, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, ctx->d_bases, (uint32_t)(lo + off),
                    (uint32_t)count, (uint32_t)lo, (uint32_t)cnt, tmp, tab);
        e = hipGetLastError();
    }
    hipError_t e2 = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = e2;
    if (tmp) (void)hipFree(tmp);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        if (tab) (void)hipFree(tab);
        ctx->fold_table_mode = 0;  // this context carries on with the generic fold and does not try again
        fprintf(stderr, "[halo] fold table of %zu bytes not built (%s): this context continues without it\n", bytes, hipGetErrorString(e));
        return HALO_OK;
    }
    alloc_epoch_bump(ctx);
    ctx->d_foldtab = tab;
    ctx->foldtab_bytes = bytes;
    ctx->foldtab_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (debug_trace()) fprintf(stderr, "[halo] fold table ctx=%p [%p, +%zu) built in %.1f ms\n", (void *)ctx, (void *)tab, bytes, ctx->foldtab_build_ms);
    return HALO_OK;
}

// Called by the IPA at the two-level fold from the context's own key (m = n / 4 outputs).  Returns 1 if the table kernel
// ran, 0 if the caller should take the generic kernel, < 0 on a launch error.
int fold_points4_tab(halo_ctx *ctx, const uint32_t *d_src, uint32_t *d_dst, size_t m, const host::Fr s[3]) {
    if (d_src != ctx->d_bases || 4 * m != ctx->n || m < 16 || d_dst == d_src || ctx->fold_table_mode == 0) return 0;
    if (!ctx->d_foldtab) {
        // mode 1: at the first full-size open; default: at the second (a context that opens once never pays the build)
        ctx->foldtab_opens++;
        bool now = ctx->fold_table_mode == 1 || (ctx->fold_table_mode < 0 && ctx->n >= ((size_t)1 << 18) && ctx->foldtab_opens >= 2);
        if (!now) return 0;
        int rc = foldtab_build(ctx);
        if (rc) return rc;
        if (!ctx->d_foldtab) return 0;
    }
    FoldDigits dg;
    for (int t = 0; t < 3; ++t) {
        int8_t d[64];
        signed_digits16(s[t], d);
        for (int q = 0; q < 16; ++q)
            dg.w[t][q] = (uint32_t)(uint8_t)d[4 * q] | ((uint32_t)(uint8_t)d[4 * q + 1] << 8) | ((uint32_t)(uint8_t)d[4 * q + 2] << 16) |
                         ((uint32_t)(uint8_t)d[4 * q + 3] << 24);
    }
    size_t half = m >= 512 ? (m + 1) / 2 : m;
    size_t lo = ctx->n / 4, cnt = ctx->n - lo;
    HALO_LAUNCH(ctx, "k_fold_points4_tab", k_fold_tab4, dim3((unsigned)((half + 255) / 256)), dim3(256), 0, d_src, ctx->d_foldtab, d_dst, (uint32_t)m,
                (uint32_t)half, (uint32_t)lo, (uint32_t)cnt, dg);
    HALO_HIP(hipGetLastError());
    return 1;
}

}  // namespace halo
