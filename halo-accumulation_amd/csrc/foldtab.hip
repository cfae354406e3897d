// Fixed-base comb tables for the FIRST fold of pcdl::open (pcdl.rs:216-219 applied twice, ipa.hip k_fold_points4):
//     G''[j] = G[j] + s1 G[j+m] + s2 G[j+2m] + s3 G[j+3m],   m = n / 4,
// where G is the context's own key -- a constant (consts.rs:68), like the bases of the MSM tables of msm.hip.  The generic
// kernel multiplies each of the three points by its scalar with a shared 128-step doubling chain (Straus over GLV digit
// strings: ~128 doublings + ~213 mixed additions, ~3240 field products per output).  With
//     E[w][d][i] = d * 64^w * G_i   (w < 22, d = 1..32, i in [n/4, n), affine, 64 bytes each, 44 KiB per point)
// and the endomorphism (s = k1 + k2 lambda with |k1|, |k2| < 2^128, lambda (x, y) = (beta x, y): host_math.hpp glv_decompose)
// a scalar multiple is 2 x 22 table entries added up (signed base-64 digits of k1 and k2; the k2 entries are summed as they
// are and lambda is applied ONCE to their sum), no doublings at all: ~128 mixed additions per output, ~1350 products -- the
// pass over 2^18 outputs goes from 5.4 to 2.45 ms.  The digits depend on the challenges only, i.e. they are the same for every lane of a wave:
// entry (w, d) of consecutive points is read by consecutive lanes, so the layout [w][d][i] makes every gather a fully
// coalesced 4 KiB read (2.1 GB per pass at n = 2^20).
//
// Cost: 35.4 GB at n = 2^20 (of 288 GB) and ~0.2 s to build (704 group operations and one share of an inversion per table
// entry) -- seventy opens' worth of savings.  Mode -1 (default; halo_set_fold_table: 1 = allocated and built at the first
// full-size open, 0 = never): the first full-size open of a context of 2^18 .. 2^21 points asks for the memory on a helper
// thread (hipMalloc of 40 GB takes 0.5 ms .. 2 s) and takes the generic kernel; the first later open that finds the memory
// there builds the table -- a prover chain (acc.rs:190-228: two opens per step) gets it at once, a single open does not wait for
// the table.  (It may wait for the driver: on a box whose memory this process has not had before, the 40 GB take ~1 s inside
// hipMalloc and the HIP calls of every other thread of the process queue up behind it -- measured: the first open of a fresh
// box 1.0 s instead of 39 ms, once; tools/time_acc.py names its slowest steps.  In line it would be the same second.)
// The memory is OPTIONAL memory: it is reserved against the device's budget (halo_set_memory_budget, default 1/6 of the
// device) before it is requested, and never more than half of what is free is taken.  No budget, no memory, no table: the
// generic kernel gives the same points, and the table is tried again later.
#include "curve.hpp"
#include "internal.hpp"

namespace halo {

constexpr int FT_WINDOWS = 22, FT_BITS = 6, FT_MULT = 32, FT_ENTRIES = FT_WINDOWS * FT_MULT;  // 22 x 6 = 132 bits >= the 129 of a GLV half
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

// ---- build: lane s of a slice takes point i = first + s.  Forward: the 704 multiples in XYZZ form, window by window
// (d P = 2 ((d / 2) P) for even d, (d - 1) P + P for odd d; the next window starts at 2 (32 P) = 64 P), each written to
// tmp[e][s] next to the running product of the ZZZ's.  One inversion per point.  Backward: 1 / ZZZ_e from the running
// products, x = X ZZ^2 / ZZZ^2 (ZZ^3 = ZZZ^2), y = Y / ZZZ, packed into E[e][i - lo].
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
        XyzzN t = P1;
        emit(e0, t);
#pragma unroll 1
        for (int d = 2; d <= FT_MULT; d++) {
            if (d & 1) xyzz_add(t, P1);                          // (d - 1) P + P
            else t = xyzz_dbl(xyzz_load(slot(e0 + d / 2 - 1)));  // 2 ((d / 2) P)
            emit(e0 + d - 1, t);
        }
        P1 = xyzz_dbl(t);  // 64 P: the next window's unit
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

// ---- the fold: six digit strings (k1, k2 of the three scalars; signed base 64, one byte per digit, four per word) are
// kernel arguments: wave-uniform
struct FoldDigits { uint32_t w[6][6]; };
// `park` (36 words per thread, word k of thread t at park[256 k + t]) holds the plain accumulator while the lambda chain runs
HALO_DEV JacN fold_one_tab(const uint32_t *__restrict__ G, const uint32_t *__restrict__ tab, uint32_t j, uint32_t m, uint32_t lo, uint32_t cnt,
                           const FoldDigits &dg, uint32_t *park) {
    constexpr uint32_t BETA[9] = {0x1342a796, 0x3fdac51, 0x54dab11, 0x5b221a6, 0xccd27ac, 0x15cc87a4, 0x1b1533b6, 0x169e85e1, 0x3b0093};
    Fq<1> beta;
#pragma unroll
    for (int i = 0; i < 9; i++) beta.v[i] = BETA[i];
    // XYZZ accumulators (mixed addition 8M + 2S against 7M + 4S in Jacobian form).  The entries of the three plain
    // half-scalars are summed first, then those of the three lambda half-scalars AS THEY ARE into a second accumulator;
    // lambda is applied once at the end, lambda (X, Y, ZZ, ZZZ) = (beta X, Y, ZZ, ZZZ): one product per output instead of
    // one per addition.  Every branch depends only on kernel arguments (the digits), so a wave never diverges.
    auto chain = [&](XyzzN &a, int lambda_half) {
#pragma unroll 1
        for (int word = 0; word < 6; word++) {
            uint32_t d[3] = {0, 0, 0};
#pragma unroll
            for (int q = 0; q < 6; q++) {  // (no runtime-indexed argument array: a select chain over scalar registers)
#pragma unroll
                for (int h = 0; h < 3; h++) d[h] = (q == word) ? (lambda_half ? dg.w[2 * h + 1][q] : dg.w[2 * h][q]) : d[h];
            }
#pragma unroll 1
            for (int k = 0; k < 4; k++) {
                int win = word * 4 + k;
                if (win >= FT_WINDOWS) break;
#pragma unroll 1
                for (uint32_t t = 1; t <= 3; t++) {
                    uint32_t packed = t == 1 ? d[0] : (t == 2 ? d[1] : d[2]);
                    int dv = (int)(int8_t)((packed >> (8 * k)) & 0xffu);
                    if (dv == 0) continue;  // wave-uniform
                    uint32_t mag = (uint32_t)(dv < 0 ? -dv : dv);
                    AffN e = ft_load(tab + ((size_t)(win * FT_MULT + (int)mag - 1) * cnt + (j + t * m - lo)) * FT_WORDS);
                    xyzz_madd(a, aff_cneg(e, dv < 0));
                }
            }
        }
    };
    uint32_t *mine = park + threadIdx.x;
    {
        XyzzN acc = xyzz_from_aff(aff_load(G + AFF_STRIDE * (size_t)j));
        chain(acc, 0);
#pragma unroll
        for (int k = 0; k < 9; k++) {
            mine[256 * k] = acc.x.v[k]; mine[256 * (9 + k)] = acc.y.v[k]; mine[256 * (18 + k)] = acc.zz.v[k]; mine[256 * (27 + k)] = acc.zzz.v[k];
        }
    }
    XyzzN accl = xyzz_inf();
    chain(accl, 1);
    if (!xyzz_is_inf(accl)) accl.x = fq_widen<8>(fq_mul(accl.x, beta));
    XyzzN acc;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        acc.x.v[k] = mine[256 * k]; acc.y.v[k] = mine[256 * (9 + k)]; acc.zz.v[k] = mine[256 * (18 + k)]; acc.zzz.v[k] = mine[256 * (27 + k)];
    }
    xyzz_add(acc, accl);
    return xyzz_to_jac(acc);
}
// as k_fold_points4: a lane folds j and j + half and shares one inversion; src (the key) and dst are different arrays
__global__ __launch_bounds__(256, 2) void k_fold_tab4(const uint32_t *__restrict__ G, const uint32_t *__restrict__ tab, uint32_t *__restrict__ out,
                                                      uint32_t m, uint32_t half, uint32_t lo, uint32_t cnt, FoldDigits dg) {
    uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= half) return;
    bool two = j + half < m;
    // the first result waits in LDS while the second chain runs (two XYZZ accumulators + an entry + a mixed addition's
    // temporaries fill the 256 registers that two waves per SIMD allow); word k of thread t at park[256 k + t]
    __shared__ uint32_t park[27 * 256], park_acc[36 * 256];
    JacN p[2];
    {
        JacN r0 = fold_one_tab(G, tab, j, m, lo, cnt, dg, park_acc);
        uint32_t *mine = park + threadIdx.x;
#pragma unroll
        for (int k = 0; k < 9; k++) { mine[256 * k] = r0.x.v[k]; mine[256 * (9 + k)] = r0.y.v[k]; mine[256 * (18 + k)] = r0.z.v[k]; }
    }
    p[1] = two ? fold_one_tab(G, tab, j + half, m, lo, cnt, dg, park_acc) : jac_inf();
    {
        const uint32_t *mine = park + threadIdx.x;
#pragma unroll
        for (int k = 0; k < 9; k++) { p[0].x.v[k] = mine[256 * k]; p[0].y.v[k] = mine[256 * (9 + k)]; p[0].z.v[k] = mine[256 * (18 + k)]; }
    }
    AffN a[2];
    jac_batch_to_aff(p, a);
    aff_store(out + AFF_STRIDE * (size_t)j, a[0]);
    if (two) aff_store(out + AFF_STRIDE * (size_t)(j + half), a[1]);
}

// ---- host side
// signed base-64 digits of a GLV half (256-bit two's complement, |k| < 2^129): 22 digits in [-32, 32], sign of k folded in
static void signed_digits64(const uint64_t k[4], int8_t out[FT_WINDOWS]) {
    uint64_t mag[4] = {k[0], k[1], k[2], k[3]};
    bool neg = (k[3] >> 63) != 0;
    if (neg) {  // two's complement negation
        uint64_t carry = 1;
        for (int i = 0; i < 4; ++i) { mag[i] = ~mag[i] + carry; carry = (carry && mag[i] == 0) ? 1 : 0; }
    }
    int carry = 0;
    for (int i = 0; i < FT_WINDOWS; ++i) {
        int bit = FT_BITS * i, w = bit >> 6, sh = bit & 63;
        uint64_t v = mag[w] >> sh;
        if (sh > 64 - FT_BITS && w + 1 < 4) v |= mag[w + 1] << (64 - sh);
        int dgt = (int)(v & ((1u << FT_BITS) - 1u)) + carry;
        if (dgt > FT_MULT) { dgt -= 1 << FT_BITS; carry = 1; } else carry = 0;
        out[i] = (int8_t)(neg ? -dgt : dgt);
    }
}

// test hook (host only): the 2 x 22 digits the fold kernel walks for scalar s
void fold_digits_host(const host::Fr &s, int8_t out[2 * FT_WINDOWS]) {
    uint64_t k1[4], k2[4];
    host::glv_decompose(s, k1, k2);
    signed_digits64(k1, out);
    signed_digits64(k2, out + FT_WINDOWS);
}

static size_t foldtab_slice(size_t cnt) { return cnt < ((size_t)1 << 15) ? cnt : ((size_t)1 << 15); }  // 32768 points x 704 entries x 200 B = 4.6 GB of temporaries
static size_t foldtab_table_bytes(size_t n) { return (size_t)FT_ENTRIES * (n - n / 4) * FT_WORDS * 4; }
static size_t foldtab_tmp_bytes(size_t n) { return (size_t)FT_ENTRIES * foldtab_slice(n - n / 4) * FT_TMP_WORDS * 4; }

// this context stops using the table; the memory goes back unless a clone (halo_ctx_clone) still uses the key
void foldtab_release(halo_ctx *ctx) {
    foldtab_cancel_alloc(ctx);
    if (!ctx->d_foldtab) return;
    alloc_epoch_bump(ctx);
    bool free_it = false;
    {
        std::lock_guard<std::mutex> lk(ctx->share->mu);
        if (ctx->share->users == 1 && ctx->share->d_foldtab == ctx->d_foldtab) { ctx->share->d_foldtab = nullptr; ctx->share->foldtab_bytes = 0; free_it = true; }
    }
    if (free_it) {
        (void)hipFree(ctx->d_foldtab);
        table_budget_release(ctx, ctx->foldtab_bytes);
    }
    ctx->d_foldtab = nullptr;
    ctx->foldtab_bytes = 0;
    ctx->foldtab_status = 0;
}
// a clone has built the table meanwhile?  Take it.  (true = this context now has the table)
static bool foldtab_adopt(halo_ctx *ctx) {
    std::lock_guard<std::mutex> lk(ctx->share->mu);
    if (!ctx->share->d_foldtab) return false;
    alloc_epoch_bump(ctx);
    ctx->d_foldtab = ctx->share->d_foldtab;
    ctx->foldtab_bytes = ctx->share->foldtab_bytes;
    ctx->foldtab_build_ms = ctx->share->foldtab_build_ms;
    ctx->foldtab_status = 2;
    return true;
}
// one context at a time requests / builds the table of a key; the others take the generic kernel until it is published
static bool foldtab_claim(halo_ctx *ctx) {
    std::lock_guard<std::mutex> lk(ctx->share->mu);
    if (ctx->share->foldtab_busy || ctx->share->d_foldtab) return false;
    ctx->share->foldtab_busy = true;
    return true;
}
static void foldtab_unclaim(halo_ctx *ctx) {
    std::lock_guard<std::mutex> lk(ctx->share->mu);
    ctx->share->foldtab_busy = false;
}

// the table over [n/4, n) of the context's key, built in slices through a temporary of at most ~4 GiB
static hipError_t foldtab_alloc(size_t n, uint32_t **tab, uint32_t **tmp) {
    hipError_t e = dev_hooks().table_fail ? hipErrorOutOfMemory : hipMalloc(tab, foldtab_table_bytes(n));
    if (e == hipSuccess) e = hipMalloc(tmp, foldtab_tmp_bytes(n));
    return e;
}
// The memory of table + temporary is RESERVED against the device's budget for optional memory before anything is allocated
// (internal.hpp table_budget_reserve: all contexts of the process on this device together stay within it, and never more than
// half of what is free at that moment is taken) and given back on every path that does not end with a table.
static bool foldtab_reserve(halo_ctx *ctx) { return table_budget_reserve(ctx, foldtab_table_bytes(ctx->n) + foldtab_tmp_bytes(ctx->n)); }
static void foldtab_unreserve(halo_ctx *ctx, bool keep_table) {
    table_budget_release(ctx, foldtab_tmp_bytes(ctx->n) + (keep_table ? 0 : foldtab_table_bytes(ctx->n)));
}
// not now: tried again after foldtab_backoff more full-size opens (8, 16, ... 1024) -- memory comes back when other contexts
// go, budgets are raised (no latch) -- and said once on stderr; halo_ctx_info(ctx, 5) carries the status
static void foldtab_later(halo_ctx *ctx, int status, const char *why) {
    ctx->foldtab_status = status;
    ctx->foldtab_retry_at = ctx->foldtab_opens + ctx->foldtab_backoff;
    if (ctx->foldtab_backoff < 1024) ctx->foldtab_backoff *= 2;
    if (!ctx->foldtab_said)
        fprintf(stderr, "[halo] fold table of %zu bytes not built (%s): the generic fold kernel runs, same proofs (halo_ctx_info 5; tried again later)\n",
                foldtab_table_bytes(ctx->n), why);
    ctx->foldtab_said = true;
}
void foldtab_cancel_alloc(halo_ctx *ctx) {
    bool had = ctx->foldtab_alloc_state.load(std::memory_order_acquire) != 0;
    if (ctx->foldtab_alloc_thread.joinable()) ctx->foldtab_alloc_thread.join();
    if (ctx->foldtab_pending) (void)hipFree(ctx->foldtab_pending);
    if (ctx->foldtab_pending_tmp) (void)hipFree(ctx->foldtab_pending_tmp);
    ctx->foldtab_pending = ctx->foldtab_pending_tmp = nullptr;
    if (had) { foldtab_unreserve(ctx, false); foldtab_unclaim(ctx); }
    ctx->foldtab_alloc_state = 0;
    if (ctx->foldtab_status == 1) ctx->foldtab_status = 0;
}
// tab / tmp: buffers the helper thread obtained, or null (allocate here).  The caller holds the reservation.
static int foldtab_build(halo_ctx *ctx, uint32_t *tab = nullptr, uint32_t *tmp = nullptr) {
    const size_t N = ctx->n, lo = N / 4, cnt = N - lo;
    const size_t bytes = foldtab_table_bytes(N);
    auto t0 = std::chrono::steady_clock::now();
    size_t slice = foldtab_slice(cnt);
    hipError_t e = tab ? hipSuccess : foldtab_alloc(N, &tab, &tmp);
    const double alloc_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    for (size_t off = 0; off < cnt && e == hipSuccess; off += slice) {
        size_t count = cnt - off < slice ? cnt - off : slice;
        HALO_LAUNCH(ctx, "k_foldtab_build", k_foldtab_build, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, ctx->d_bases, (uint32_t)(lo + off),
                    (uint32_t)count, (uint32_t)lo, (uint32_t)cnt, tmp, tab);
        e = hipGetLastError();
        if (debug_trace() && e == hipSuccess) {  // (tracing only: per-slice times, to see where a slow build spends it)
            auto ts = std::chrono::steady_clock::now();
            e = hipStreamSynchronize(ctx->stream);
            fprintf(stderr, "[halo] fold table slice at %zu: %.1f ms since the build began (this wait %.1f ms)\n", off,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(),
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ts).count());
        }
    }
    hipError_t e2 = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = e2;
    if (tmp) (void)hipFree(tmp);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        if (tab) (void)hipFree(tab);
        foldtab_unreserve(ctx, false);
        foldtab_unclaim(ctx);
        foldtab_later(ctx, 4, hipGetErrorString(e));  // this context carries on with the generic fold
        return HALO_OK;
    }
    foldtab_unreserve(ctx, true);  // the temporary is gone, the table stays on the books
    alloc_epoch_bump(ctx);
    ctx->d_foldtab = tab;
    ctx->foldtab_bytes = bytes;
    ctx->foldtab_status = 2;
    ctx->foldtab_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    {   // published: clones of this context adopt it at their next full-size open
        std::lock_guard<std::mutex> lk(ctx->share->mu);
        ctx->share->d_foldtab = tab;
        ctx->share->foldtab_bytes = bytes;
        ctx->share->foldtab_build_ms = ctx->foldtab_build_ms;
        ctx->share->foldtab_busy = false;
    }
    if (debug_trace())
        fprintf(stderr, "[halo] fold table ctx=%p [%p, +%zu) built in %.1f ms (%.1f ms of it the two allocations)\n", (void *)ctx, (void *)tab, bytes,
                ctx->foldtab_build_ms, alloc_ms);
    return HALO_OK;
}

// Called by the IPA at the two-level fold from the context's own key (m = n / 4 outputs).  Returns 1 if the table kernel
// ran, 0 if the caller should take the generic kernel, < 0 on a launch error.
int fold_points4_tab(halo_ctx *ctx, const uint32_t *d_src, uint32_t *d_dst, size_t m, const host::Fr s[3]) {
    if (d_src != ctx->d_bases || 4 * m != ctx->n || m < 16 || d_dst == d_src || ctx->fold_table_mode == 0) return 0;
    if (!ctx->d_foldtab) (void)foldtab_adopt(ctx);
    if (!ctx->d_foldtab) {
        // mode 1: at the next full-size open.  Default (-1): the table costs as much as 64 opens save (0.17 s of kernels for 2.6 ms
        // each) and 35 GB, so a key must have shown that it is opened again and again before it gets one: from its K-th full-size
        // open on (K = tuning().fold_table_after = 8, counted over all contexts of the key) the memory is requested on a helper
        // thread and the table is built at the first later open that finds it there.  A caller with a handful of opens never pays
        // for, and never reserves memory for, a table it cannot amortise; a caller that knows better says halo_set_fold_table(ctx, 1).
        ctx->foldtab_opens++;
        long key_opens;
        {
            std::lock_guard<std::mutex> lk(ctx->share->mu);
            key_opens = ++ctx->share->full_opens;
        }
        // (automatic mode also stops at 2^21 points: 71 GB; larger keys on request only)
        int rc;
        if (ctx->fold_table_mode == 1) {
            // (a request of the automatic mode may still be under way: wait for it and take what it brought)
            uint32_t *tab = nullptr, *tmp = nullptr;
            bool reserved = false;
            if (ctx->foldtab_alloc_state.load(std::memory_order_acquire) != 0) {
                if (ctx->foldtab_alloc_thread.joinable()) ctx->foldtab_alloc_thread.join();
                if (ctx->foldtab_alloc_state.load(std::memory_order_acquire) == 2) {
                    tab = ctx->foldtab_pending; tmp = ctx->foldtab_pending_tmp;
                    ctx->foldtab_pending = ctx->foldtab_pending_tmp = nullptr;
                    ctx->foldtab_alloc_state = 0;
                    reserved = true;  // (the request's reservation passes to the build)
                } else foldtab_cancel_alloc(ctx);
            }
            if (!reserved) {
                if (ctx->foldtab_opens < ctx->foldtab_retry_at) return 0;
                if (!foldtab_claim(ctx)) return 0;  // (a clone is at it: the generic kernel this time)
                if (!foldtab_reserve(ctx)) { foldtab_unclaim(ctx); foldtab_later(ctx, 3, "over the budget for optional memory, halo_set_memory_budget"); return 0; }
            }
            rc = foldtab_build(ctx, tab, tmp);
        } else {
            if (!(ctx->fold_table_mode < 0 && ctx->n >= ((size_t)1 << 18) && ctx->n <= ((size_t)1 << 21))) return 0;
            int st = ctx->foldtab_alloc_state.load(std::memory_order_acquire);
            if (st == 0) {  // the key has earned its table: ask for the memory in the background; this open takes the generic kernel
                const int K = tuning().fold_table_after;
                if (K <= 0 || key_opens < (long)K) return 0;
                if (ctx->foldtab_opens < ctx->foldtab_retry_at) return 0;
                if (!foldtab_claim(ctx)) return 0;  // (a clone is at it: the generic kernel this time)
                if (!foldtab_reserve(ctx)) { foldtab_unclaim(ctx); foldtab_later(ctx, 3, "over the budget for optional memory, halo_set_memory_budget"); return 0; }
                ctx->foldtab_alloc_state = 1;
                ctx->foldtab_status = 1;
                ctx->foldtab_alloc_thread = std::thread([ctx] {
                    (void)hipSetDevice(ctx->device);
                    hipError_t e = foldtab_alloc(ctx->n, &ctx->foldtab_pending, &ctx->foldtab_pending_tmp);
                    if (e != hipSuccess) (void)hipGetLastError();
                    ctx->foldtab_alloc_state.store(e == hipSuccess ? 2 : 3, std::memory_order_release);
                });
                return 0;
            }
            if (st == 1) return 0;  // not there yet: the generic kernel once more
            ctx->foldtab_alloc_thread.join();
            if (st == 3) {
                foldtab_cancel_alloc(ctx);  // (frees what the request got, gives the reservation back)
                foldtab_later(ctx, 4, "no memory");
                return 0;
            }
            uint32_t *tab = ctx->foldtab_pending, *tmp = ctx->foldtab_pending_tmp;
            ctx->foldtab_pending = ctx->foldtab_pending_tmp = nullptr;
            ctx->foldtab_alloc_state = 0;
            rc = foldtab_build(ctx, tab, tmp);
        }
        if (rc) return rc;
        if (!ctx->d_foldtab) return 0;
    }
    FoldDigits dg;
    for (int t = 0; t < 3; ++t) {
        uint64_t k[2][4];
        host::glv_decompose(s[t], k[0], k[1]);  // s = k1 + k2 lambda
        for (int h = 0; h < 2; ++h) {
            int8_t d[24] = {0};
            signed_digits64(k[h], d);
            for (int q = 0; q < 6; ++q)
                dg.w[2 * t + h][q] = (uint32_t)(uint8_t)d[4 * q] | ((uint32_t)(uint8_t)d[4 * q + 1] << 8) | ((uint32_t)(uint8_t)d[4 * q + 2] << 16) |
                                     ((uint32_t)(uint8_t)d[4 * q + 3] << 24);
        }
    }
    size_t half = m >= 512 ? (m + 1) / 2 : m;
    size_t lo = ctx->n / 4, cnt = ctx->n - lo;
    HALO_LAUNCH(ctx, "k_fold_points4_tab", k_fold_tab4, dim3((unsigned)((half + 255) / 256)), dim3(256), 0, d_src, ctx->d_foldtab, d_dst, (uint32_t)m,
                (uint32_t)half, (uint32_t)lo, (uint32_t)cnt, dg);
    HALO_HIP(hipGetLastError());
    return 1;
}

}  // namespace halo
