// K1/K2 Pippenger MSM on gfx950 (replaces ark-ec's msm_unchecked as called from
// group.rs:18-26), K10 batch_to_affine, K11 URS generation (main.rs:18-45), the conversions between
// the ABI's arkworks limbs and the native base-table format, and the primitive test hooks.
//
// Pipeline for n points, window c bits, W = ceil(256/c) windows, B = 2^(c-1) buckets each
// (signed digits: a point with digit d lands in bucket |d|-1 of its window, negated if d < 0):
//   k_msm_recode      scalar out of Montgomery form (arkworks `into_bigint`), signed-digit recode,
//                     u16 digits [window][i]                      (32 B in per scalar, coalesced)
//   k_msm_hist        per (window, chunk) histogram with all 2^(c-1) counters in LDS
//   k_msm_colsum      bucket sizes + per-chunk prefixes; k_scan_*: exclusive scan of the sizes
//   k_msm_scatter     point indices grouped by bucket (LDS cursors)
//   k_msm_task_bins/_task_order
//                     buckets cut into tasks of <= kmax entries, tasks sorted by decreasing length
//   k_msm_accumulate  one lane per task: XYZZ mixed adds over its index list (dominant kernel)
//   k_msm_combine_*   partials of multi-task buckets folded into the bucket value
//   k_msm_reduce1     sum_k k*B_k per segment of a window: lane-local running sums + wave64 shuffle scans
//   (k_smsm_final)    the segments of a window -> window sum (smsm.hip; quad-parallel, latency-bound)
//   k_msm_reduce_rc   the table plans' form of the two: plain sums over the rows and columns of the bucket index (one set of
//   (k_rc_mid)        buckets), weights applied to rows + columns points afterwards (smsm.hip), the last ~70 additions on the host
//   host              Horner over the W window sums (240 doublings are a 60 us job for one CPU
//                     core and a > 1 ms serial chain for one GPU lane)
// msm_enqueue / msm_finish split the launch sequence from the final wait so that independent MSMs
// overlap on the context's slots (workspace + stream each).
//
// This unit: the window plan, the general pipeline's recode and sort kernels and its launch sequence (msm_enqueue_launches).
// The bucket kernels behind the sort are msm_buckets.hip, the fixed-base-table pipeline msm_table.hip, workspaces / launch graphs /
// wait and combine msm_driver.hip, the format conversions and the URS kernels urs.hip.
#include <atomic>
#include <cstring>
#include <thread>

#include "msm_kernels.hpp"

namespace halo {

// ------------------------------------------------------------------------------ plan
MsmPlan msm_plan(size_t n, int forced_c) {
    int lg = 0;
    while (((size_t)1 << (lg + 1)) <= n) lg++;
    // Measured on gfx950 (tools/sweep_msm.py, solo latency and 4-deep pipelined throughput agree):
    // large MSMs are throughput-bound, ~32 points per bucket amortise the bucket reduction; below
    // 2^19 points the serial chains dominate.  Window sizes whose top window keeps only 2-3 scalar
    // bits (c = 14, 12, 11, 9) are avoided: that window puts n/4 points into each of ~4 buckets.
    static const int table[] = {/*lg 10*/ 8, 8, 8, 10, 10, 13, 13, 15, 15, /*lg 19*/ 15};
    int c = forced_c > 0 ? forced_c : (lg >= 20 ? 16 : lg >= 10 ? table[lg - 10] : lg - 2);
    // development override, e.g. HALO_PLAN="16:12,15:12": window bits for MSMs of 2^lg <= n < 2^(lg+1) points
    const char *plan_env = tuning().plan;
    if (plan_env && forced_c <= 0) {
        for (const char *q = plan_env; *q;) {
            int l = atoi(q);
            const char *colon = strchr(q, ':');
            if (!colon) break;
            if (l == lg) c = atoi(colon + 1);
            const char *comma = strchr(colon, ',');
            if (!comma) break;
            q = comma + 1;
        }
    }
    if (c < 4) c = 4;
    if (c > 16) c = 16;
    MsmPlan p;
    p.c = c;
    p.W = (256 + c - 1) / c;
    p.B = 1u << (c - 1);
    p.batch = 1;
    p.w0 = 0;
    p.w1 = p.W;
    return p;
}

// Top-window spread of a plan (k_msm_recode): modulus of k and the top window's first bit; 0 when the plan has fewer than
// two spare bits (c W - 255) or its top window starts at bit 254 or later (nothing but a carry lands there).
// Any s < 2^255 < 2 r and k <= mod - 1 give s + k r < (mod + 1) r <= (2^(cW-255) - 1) r < 2^(cW-1).
uint32_t msm_spread(const MsmPlan &p, uint32_t *top_bit) {
    int spare = p.c * p.W - 255;
    *top_bit = (uint32_t)(p.c * (p.W - 1));
    if (spare < 2 || *top_bit >= 254 || *top_bit < 224) return 0;
    uint32_t mod = (spare >= 6 ? 64u : (1u << spare)) - 2u;
    return mod;
}

// ------------------------------------------------------------------------------ recode

// Windows [w0, w1) are written (a window shard still walks the carry chain from window 0).
// Block (0, 0) also clears the launch's small state (meta: 256 words; zero_b: the 1024 block offsets where the sort
// writes absolute bucket starts) -- nothing reads either before the sort passes that follow.  blockIdx.y = member
// of a batched launch.  A window shard (w0 > 0) does not walk the carry chain from window 0: the carry into w0 is
// decided by the nearest lower window whose raw digit differs from B (raw < B: 0, raw > B: 1, raw == B: passes on).
// spread_mod > 0: the plan's top window holds only a few scalar bits (c = 10: 5 of 10), so its n digits would pile into a
// handful of buckets.  Every base of this library has order r (Pallas has cofactor 1), so s + k r gives the same point:
// k = i mod spread_mod makes the top digit floor((s + k r) / 2^top_bit) uniform over the window's buckets, at no cost
// (msm_spread below: s + k r < 2^(c W - 1), the top window still cannot carry out).  Scalars with an empty top window
// (zero, short challenges) and unreduced inputs >= 2^255 stay as they are.
__global__ __launch_bounds__(256) void k_msm_recode(MemberScalars scalars, int mont, uint32_t n, int c, int w0, int w1, uint32_t B,
                                                    uint32_t spread_mod, uint32_t top_bit,
                                                    uint16_t *__restrict__ digits, uint32_t *__restrict__ meta,
                                                    uint32_t *__restrict__ zero_b, uint32_t *__restrict__ zero_t) {
    __shared__ uint32_t sw[256 * 9];
    if (blockIdx.x == 0 && blockIdx.y == 0) {
        meta[threadIdx.x] = 0;
        if (zero_b)
            for (int k = 0; k < 4; k++) zero_b[threadIdx.x + 256 * k] = 0;
        if (zero_t)  // two-level sort: the fine pass writes absolute first-task ids (block offsets stay zero)
            for (int k = 0; k < 4; k++) zero_t[threadIdx.x + 256 * k] = 0;
    }
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint64_t *src = nullptr;
#pragma unroll
    for (int b = 0; b < MSM_MAX_BATCH; b++) src = ((int)blockIdx.y == b) ? scalars.p[b] : src;
    uint16_t *out = digits + (size_t)blockIdx.y * (size_t)(w1 - w0) * n;
    Fe s = fe_load(src + 4 * (size_t)i);
    if (mont) s = fe_from_mont<FrCfg>(s);  // arkworks `into_bigint`
    uint32_t *my = sw + threadIdx.x * 9;
#pragma unroll
    for (int k = 0; k < 8; k++) my[k] = s.v[k];
    my[8] = 0;
    if (spread_mod) {
        uint32_t top = my[7] >> (top_bit & 31u);  // top_bit is in word 7; bit 255 set: not below 2 r, left alone
        uint32_t k = (top != 0 && (my[7] >> 31) == 0) ? i % spread_mod : 0u;
        uint64_t acc = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            acc += (uint64_t)k * FrCfg::P[j] + my[j];
            my[j] = (uint32_t)acc;
            acc >>= 32;
        }
        my[8] = (uint32_t)acc;
    }
    uint32_t carry = 0;
    for (int j = w0 - 1; j >= 0; j--) {
        uint32_t bit = (uint32_t)j * (uint32_t)c;
        uint64_t two = (uint64_t)my[bit >> 5] | ((uint64_t)my[(bit >> 5) + 1] << 32);
        uint32_t raw = (uint32_t)(two >> (bit & 31)) & ((1u << c) - 1u);
        if (raw != B) { carry = raw > B ? 1u : 0u; break; }
    }
    for (int w = w0; w < w1; w++) {
        Digit d = next_digit(my, w, c, B, carry);
        out[(size_t)(w - w0) * n + i] = (uint16_t)(d.mag ? ((d.mag - 1) | (d.neg << 15)) : DIGIT_NONE);
    }
}

// Counting sort of one window's digits with the whole histogram in LDS (B <= 2^15 counters =
// 128 KiB of the CU's 160 KiB).  Block (w, chunk) covers scalars [chunk*len, (chunk+1)*len).
// hist layout [w][chunk][b] so that every global access is coalesced.
__global__ __launch_bounds__(1024) void k_msm_hist(const uint16_t *__restrict__ digits, uint32_t n, uint32_t B, uint32_t nchunks,
                                                   uint32_t chunk_len, int vec, uint32_t *__restrict__ hist) {
    extern __shared__ uint32_t lds[];
    uint32_t w = blockIdx.x / nchunks, chunk = blockIdx.x % nchunks;
    for (uint32_t b = threadIdx.x; b < B; b += 1024) lds[b] = 0;
    __syncthreads();
    uint32_t lo = chunk * chunk_len, hi = lo + chunk_len < n ? lo + chunk_len : n;
    const uint16_t *dg = digits + (size_t)w * n;
    if (vec) {  // n and chunk_len are multiples of 8: eight digits per 16-byte load, eight atomics in flight
        for (uint32_t i = lo + 8 * threadIdx.x; i < hi; i += 8 * 1024) {
            uint4 q = *reinterpret_cast<const uint4 *>(dg + i);
            uint32_t v[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                uint32_t d0 = v[k] & 0xFFFFu, d1 = v[k] >> 16;
                if (d0 != DIGIT_NONE) atomicAdd(&lds[d0 & 0x7FFFu], 1u);
                if (d1 != DIGIT_NONE) atomicAdd(&lds[d1 & 0x7FFFu], 1u);
            }
        }
    } else {
        for (uint32_t i = lo + threadIdx.x; i < hi; i += 1024) {
            uint32_t d = dg[i];
            if (d != DIGIT_NONE) atomicAdd(&lds[d & 0x7FFFu], 1u);
        }
    }
    __syncthreads();
    uint32_t *out = hist + ((size_t)w * nchunks + chunk) * B;
    for (uint32_t b = threadIdx.x; b < B; b += 1024) out[b] = lds[b];
}
// per bucket: total over chunks -> counts[g]; hist[w][chunk][b] <- exclusive prefix over chunks
__global__ __launch_bounds__(256) void k_msm_colsum(uint32_t *__restrict__ hist, uint32_t B, uint32_t nchunks, uint32_t total, uint32_t kmax,
                                                    uint32_t *__restrict__ counts, uint32_t *__restrict__ ntask) {
    uint32_t g = blockIdx.x * 256 + threadIdx.x;
    if (g >= total) return;
    uint32_t w = g / B, b = g % B;
    uint32_t run = 0;
    for (uint32_t ch = 0; ch < nchunks; ch++) {
        uint32_t *p = hist + ((size_t)w * nchunks + ch) * B + b;
        uint32_t t = *p;
        *p = run;
        run += t;
    }
    counts[g] = run;
    ntask[g] = (run + kmax - 1) / kmax;  // tasks of at most kmax entries (see the accumulate section)
}
__global__ __launch_bounds__(1024) void k_msm_scatter(const uint16_t *__restrict__ digits, uint32_t n, uint32_t B, uint32_t nchunks,
                                                      uint32_t chunk_len, const uint32_t *__restrict__ hist,
                                                      const uint32_t *__restrict__ starts, const uint32_t *__restrict__ blockoff,
                                                      uint32_t W_member, MemberOffsets offs, int vec, uint32_t *__restrict__ sorted) {
    extern __shared__ uint32_t lds[];
    uint32_t w = blockIdx.x / nchunks, chunk = blockIdx.x % nchunks;
    uint32_t off = offs.v[w / W_member];  // this window's member reads its bases from point index `off` on
    const uint32_t *pre = hist + ((size_t)w * nchunks + chunk) * B;
    for (uint32_t b = threadIdx.x; b < B; b += 1024) {
        uint32_t g = w * B + b;
        lds[b] = starts[g] + blockoff[g >> 12] + pre[b];
    }
    __syncthreads();
    uint32_t lo = chunk * chunk_len, hi = lo + chunk_len < n ? lo + chunk_len : n;
    const uint16_t *dg = digits + (size_t)w * n;
    if (vec) {
        for (uint32_t i = lo + 8 * threadIdx.x; i < hi; i += 8 * 1024) {
            uint4 q = *reinterpret_cast<const uint4 *>(dg + i);
            uint32_t v[4] = {q.x, q.y, q.z, q.w};
            uint32_t pos[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                uint32_t d = (v[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
                pos[k] = d != DIGIT_NONE ? atomicAdd(&lds[d & 0x7FFFu], 1u) : 0xFFFFFFFFu;
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                uint32_t d = (v[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
                if (pos[k] != 0xFFFFFFFFu) sorted[pos[k]] = (i + k + off) | ((d >> 15) << 31);
            }
        }
    } else {
        for (uint32_t i = lo + threadIdx.x; i < hi; i += 1024) {
            uint32_t d = dg[i];
            if (d != DIGIT_NONE) {
                uint32_t pos = atomicAdd(&lds[d & 0x7FFFu], 1u);
                sorted[pos] = (i + off) | ((d >> 15) << 31);
            }
        }
    }
}


// ------------------------------------------------------------------------------ two-level sort (large MSMs)
// k_msm_scatter writes every entry to its final place: 16.8 M isolated 4-byte stores at n = 2^20, each of
// which costs a 32-byte HBM transaction (WRITE_SIZE 525 MB for 67 MB of payload).  For large MSMs the sort
// is split so that every store lands next to recent ones:
//   1. coarse: the window's buckets are cut into NC = B / 2^F ranges; block (window, chunk) appends its
//      entries to NC runs, each advancing sequentially (the open lines stay in L2 until they are full);
//   2. fine: block (window, range) reads its run (about n / NC entries, contiguous), counts its 2^F buckets
//      in LDS, and places the entries inside its own region of the output (~128 KiB: L2-resident).
// The fine pass produces the bucket counts and absolute start offsets as a by-product (no global scan).
#ifndef HALO_FINE_BITS
#define HALO_FINE_BITS 10
#endif
constexpr int FINE_BITS = HALO_FINE_BITS;
constexpr uint32_t NC_MAX = 32768u >> FINE_BITS;  // bucket ranges per window at c = 16

// coarse histogram: chist[(w * nchunks + chunk) * NC + c]; one private row of counters per wave
__global__ __launch_bounds__(1024) void k_msm_coarse_hist(const uint16_t *__restrict__ digits, uint32_t n, uint32_t NC, uint32_t nchunks,
                                                          uint32_t chunk_len, uint32_t *__restrict__ chist) {
    __shared__ uint32_t cnt[16 * NC_MAX];
    uint32_t w = blockIdx.x / nchunks, chunk = blockIdx.x % nchunks;
    for (uint32_t k = threadIdx.x; k < 16 * NC_MAX; k += 1024) cnt[k] = 0;
    __syncthreads();
    uint32_t *mine = cnt + NC_MAX * (threadIdx.x >> 6);
    uint32_t lo = chunk * chunk_len, hi = lo + chunk_len < n ? lo + chunk_len : n;
    const uint16_t *dg = digits + (size_t)w * n;
    for (uint32_t i = lo + 8 * threadIdx.x; i < hi; i += 8 * 1024) {  // n and chunk_len are multiples of 8
        uint4 q = *reinterpret_cast<const uint4 *>(dg + i);
        uint32_t v[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 8; k++) {
            uint32_t d = (v[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
            if (d != DIGIT_NONE) atomicAdd(&mine[(d & 0x7FFFu) >> FINE_BITS], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < NC) {
        uint32_t t = 0;
        for (int r = 0; r < 16; r++) t += cnt[NC_MAX * r + threadIdx.x];
        chist[((size_t)w * nchunks + chunk) * NC + threadIdx.x] = t;
    }
}
// one block: chist <- exclusive prefix over the chunks of each (window, range); cstart[p] = start of run p = w * NC + c
// in the presorted array, cstart[P] = number of entries.  P <= 4096.
__global__ __launch_bounds__(1024) void k_msm_coarse_scan(uint32_t *__restrict__ chist, uint32_t P, uint32_t NC, uint32_t nchunks,
                                                          uint32_t *__restrict__ cstart) {
    __shared__ uint32_t part[1024];
    uint32_t tot[4];
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t p = threadIdx.x * 4 + k, run = 0;
        if (p < P) {
            uint32_t w = p / NC, c = p % NC;
            for (uint32_t ch = 0; ch < nchunks; ch++) {
                uint32_t *q = chist + ((size_t)w * nchunks + ch) * NC + c;
                uint32_t t = *q;
                *q = run;
                run += t;
            }
        }
        tot[k] = sum;
        sum += run;
    }
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        uint32_t v = (threadIdx.x >= (uint32_t)off) ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t excl = part[threadIdx.x] - sum;
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (threadIdx.x * 4 + k < P) cstart[threadIdx.x * 4 + k] = excl + tot[k];
    if (threadIdx.x == 1023) cstart[P] = part[1023];
}
// coarse scatter: presort[run position] = i | sign << 31 (plain index; the member's base offset is added by the fine pass)
__global__ __launch_bounds__(1024) void k_msm_coarse_scatter(const uint16_t *__restrict__ digits, uint32_t n, uint32_t NC, uint32_t nchunks,
                                                             uint32_t chunk_len, const uint32_t *__restrict__ chist,
                                                             const uint32_t *__restrict__ cstart, int packed, uint32_t *__restrict__ presort) {
    __shared__ uint32_t cur[NC_MAX];
    uint32_t w = blockIdx.x / nchunks, chunk = blockIdx.x % nchunks;
    if (threadIdx.x < NC) cur[threadIdx.x] = cstart[w * NC + threadIdx.x] + chist[((size_t)w * nchunks + chunk) * NC + threadIdx.x];
    __syncthreads();
    uint32_t lo = chunk * chunk_len, hi = lo + chunk_len < n ? lo + chunk_len : n;
    const uint16_t *dg = digits + (size_t)w * n;
    for (uint32_t i = lo + 8 * threadIdx.x; i < hi; i += 8 * 1024) {
        uint4 q = *reinterpret_cast<const uint4 *>(dg + i);
        uint32_t v[4] = {q.x, q.y, q.z, q.w};
        uint32_t pos[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            uint32_t d = (v[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
            pos[k] = d != DIGIT_NONE ? atomicAdd(&cur[(d & 0x7FFFu) >> FINE_BITS], 1u) : 0xFFFFFFFFu;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            uint32_t d = (v[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
            // packed (n <= 2^21): the bucket's low bits ride along in bits 21..30, the fine pass needs no digit lookup
            if (pos[k] != 0xFFFFFFFFu) presort[pos[k]] = (i + k) | ((d >> 15) << 31) | (packed ? (d & ((1u << FINE_BITS) - 1u)) << 21 : 0u);
        }
    }
}
// fine sort of run p = (window, range): counts / absolute starts of its 2^F buckets, entries placed in [lo, hi) of `sorted`
template <bool PACKED>
__global__ __launch_bounds__(1024) void k_msm_fine_sort(const uint32_t *__restrict__ presort, const uint16_t *__restrict__ digits, uint32_t n,
                                                        uint32_t B, uint32_t NC, const uint32_t *__restrict__ cstart, uint32_t W_member,
                                                        MemberOffsets offs, uint32_t kmax, uint32_t total_buckets, uint32_t *__restrict__ counts,
                                                        uint32_t *__restrict__ starts, uint32_t *__restrict__ ntask, uint32_t *__restrict__ toff,
                                                        uint32_t *__restrict__ task_g, uint32_t *__restrict__ biglist, uint32_t *__restrict__ meta,
                                                        uint32_t *__restrict__ sorted) {
    // Besides the sort: the task lists (a bucket of c entries = ceil(c / kmax) tasks with consecutive ids reserved with one
    // atomic per block, toff = absolute first id, task_g, tasks-per-length counts for k_msm_task_order in meta[2 ..], the
    // multi-task buckets for k_msm_combine) -- what k_scan_blocks/_top + k_msm_task_bins do for the one-level sort.
    __shared__ uint32_t hist[1 << FINE_BITS], scan[1 << FINE_BITS], tscan[1 << FINE_BITS], lbin[KMAX + 8], misc[2];
    constexpr uint32_t FMASK = (1u << FINE_BITS) - 1u, IMASK = PACKED ? 0x1FFFFFu : 0x7FFFFFFFu;
    uint32_t p = blockIdx.x, w = p / NC, c = p % NC;
    uint32_t lo = cstart[p], hi = cstart[p + 1];
    uint32_t off = offs.v[w / W_member];
    const uint16_t *dg = digits + (size_t)w * n;
    auto fine_of = [&](uint32_t v) -> uint32_t { return PACKED ? (v >> 21) & FMASK : (uint32_t)dg[v & IMASK] & FMASK; };
    constexpr uint32_t NB = 1u << FINE_BITS;  // buckets of this block (<= 1024 threads: one bucket per thread at most)
    bool owner = threadIdx.x < NB;
    if (owner) hist[threadIdx.x] = 0;
    if (threadIdx.x < KMAX + 8) lbin[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t e = lo + threadIdx.x; e < hi; e += 4 * 1024) {  // four independent entries per lane per trip
        uint32_t v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = e + k * 1024 < hi ? presort[e + k * 1024] : 0xFFFFFFFFu;
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (e + k * 1024 < hi) atomicAdd(&hist[fine_of(v[k])], 1u);
    }
    __syncthreads();
    uint32_t mine = owner ? hist[threadIdx.x] : 0u, nt = (mine + kmax - 1) / kmax;
    if (owner) { scan[threadIdx.x] = mine; tscan[threadIdx.x] = nt; }
    if (nt) {  // lengths of this bucket's tasks: kmax for all but the last
        if (nt > 1) atomicAdd(&lbin[KMAX - kmax], nt - 1);
        atomicAdd(&lbin[KMAX - (mine - (nt - 1) * kmax)], 1u);
    }
    __syncthreads();
    for (uint32_t o = 1; o < NB; o <<= 1) {
        uint32_t t = (owner && threadIdx.x >= o) ? scan[threadIdx.x - o] : 0u, t2 = (owner && threadIdx.x >= o) ? tscan[threadIdx.x - o] : 0u;
        __syncthreads();
        if (owner) { scan[threadIdx.x] += t; tscan[threadIdx.x] += t2; }
        __syncthreads();
    }
    if (threadIdx.x == NB - 1) misc[0] = atomicAdd(&meta[0], tscan[NB - 1]);  // this block's task ids: [base, base + total)
    if (threadIdx.x <= KMAX && lbin[threadIdx.x]) atomicAdd(&meta[2 + threadIdx.x], lbin[threadIdx.x]);
    __syncthreads();
    if (owner) {
        uint32_t begin = lo + scan[threadIdx.x] - mine, tfirst = misc[0] + tscan[threadIdx.x] - nt;
        uint32_t g = w * B + (c << FINE_BITS) + threadIdx.x;
        counts[g] = mine;
        ntask[g] = nt;
        starts[g] = begin;  // absolute: the block offsets of the two-level scan format are zeroed by the recode kernel
        toff[g] = tfirst;   // likewise
        for (uint32_t j = 0; j < nt; j++) task_g[tfirst + j] = g | ((KMAX - (j + 1 < nt ? kmax : mine - (nt - 1) * kmax)) << 24);
        hist[threadIdx.x] = begin;  // now the bucket's write cursor
    }
    // multi-task buckets for k_msm_combine: counted in LDS, one reservation per block and list
    uint32_t big_rank = 0, small_rank = 0;
    if (owner && nt > 8) big_rank = atomicAdd(&lbin[KMAX + 1], 1u);
    else if (owner && nt > 1) small_rank = atomicAdd(&lbin[KMAX + 2], 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
        lbin[KMAX + 3] = lbin[KMAX + 1] ? atomicAdd(&meta[1], lbin[KMAX + 1]) : 0u;
        lbin[KMAX + 4] = lbin[KMAX + 2] ? atomicAdd(&meta[140], lbin[KMAX + 2]) : 0u;
    }
    __syncthreads();
    {
        uint32_t g = w * B + (c << FINE_BITS) + threadIdx.x;
        if (owner && nt > 8) biglist[lbin[KMAX + 3] + big_rank] = g;
        else if (owner && nt > 1) biglist[total_buckets - 1 - (lbin[KMAX + 4] + small_rank)] = g;
    }
    extern __shared__ uint32_t stage[];  // FINE_STAGE entries: the block's whole output region when it fits
    bool staged = hi - lo <= FINE_STAGE;
    for (uint32_t e = lo + threadIdx.x; e < hi; e += 4 * 1024) {
        uint32_t v[4], pos[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = e + k * 1024 < hi ? presort[e + k * 1024] : 0xFFFFFFFFu;
#pragma unroll
        for (int k = 0; k < 4; k++) pos[k] = e + k * 1024 < hi ? atomicAdd(&hist[fine_of(v[k])], 1u) : 0u;
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (e + k * 1024 < hi) {
                uint32_t out = ((v[k] & IMASK) + off) | (v[k] & 0x80000000u);
                if (staged) stage[pos[k] - lo] = out;  // random within LDS ...
                else sorted[pos[k]] = out;             // a run longer than the staging area (skewed scalars): placed directly
            }
    }
    if (staged) {
        __syncthreads();
        for (uint32_t e = lo + threadIdx.x; e < hi; e += 1024) sorted[e] = stage[e - lo];  // ... sequential to HBM
    }
}

int msm_general_prepare() {
    HALO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_msm_hist), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    HALO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_msm_scatter), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    HALO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_msm_fine_sort<true>), hipFuncAttributeMaxDynamicSharedMemorySize, FINE_STAGE * 4));
    HALO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_msm_fine_sort<false>), hipFuncAttributeMaxDynamicSharedMemorySize, FINE_STAGE * 4));
    return HALO_OK;
}

// the launch sequence proper (recorded into a graph when the stream is capturing); sets ws.plan.
// Wt = W * batch windows go through the sort / accumulate / reduce kernels as if they belonged to one MSM;
// only the recode (one scalar array per member) and the scatter (one base offset per member) know better.
int msm_enqueue_launches(halo_ctx *ctx, MsmWorkspace &ws, const uint32_t *d_bases, const MsmBatch &members, bool mont, size_t n, int partner) {
    if (n > ws.cap_n) { set_error("msm: n exceeds the context's workspace"); return HALO_E_ARG; }
    if (ctx->d_table && table_eligible(ctx, d_bases, members, n)) return tmsm_enqueue_launches(ctx, ws, d_bases, members, mont, n, partner);
    MsmPlan p = msm_plan(n, launch_c(ctx, members));
    p.batch = members.count;
    p.w0 = p.W * members.part / members.parts;
    p.w1 = p.W * (members.part + 1) / members.parts;
    uint32_t Wm = (uint32_t)(p.w1 - p.w0);  // windows per member in this launch
    uint32_t Wt = Wm * (uint32_t)p.batch;
    size_t total = (size_t)Wt * p.B;
    if (total > ws.cap_counts || n * (size_t)Wt > ws.cap_sorted || Wt > ws.cap_windows) {
        set_error("msm: window plan exceeds workspace");
        return HALO_E_ARG;
    }
    hipStream_t s = ctx->stream;
    dim3 gridn((unsigned)((n + 255) / 256)), b256(256);
    uint16_t *d_digits = reinterpret_cast<uint16_t *>(ws.d_canon);  // n * Wt * 2 bytes, layout [member][w][i]
    MemberOffsets offs{};
    MemberScalars srcs{};
    for (int b = 0; b < p.batch; ++b) {
        offs.v[b] = members.base_off[b];
        srcs.p[b] = members.scalars[b];
    }
    uint32_t top_bit = 0, spread_mod = msm_spread(p, &top_bit);
    HALO_LAUNCH(ctx, "k_msm_recode", k_msm_recode, dim3(gridn.x, (unsigned)p.batch), b256, 0, srcs, mont ? 1 : 0, (uint32_t)n, p.c, p.w0, p.w1, p.B,
                spread_mod, top_bit, d_digits, ws.d_meta, ws.d_blockoff, ws.d_tblockoff);
    // chain bound per lane of the bucket kernel: 64 where the launch is throughput-bound, 16 where it is latency-bound
    uint32_t kmax = msm_kmax(ctx, n);
    bool same_bases = true;  // the small pipeline takes one base offset: members over the same points (the L and R of an IPA round)
    for (int b = 1; b < p.batch; ++b) same_bases = same_bases && members.base_off[b] == members.base_off[0];
    if (ctx->small_path != 0 && n <= ((size_t)1 << 16) && same_bases && p.B <= 16384 &&
        (size_t)Wt * p.B + n * (size_t)Wt / kmax + 1 <= (ws.cap_counts < ws.cap_tasks ? ws.cap_counts : ws.cap_tasks)) {  // smsm.hip: 4-5 launches in all
        int rc = smsm_enqueue(ctx, ws, d_bases, members.base_off[0], n, p, Wt, kmax);
        if (rc) return rc;
        HALO_HIP(hipGetLastError());
        if (!ctx->sink_done) HALO_HIP(hipMemcpyAsync(ws.h_winsum, ws.d_winsum, (size_t)Wt * 96, hipMemcpyDeviceToHost, s));
        ws.plan = p;
        return HALO_OK;
    }
    // one block per (window, chunk): about one block per CU, chunks of at least 1024 scalars
    uint32_t nchunks = 256u / Wt;
    if (nchunks < 1) nchunks = 1;
    while (nchunks > 1 && (n + nchunks - 1) / nchunks < 1024) nchunks--;
    if ((size_t)Wt * nchunks * p.B > ws.cap_hist) { set_error("msm: window plan exceeds workspace"); return HALO_E_ARG; }
    uint32_t chunk_len = (uint32_t)((n + nchunks - 1) / nchunks);
    int vec = n % 8 == 0 ? 1 : 0;  // digit rows stay 16-byte aligned: vector loads of eight digits
    if (vec) chunk_len = (chunk_len + 7) / 8 * 8;
    dim3 gridh((unsigned)(Wt * nchunks)), b1024(1024);
    size_t lds_bytes = (size_t)p.B * 4;
    uint32_t nblocks = (uint32_t)((total + 4095) / 4096);
    // large MSMs: two-level sort (coarse runs, then a fine sort per run) -- every store lands next to recent ones
    uint32_t NC = p.B >> FINE_BITS;
    bool two_level = vec && p.B >= (1u << FINE_BITS) && (size_t)Wt * NC <= 4096 && ws.d_presort &&
                     (ctx->sort_two_level > 0 || (ctx->sort_two_level < 0 && n >= ((size_t)1 << 17)));
    if (two_level) {
        uint32_t P = Wt * NC;
        uint32_t *chist = ws.d_hist, *cstart = ws.d_hist + 16384;  // Wt * nchunks * NC <= 8192 and P + 1 <= 4097 words
        HALO_LAUNCH(ctx, "k_msm_coarse_hist", k_msm_coarse_hist, gridh, b1024, 0, d_digits, (uint32_t)n, NC, nchunks, chunk_len, chist);
        HALO_LAUNCH(ctx, "k_msm_coarse_scan", k_msm_coarse_scan, dim3(1), b1024, 0, chist, P, NC, nchunks, cstart);
        int packed = n <= ((size_t)1 << 21) ? 1 : 0;  // index (21 bits) + fine bucket bits (10) + sign fit one word
        HALO_LAUNCH(ctx, "k_msm_coarse_scatter", k_msm_coarse_scatter, gridh, b1024, 0, d_digits, (uint32_t)n, NC, nchunks, chunk_len, chist, cstart,
                    packed, ws.d_presort);
        if (packed)
            HALO_LAUNCH(ctx, "k_msm_fine_sort", k_msm_fine_sort<true>, dim3(P), b1024, FINE_STAGE * 4, ws.d_presort, d_digits, (uint32_t)n, p.B, NC, cstart, Wm,
                        offs, kmax, (uint32_t)total, ws.d_counts, ws.d_starts, ws.d_ntask, ws.d_toff, ws.d_task_g, ws.d_biglist, ws.d_meta, ws.d_sorted);
        else
            HALO_LAUNCH(ctx, "k_msm_fine_sort", k_msm_fine_sort<false>, dim3(P), b1024, FINE_STAGE * 4, ws.d_presort, d_digits, (uint32_t)n, p.B, NC, cstart, Wm,
                        offs, kmax, (uint32_t)total, ws.d_counts, ws.d_starts, ws.d_ntask, ws.d_toff, ws.d_task_g, ws.d_biglist, ws.d_meta, ws.d_sorted);
    } else {
        HALO_LAUNCH(ctx, "k_msm_hist", k_msm_hist, gridh, b1024, lds_bytes, d_digits, (uint32_t)n, p.B, nchunks, chunk_len, vec, ws.d_hist);
        HALO_LAUNCH(ctx, "k_msm_colsum", k_msm_colsum, dim3((unsigned)((total + 255) / 256)), b256, 0, ws.d_hist, p.B, nchunks, (uint32_t)total,
                    kmax, ws.d_counts, ws.d_ntask);
        HALO_LAUNCH(ctx, "k_scan_blocks", k_scan_blocks, dim3(nblocks), b256, 0, ws.d_counts, (uint32_t)total, ws.d_starts, ws.d_blockoff);
        HALO_LAUNCH(ctx, "k_scan_top", k_scan_top, dim3(1), dim3(1024), 0, ws.d_blockoff, nblocks);
        HALO_LAUNCH(ctx, "k_msm_scatter", k_msm_scatter, gridh, b1024, lds_bytes, d_digits, (uint32_t)n, p.B, nchunks, chunk_len, ws.d_hist,
                    ws.d_starts, ws.d_blockoff, Wm, offs, vec, ws.d_sorted);
    }
    size_t max_tasks = total + n * (size_t)Wt / kmax + 1;
    if (max_tasks > ws.cap_tasks) max_tasks = ws.cap_tasks;
    dim3 gridt((unsigned)((max_tasks + 255) / 256));
    if (!two_level) {  // (the two-level sort's fine pass has already written the task lists)
        HALO_LAUNCH(ctx, "k_scan_blocks", k_scan_blocks, dim3(nblocks), b256, 0, ws.d_ntask, (uint32_t)total, ws.d_toff, ws.d_tblockoff);
        HALO_LAUNCH(ctx, "k_scan_top", k_scan_top, dim3(1), dim3(1024), 0, ws.d_tblockoff, nblocks);
        HALO_LAUNCH(ctx, "k_msm_task_bins", k_msm_task_bins, gridt, b256, 0, ws.d_ntask, ws.d_toff, ws.d_tblockoff, ws.d_counts, (uint32_t)total, kmax,
                    ws.d_meta, ws.d_task_g, ws.d_biglist);
    }
    HALO_LAUNCH(ctx, "k_msm_task_order", k_msm_task_order, gridt, b256, 0, ws.d_task_g, ws.d_meta, ws.d_sorted, ws.d_starts, ws.d_blockoff, ws.d_counts,
                ws.d_toff, ws.d_tblockoff, kmax, reinterpret_cast<uint4 *>(ws.d_order));
    HALO_LAUNCH(ctx, "k_msm_accumulate", k_msm_accumulate, gridt, b256, 0, d_bases, ws.d_sorted, ws.d_meta, reinterpret_cast<const uint4 *>(ws.d_order),
                ws.d_buckets);
    HALO_LAUNCH(ctx, "k_msm_combine", k_msm_combine, dim3(512 + 1024), dim3(64), 0, ws.d_ntask, ws.d_toff, ws.d_tblockoff, ws.d_meta, ws.d_biglist,
                (uint32_t)total, 512u, ws.d_buckets);
    uint32_t L, nseg;
    int logL = 0;
    // buckets per lane (L) against segments per window: a lane's 2L serial adds are all useful work,
    // the ~16 wave-wide scan steps that follow are mostly not, so L grows with the window
    if (p.B <= 64) { nseg = 1; L = 1; }
    else {
        L = p.B >= 16384 ? 8 : p.B >= 4096 ? 4 : p.B >= 1024 ? 2 : 1;
        nseg = p.B / (64 * L);
    }
    {
        // no more waves than SIMDs (one wave each): a SIMD that holds two runs both chains at about half speed and the kernel
        // waits for it (development switch HALO_REDUCE1_WAVES, 0 = off)
        const int waves_env = tuning().reduce1_waves;
        while (waves_env > 0 && p.B > 64 && (size_t)Wt * nseg > (size_t)waves_env && nseg > 1 && L < 64) { L <<= 1; nseg >>= 1; }
    }
    if (ctx->reduce_span > 0 && p.B > 64) {
        L = (uint32_t)ctx->reduce_span;
        while (64 * L > p.B) L >>= 1;
        while (p.B / (64 * L) > 64) L <<= 1;
        nseg = p.B / (64 * L);
    }
    while ((1u << logL) < L) logL++;
    HALO_LAUNCH(ctx, "k_msm_reduce1", k_msm_reduce1, dim3((unsigned)(Wt * nseg)), dim3(64), 0, ws.d_buckets, ws.d_ntask, ws.d_toff,
                ws.d_tblockoff, p.B, L, logL, nseg, ws.d_seg);
    {
        int rc = quad_final_enqueue(ctx, ws, (uint32_t)Wt, nseg, logL + 6, ctx->sink_done ? ws.h_winsum : ws.d_winsum, nullptr, nullptr, ctx->sink_done);
        if (rc) return rc;
    }
    HALO_HIP(hipGetLastError());
    if (!ctx->sink_done) HALO_HIP(hipMemcpyAsync(ws.h_winsum, ws.d_winsum, (size_t)Wt * 96, hipMemcpyDeviceToHost, s));
    ws.plan = p;
    return HALO_OK;
}

}  // namespace halo
