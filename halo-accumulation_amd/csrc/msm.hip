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
#include <atomic>
#include <cstring>
#include <thread>

#include "curve_quad.hpp"
#include "internal.hpp"

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
// Signed digit of window w: v = bits + carry; v > B  =>  v - 2^c (carry 1).  Top window never
// carries out because c*W >= 256 > 255 bits.  Returns magnitude (0 = skip) and sign.
struct Digit { uint32_t mag; uint32_t neg; };
HALO_DEV Digit next_digit(const uint32_t *words /*9 words in LDS*/, int w, int c, uint32_t B, uint32_t &carry) {
    uint32_t bit = (uint32_t)w * (uint32_t)c;
    uint32_t word = bit >> 5, sh = bit & 31;
    uint64_t two = (uint64_t)words[word] | ((uint64_t)words[word + 1] << 32);
    uint32_t raw = (uint32_t)(two >> sh) & ((1u << c) - 1u);
    uint32_t v = raw + carry;
    Digit d;
    if (v > B) { d.mag = (1u << c) - v; d.neg = 1; carry = 1; }
    else { d.mag = v; d.neg = 0; carry = 0; }
    return d;
}

// Digits are stored once as u16: (|d| - 1) | sign << 15, 0xFFFF for a zero digit (|d| - 1 = 2^15 - 1
// with the sign set cannot occur: negative digits have magnitude <= B - 1).  Layout [w][i].
constexpr uint32_t DIGIT_NONE = 0xFFFFu;
constexpr uint32_t KMAX = 64;  // largest task length of the bucket kernel (plan.kmax <= KMAX)

// Windows [w0, w1) are written (a window shard still walks the carry chain from window 0).
// Block (0, 0) also clears the launch's small state (meta: 256 words; zero_b: the 1024 block offsets where the sort
// writes absolute bucket starts) -- nothing reads either before the sort passes that follow.  blockIdx.y = member
// of a batched launch.  A window shard (w0 > 0) does not walk the carry chain from window 0: the carry into w0 is
// decided by the nearest lower window whose raw digit differs from B (raw < B: 0, raw > B: 1, raw == B: passes on).
struct MemberScalars { const uint64_t *p[MSM_MAX_BATCH]; };
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
struct MemberOffsets { uint32_t v[MSM_MAX_BATCH]; };
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
constexpr uint32_t FINE_STAGE = 36864;             // entries staged in LDS by the fine pass: 144 KiB of the CU's 160 KiB

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

// ------------------------------------------------------------------------------ fixed-base tables (the commitment key is constant)
// The key of a context never changes (consts.rs:68: GS is a compile-time constant of the reference), and a 288 GB device has
// room to spare, so for large MSMs over the context's own bases the context keeps T[w][i] = 2^(c w) G_i for every window w
// (c = 20: 13 windows, 13 x 128 B per point).  A signed digit d of window w then sends the point T[w][i] to bucket |d| of ONE
// set of 2^19 buckets shared by all windows:
//   * 13 n mixed additions instead of 16 n (the general pipeline cannot go past c = 16: every window would need its own
//     2^(c-1) buckets reduced);
//   * the weighted bucket sum is taken once over 2^19 buckets -- as many as 16 windows x 2^15 today -- and the host no longer
//     runs a 240-doubling Horner chain: it combines 16 (plain, weighted) pairs with ~50 additions.
// The sort is the two-level one (coarse: 512 bucket ranges, runs advance sequentially; fine: one block per range, 1024 buckets,
// staged in LDS); everything after it (tasks, k_msm_accumulate, combine, window sums) is the general pipeline's, which sees
// 16 "virtual windows" of 2^15 buckets.
//
// Two plans (TblPlan, fixed when a context builds its table):
//   key of >= 2^20 points: c = 20, 13 windows, 2^19 buckets, 512 coarse ranges of 1024 buckets, 16 virtual windows of 2^15;
//   key of 2^17 .. 2^19 points (a rank's index shard of a 2^20-point MSM): c = 17, 15 windows (15 x 17 = 255 bits exactly),
//     2^16 buckets, n / 2048 coarse ranges (so that a range's 15 n / ranges entries fit the fine sort's LDS stage), 16 virtual
//     windows of 2^12 -- 15 n additions against 16 n, an eighth of the buckets of the general plan's 16 x 2^15.
constexpr uint32_t TBL_MAX_RANGES = 512;
// An MSM of more points than this runs as consecutive pieces on its stream (c = 20 plan): a coarse range of a piece then
// holds <= 13 * 1310720 / 512 = 33 k entries and fits the fine sort's LDS stage, buckets keep ~26-33 entries (one task
// each) -- at 2^21 .. 2^24 points in one piece the fine sort placed entries straight to HBM (0.36 ms at 2^21, 8.5 ms at
// 2^24) and k_msm_combine folded 2 .. 8 partials per bucket.  The pieces' window sums are added on the host.
constexpr size_t TBL_PIECE = 1310720;
constexpr uint32_t TDIGIT_NONE = 0xFFFFFFFFu;
TblPlan table_plan(size_t key_n) {
    TblPlan t{};
    if (key_n >= ((size_t)1 << 20)) {
        t.c = 20; t.W = 13; t.B = 1u << 19; t.fbits = 10; t.vw_bits = 15; t.spread = 31; t.fold_top = 0;
    } else {
        t.c = 17; t.W = 15; t.B = 1u << 16; t.vw_bits = 12; t.spread = 0; t.fold_top = 1;
        uint32_t ranges = 64;
        while ((size_t)ranges * 2048 < key_n && ranges < 256) ranges <<= 1;
        t.fbits = 16;
        for (uint32_t r = ranges; r > 1; r >>= 1) t.fbits--;
    }
    t.ranges = t.B >> t.fbits;
    t.vw = t.B >> t.vw_bits;
    return t;
}

// next[i] = 2^c * prev[i], affine in, affine out.  A lane takes TBL_E points (i, i + stride, ...) and brings them back to
// affine with one shared inversion (curve.hpp jac_batch_to_aff): the inversion was 70 % of a point's work.
constexpr int TBL_E = 4;
__global__ __launch_bounds__(256) void k_table_step(const uint32_t *__restrict__ prev, uint32_t n, int c, uint32_t *__restrict__ next) {
    uint32_t t = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
    if (t >= n) return;
    JacN p[TBL_E];
    static_for<0, TBL_E>([&](auto ic) {
        constexpr int e = decltype(ic)::value;
        uint32_t i = t + (uint32_t)e * stride;
        p[e] = i < n ? jac_from_aff(aff_load(prev + AFF_STRIDE * (size_t)i)) : jac_inf();
    });
#pragma unroll 1
    for (int k = 0; k < c; k++) static_for<0, TBL_E>([&](auto ic) { p[decltype(ic)::value] = jac_dbl(p[decltype(ic)::value]); });
    AffN a[TBL_E];
    jac_batch_to_aff(p, a);
    static_for<0, TBL_E>([&](auto ic) {
        constexpr int e = decltype(ic)::value;
        uint32_t i = t + (uint32_t)e * stride;
        if (i < n) aff_store(next + AFF_STRIDE * (size_t)i, a[e]);
    });
}

// signed 20-bit digits, u32 [w][i]: (|d| - 1) | sign << 31, TDIGIT_NONE for zero; block 0 clears the launch's small state
// blockIdx.y = member of a batched launch (small-key plan): its digits go to rows [member W, (member + 1) W) and carry
// member * B on top of the bucket number, so that every later kernel sees one MSM with count * B buckets.
struct TblScalars { const uint64_t *p[MSM_MAX_BATCH]; };
// tagged (MsmBatch::tagged): ONE scalar array, canonical, whose bit 255 picks the bucket set of point i (0 / 1) the way blockIdx.y
// does for the members of a batch: rows stay W, buckets become 2 B.
__global__ __launch_bounds__(256) void k_tmsm_recode(TblScalars members, int mont, int tagged, uint32_t n, TblPlan tp, uint32_t *__restrict__ digits,
                                                     uint32_t *__restrict__ meta, uint32_t *__restrict__ zero_b, uint32_t *__restrict__ zero_t) {
    __shared__ uint32_t sw[256 * 9];
    const uint64_t *__restrict__ scalars = members.p[blockIdx.y];
    if (blockIdx.x == 0 && blockIdx.y == 0) {
        meta[threadIdx.x] = 0;
        for (int k = 0; k < 4; k++) { zero_b[threadIdx.x + 256 * k] = 0; zero_t[threadIdx.x + 256 * k] = 0; }
    }
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    Fe s = fe_load(scalars + 4 * (size_t)i);
    if (mont) s = fe_from_mont<FrCfg>(s);
    uint32_t *my = sw + threadIdx.x * 9;
#pragma unroll
    for (int k = 0; k < 8; k++) my[k] = s.v[k];
    my[8] = 0;
    uint32_t set = blockIdx.y;
    if (tagged) { set = my[7] >> 31; my[7] &= 0x7fffffffu; }
    uint32_t flip = 0;
    if (tp.fold_top && (my[7] >> 30) != 0) {
        // c = 17: the top window would hold 2^16 (+ carry) = B + 1 for a scalar >= 2^254 -- one value too many.  Such a
        // scalar is within 2^126 of r: take r - s with every digit's sign flipped (or s - r for an unreduced input).
        bool ge = true;  // s >= r ?
#pragma unroll
        for (int j = 7; j >= 0; j--) {
            if (my[j] != FrCfg::P[j]) { ge = my[j] > FrCfg::P[j]; break; }
        }
        uint64_t borrow = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint64_t a = ge ? my[j] : FrCfg::P[j], b = ge ? FrCfg::P[j] : my[j];
            uint64_t d = a - b - borrow;
            my[j] = (uint32_t)d;
            borrow = (d >> 32) & 1u;
        }
        flip = ge ? 0u : 1u;
    }
    if (tp.spread) {
        // The top window holds 255 - 240 = 15 scalar bits: its 2^20 digits would all land in the lowest 16 of the 512
        // coarse ranges (3.7 x the entries of the others: their fine-sort blocks ran 110-160 us against 15 us, and the
        // kernel waited for them).  Every base has order r, so s + k r gives the same point for any k: k = i mod 31
        // spreads the top digit floor((s + k r) / 2^240) evenly over [0, 31 * 2^14] <= 2^19 at no cost.  Scalars with an
        // empty top window (zero, short challenges) add nothing to it and stay as they are; so does anything >= 2^254 + 2^240.
        // (plan c = 20 only: tp.spread = 31)
        uint32_t top = my[7] >> 16;
        uint32_t k = (top != 0 && top <= 16384u) ? i % tp.spread : 0u;
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
    for (int w = 0; w < tp.W; w++) {
        Digit d = next_digit(my, w, tp.c, tp.B, carry);
        digits[((size_t)blockIdx.y * tp.W + w) * n + i] = d.mag ? ((d.mag - 1 + set * tp.B) | ((d.neg ^ flip) << 31)) : TDIGIT_NONE;
    }
}
// block (w, chunk): counts of the 512 coarse ranges, one private row per wave
// (MAXR = 1024: the c = 20 plan with TWO bucket sets, a `tagged` launch -- the kernels with a 2 in their names)
template <uint32_t MAXR>
HALO_DEV void tmsm_coarse_hist_body(const uint32_t *__restrict__ digits, uint32_t n, uint32_t nchunks, uint32_t chunk_len, const TblPlan &tp,
                                    uint32_t *__restrict__ chist) {
    __shared__ uint32_t cnt[16 * MAXR];
    uint32_t w = blockIdx.x / nchunks, chunk = blockIdx.x % nchunks;
    for (uint32_t k = threadIdx.x; k < 16 * MAXR; k += 1024) cnt[k] = 0;
    __syncthreads();
    uint32_t *mine = cnt + MAXR * (threadIdx.x >> 6);
    uint32_t lo = chunk * chunk_len, hi = lo + chunk_len < n ? lo + chunk_len : n;
    const uint32_t *dg = digits + (size_t)w * n;
    for (uint32_t i = lo + 4 * threadIdx.x; i < hi; i += 4 * 1024) {  // n and chunk_len are multiples of 4
        uint4 q = *reinterpret_cast<const uint4 *>(dg + i);
        uint32_t v[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (v[k] != TDIGIT_NONE) atomicAdd(&mine[(v[k] & (tp.B - 1u)) >> tp.fbits], 1u);
    }
    __syncthreads();
    if (threadIdx.x < tp.ranges) {
        uint32_t t = 0;
        for (int r = 0; r < 16; r++) t += cnt[MAXR * r + threadIdx.x];
        chist[(size_t)blockIdx.x * tp.ranges + threadIdx.x] = t;
    }
}
__global__ __launch_bounds__(1024) void k_tmsm_coarse_hist(const uint32_t *__restrict__ digits, uint32_t n, uint32_t nchunks, uint32_t chunk_len,
                                                           TblPlan tp, uint32_t *__restrict__ chist) {
    tmsm_coarse_hist_body<TBL_MAX_RANGES>(digits, n, nchunks, chunk_len, tp, chist);
}
__global__ __launch_bounds__(1024) void k_tmsm_coarse_hist2(const uint32_t *__restrict__ digits, uint32_t n, uint32_t nchunks, uint32_t chunk_len,
                                                            TblPlan tp, uint32_t *__restrict__ chist) {
    tmsm_coarse_hist_body<2 * TBL_MAX_RANGES>(digits, n, nchunks, chunk_len, tp, chist);
}
// chist[chunk][range] -> exclusive prefix over the chunks of each range (in place); rtotal[range] = the range's size.
// One block per range: 247 counters, loaded once, scanned in LDS.
__global__ __launch_bounds__(256) void k_tmsm_scan_chunks(uint32_t *__restrict__ chist, uint32_t nchunks_all, uint32_t ranges, uint32_t *__restrict__ rtotal) {
    __shared__ uint32_t part[256];
    uint32_t r = blockIdx.x, t = threadIdx.x;
    uint32_t v = t < nchunks_all ? chist[(size_t)t * ranges + r] : 0u;  // nchunks_all <= 256
    part[t] = v;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        uint32_t o = t >= (uint32_t)off ? part[t - off] : 0u;
        __syncthreads();
        part[t] += o;
        __syncthreads();
    }
    if (t < nchunks_all) chist[(size_t)t * ranges + r] = part[t] - v;
    if (t == 255) rtotal[r] = part[255];
}
// cstart[r] = start of run r in the presorted array, cstart[512] = number of entries
__global__ __launch_bounds__(1024) void k_tmsm_scan_ranges(const uint32_t *__restrict__ rtotal, uint32_t *__restrict__ cstart) {
    __shared__ uint32_t part[2 * TBL_MAX_RANGES];
    const uint32_t ranges = blockDim.x;  // one thread per range
    uint32_t t = threadIdx.x, v = rtotal[t];
    part[t] = v;
    __syncthreads();
    for (uint32_t off = 1; off < ranges; off <<= 1) {
        uint32_t o = t >= off ? part[t - off] : 0u;
        __syncthreads();
        part[t] += o;
        __syncthreads();
    }
    cstart[t] = part[t] - v;
    if (t == ranges - 1) cstart[ranges] = part[t];
}
// Block (w, chunk) appends its entries to the 512 runs: table index | sign << 31, and the bucket's low 10 bits beside it.
// 247 blocks x 512 runs are too many open cache lines for direct appends (partially filled lines would be evicted and
// rewritten), so the block goes through its chunk in tiles of 8192 entries: a tile is grouped by range in LDS (local
// ranks from an LDS histogram), then written out in that order -- entries of one range land on consecutive addresses.
constexpr uint32_t TBL_TILE = 8192;
template <uint32_t MAXR>
HALO_DEV void tmsm_coarse_scatter_body(const uint32_t *__restrict__ digits, uint32_t n, uint32_t nchunks, uint32_t chunk_len,
                                       const uint32_t *__restrict__ chist, const uint32_t *__restrict__ cstart, uint32_t table_n, uint32_t base_off,
                                       const TblPlan &tp, uint32_t *__restrict__ presort, uint16_t *__restrict__ presort_fine) {
    __shared__ uint32_t cur[MAXR], tcount[MAXR], toff[MAXR], wsum[MAXR / 64];
    const uint32_t ranges = tp.ranges, fmask = (1u << tp.fbits) - 1u;
    __shared__ uint32_t t_idx[TBL_TILE], t_dest[TBL_TILE];
    __shared__ uint16_t t_fine[TBL_TILE];
    uint32_t row = blockIdx.x / nchunks, chunk = blockIdx.x % nchunks, tid = threadIdx.x;  // row = member * W + w
    if (tid < ranges) cur[tid] = cstart[tid] + chist[(size_t)blockIdx.x * ranges + tid];
    uint32_t lo = chunk * chunk_len, hi = lo + chunk_len < n ? lo + chunk_len : n;
    const uint32_t *dg = digits + (size_t)row * n;
    uint32_t tbase = (row % (uint32_t)tp.W) * table_n + base_off;
    for (uint32_t t0 = lo; t0 < hi; t0 += TBL_TILE) {
        if (tid < ranges) tcount[tid] = 0;
        __syncthreads();
        // eight digits per thread: two 16-byte loads (n, chunk_len and the tile are multiples of 4)
        uint32_t v[8], rank[8];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            uint32_t i = t0 + 4 * tid + (uint32_t)h * 4096;
            uint4 q = i < hi ? *reinterpret_cast<const uint4 *>(dg + i) : make_uint4(TDIGIT_NONE, TDIGIT_NONE, TDIGIT_NONE, TDIGIT_NONE);
            v[4 * h] = q.x; v[4 * h + 1] = q.y; v[4 * h + 2] = q.z; v[4 * h + 3] = q.w;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) rank[k] = v[k] != TDIGIT_NONE ? atomicAdd(&tcount[(v[k] & (tp.B - 1u)) >> tp.fbits], 1u) : 0u;
        __syncthreads();
        {   // inclusive scan of the tile's counts: shuffles within a wave, the <= 8 wave totals through LDS -- two barriers
            // (a Hillis-Steele pass over LDS took 18, per tile, for sixteen waves)
            uint32_t x = tid < ranges ? tcount[tid] : 0u;
            uint32_t lane = tid & 63u;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                uint32_t o = (uint32_t)__shfl_up((int)x, off, 64);
                if (lane >= (uint32_t)off) x += o;
            }
            if (lane == 63 && tid < MAXR) wsum[tid >> 6] = x;
            __syncthreads();
            uint32_t before = 0;
#pragma unroll
            for (uint32_t wv = 0; wv < MAXR / 64; wv++) before += (wv < (tid >> 6)) ? wsum[wv] : 0u;
            if (tid < ranges) toff[tid] = x + before;
            __syncthreads();
        }
        uint32_t total = toff[ranges - 1];
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (v[k] != TDIGIT_NONE) {
                uint32_t r = (v[k] & (tp.B - 1u)) >> tp.fbits;
                uint32_t slot = toff[r] - tcount[r] + rank[k];
                uint32_t i = t0 + 4 * tid + (uint32_t)(k >> 2) * 4096 + (uint32_t)(k & 3);
                t_idx[slot] = (tbase + i) | (v[k] & 0x80000000u);
                t_fine[slot] = (uint16_t)(v[k] & fmask);
                t_dest[slot] = cur[r] + rank[k];
            }
        __syncthreads();
        for (uint32_t j = tid; j < total; j += 1024) {
            uint32_t d = t_dest[j];
            presort[d] = t_idx[j];
            presort_fine[d] = t_fine[j];
        }
        __syncthreads();
        if (tid < ranges) cur[tid] += tcount[tid];
    }
}
__global__ __launch_bounds__(1024) void k_tmsm_coarse_scatter(const uint32_t *__restrict__ digits, uint32_t n, uint32_t nchunks, uint32_t chunk_len,
                                                              const uint32_t *__restrict__ chist, const uint32_t *__restrict__ cstart,
                                                              uint32_t table_n, uint32_t base_off, TblPlan tp, uint32_t *__restrict__ presort,
                                                              uint16_t *__restrict__ presort_fine) {
    tmsm_coarse_scatter_body<TBL_MAX_RANGES>(digits, n, nchunks, chunk_len, chist, cstart, table_n, base_off, tp, presort, presort_fine);
}
__global__ __launch_bounds__(1024) void k_tmsm_coarse_scatter2(const uint32_t *__restrict__ digits, uint32_t n, uint32_t nchunks, uint32_t chunk_len,
                                                               const uint32_t *__restrict__ chist, const uint32_t *__restrict__ cstart,
                                                               uint32_t table_n, uint32_t base_off, TblPlan tp, uint32_t *__restrict__ presort,
                                                               uint16_t *__restrict__ presort_fine) {
    tmsm_coarse_scatter_body<2 * TBL_MAX_RANGES>(digits, n, nchunks, chunk_len, chist, cstart, table_n, base_off, tp, presort, presort_fine);
}
// Fine sort of run r: counts and absolute starts of its 1024 buckets, entries placed in [lo, hi) of `sorted` -- and the
// task lists the general pipeline builds with four more kernels (k_scan_blocks/_top, k_msm_task_bins, k_msm_task_order):
// a bucket of c entries is ceil(c / kmax) tasks with consecutive ids; the block reserves its ids with one atomic
// (meta[0]), writes toff[g] = first id (absolute: the block offsets of the two-level scan format stay zero) and task_g, and
// adds its tasks-per-length counts to meta[2 ..]: k_msm_task_order then lays the tasks out by decreasing length over the
// WHOLE launch (longest first: with a per-block order the last waves of k_msm_accumulate were long ones, +35 % on it).
// Multi-task buckets are listed for k_msm_combine (meta[1], meta[140]).
constexpr uint32_t TBL_STAGE = 35840;  // entries staged in LDS: 140 KiB next to 20 KiB of counters
__global__ __launch_bounds__(1024) void k_tmsm_fine_sort(const uint32_t *__restrict__ presort, const uint16_t *__restrict__ presort_fine,
                                                         const uint32_t *__restrict__ cstart, uint32_t kmax, TblPlan tp, uint32_t *__restrict__ counts,
                                                         uint32_t *__restrict__ starts, uint32_t *__restrict__ ntask, uint32_t *__restrict__ toff,
                                                         uint32_t *__restrict__ task_g, uint32_t *__restrict__ biglist,
                                                         uint32_t *__restrict__ meta, uint32_t *__restrict__ sorted) {
    __shared__ uint32_t hist[1024], scan[1024], tscan[1024], lbin[KMAX + 8], misc[2];
    uint32_t r = blockIdx.x, lo = cstart[r], hi = cstart[r + 1], tid = threadIdx.x;
#ifdef TMSM_TIMING
    uint64_t tm[10]; int tmi = 0;
#define TMARK() do { __syncthreads(); tm[tmi++] = wall_clock64(); } while (0)
#else
#define TMARK() do {} while (0)
#endif
    TMARK();
    hist[tid] = 0;
    if (tid < KMAX + 8) lbin[tid] = 0;
    // A run that fits the LDS stage (every run, for uniform scalars) is read ONCE, all loads in flight together, and kept
    // in registers across the counting and the placement: the kernel is bound by global-load latency (one block of 16
    // waves per CU), not by LDS or bandwidth.
    constexpr int PER = TBL_STAGE / 1024;
    bool staged = hi - lo <= TBL_STAGE;
    uint32_t rv[PER], rf[PER];
    if (staged) {
#pragma unroll
        for (int k = 0; k < PER; k++) {
            uint32_t e = lo + tid + (uint32_t)k * 1024;
            bool in = e < hi;
            rv[k] = in ? presort[e] : 0u;
            rf[k] = in ? (uint32_t)presort_fine[e] : 0xffffffffu;
        }
    }
    __syncthreads();
    TMARK();
    if (staged) {
#pragma unroll
        for (int k = 0; k < PER; k++)
            if (rf[k] != 0xffffffffu) atomicAdd(&hist[rf[k]], 1u);
    } else {
        for (uint32_t e = lo + tid; e < hi; e += 4 * 1024) {
            uint32_t f[4];
#pragma unroll
            for (int k = 0; k < 4; k++) f[k] = e + k * 1024 < hi ? presort_fine[e + k * 1024] : 0u;
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (e + k * 1024 < hi) atomicAdd(&hist[f[k]], 1u);
        }
    }
    __syncthreads();
    TMARK();
    bool owner = tid < (1u << tp.fbits);  // one bucket per thread
    uint32_t mine = owner ? hist[tid] : 0u, nt = (mine + kmax - 1) / kmax;
    scan[tid] = mine;
    tscan[tid] = nt;
    // lengths of this bucket's tasks: kmax for all but the last
    if (nt) {
        if (nt > 1) atomicAdd(&lbin[KMAX - kmax], nt - 1);
        atomicAdd(&lbin[KMAX - (mine - (nt - 1) * kmax)], 1u);
    }
    __syncthreads();
    for (uint32_t o = 1; o < 1024; o <<= 1) {
        uint32_t a = tid >= o ? scan[tid - o] : 0u, b = tid >= o ? tscan[tid - o] : 0u;
        __syncthreads();
        scan[tid] += a;
        tscan[tid] += b;
        __syncthreads();
    }
    TMARK();
    if (tid == 1023) misc[0] = atomicAdd(&meta[0], tscan[1023]);  // this block's task ids: [base, base + total)
    if (tid <= KMAX && lbin[tid]) atomicAdd(&meta[2 + tid], lbin[tid]);  // tasks per length, whole launch
    __syncthreads();
    TMARK();
    uint32_t begin = lo + scan[tid] - mine, tfirst = misc[0] + tscan[tid] - nt;
    if (owner) {
        uint32_t g = (r << tp.fbits) + tid;
        counts[g] = mine;
        ntask[g] = nt;
        starts[g] = begin;  // absolute: the block offsets of the two-level scan format are zeroed by the recode kernel
        toff[g] = tfirst;   // likewise
        for (uint32_t j = 0; j < nt; j++) {
            uint32_t len = j + 1 < nt ? kmax : mine - (nt - 1) * kmax;
            task_g[tfirst + j] = g | ((KMAX - len) << 24);
        }
        hist[tid] = begin;
    }
    // multi-task buckets for k_msm_combine: counted in LDS, one reservation per block and list (a global atomic per bucket
    // serialises on one address: 1 ms when a third of the buckets hold more than kmax entries)
    TMARK();
    uint32_t big_rank = 0, small_rank = 0;
    if (owner && nt > 8) big_rank = atomicAdd(&lbin[KMAX + 1], 1u);
    else if (owner && nt > 1) small_rank = atomicAdd(&lbin[KMAX + 2], 1u);
    __syncthreads();
    if (tid == 0) {
        lbin[KMAX + 3] = lbin[KMAX + 1] ? atomicAdd(&meta[1], lbin[KMAX + 1]) : 0u;
        lbin[KMAX + 4] = lbin[KMAX + 2] ? atomicAdd(&meta[140], lbin[KMAX + 2]) : 0u;
    }
    __syncthreads();
    if (owner && nt > 8) biglist[lbin[KMAX + 3] + big_rank] = (r << tp.fbits) + tid;
    else if (owner && nt > 1) biglist[tp.B - 1 - (lbin[KMAX + 4] + small_rank)] = (r << tp.fbits) + tid;
    extern __shared__ uint32_t stage[];  // TBL_STAGE entries
    TMARK();
    if (staged) {
#pragma unroll
        for (int k = 0; k < PER; k++)
            if (rf[k] != 0xffffffffu) stage[atomicAdd(&hist[rf[k]], 1u) - lo] = rv[k];
        __syncthreads();
        TMARK();
        for (uint32_t e = lo + tid; e < hi; e += 1024) sorted[e] = stage[e - lo];
#ifdef TMSM_TIMING
        TMARK();
        if (tid == 0 && (blockIdx.x % 100 == 0 || tm[8] - tm[0] > 3000))
            printf("fine block %u (%u entries): load %llu hist %llu scan %llu reserve %llu owner %llu biglist %llu place %llu write %llu (x10 ns)\n", blockIdx.x, hi - lo,
                   (unsigned long long)(tm[1] - tm[0]), (unsigned long long)(tm[2] - tm[1]), (unsigned long long)(tm[3] - tm[2]), (unsigned long long)(tm[4] - tm[3]),
                   (unsigned long long)(tm[5] - tm[4]), (unsigned long long)(tm[6] - tm[5]), (unsigned long long)(tm[7] - tm[6]), (unsigned long long)(tm[8] - tm[7]));
#endif
        return;
    }
    for (uint32_t e = lo + tid; e < hi; e += 4 * 1024) {  // an oversized run (skewed scalars): placed directly
        uint32_t v[4], f[4], pos[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            bool in = e + k * 1024 < hi;
            v[k] = in ? presort[e + k * 1024] : 0u;
            f[k] = in ? presort_fine[e + k * 1024] : 0u;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) pos[k] = e + k * 1024 < hi ? atomicAdd(&hist[f[k]], 1u) : 0u;
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (e + k * 1024 < hi) sorted[pos[k]] = v[k];
    }
#ifdef TMSM_TIMING
    TMARK();
    if (tid == 0) printf("fine block %u UNSTAGED (%u entries): total %llu (x10 ns)\n", blockIdx.x, hi - lo, (unsigned long long)(tm[tmi - 1] - tm[0]));
#endif
}

// ------------------------------------------------------------------------------ scan
// 4096 entries per block: local exclusive scan + block total.
__global__ __launch_bounds__(256) void k_scan_blocks(const uint32_t *__restrict__ in, uint32_t total, uint32_t *__restrict__ out,
                                                     uint32_t *__restrict__ blocksum) {
    __shared__ uint32_t part[256];
    uint32_t base = blockIdx.x * 4096 + threadIdx.x * 16;
    uint32_t loc[16];
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        uint32_t v = (base + k < total) ? in[base + k] : 0u;
        loc[k] = sum;
        sum += v;
    }
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        uint32_t v = (threadIdx.x >= (uint32_t)off) ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t excl = part[threadIdx.x] - sum;
#pragma unroll
    for (int k = 0; k < 16; k++)
        if (base + k < total) out[base + k] = excl + loc[k];
    if (threadIdx.x == 255) blocksum[blockIdx.x] = part[255];
}
// exclusive scan of up to 1024 block totals, in place
__global__ __launch_bounds__(1024) void k_scan_top(uint32_t *blocksum, uint32_t nblocks) {
    __shared__ uint32_t part[1024];
    uint32_t v0 = threadIdx.x < nblocks ? blocksum[threadIdx.x] : 0u;
    part[threadIdx.x] = v0;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        uint32_t v = (threadIdx.x >= (uint32_t)off) ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    if (threadIdx.x < nblocks) blocksum[threadIdx.x] = part[threadIdx.x] - v0;
}

// ------------------------------------------------------------------------------ accumulate
// Buckets are cut into tasks of at most KMAX entries so that no lane ever runs a chain longer
// than KMAX mixed adds, whatever the scalar distribution (all-equal scalars, or a top window
// with two real bits, put n/4 .. n points into one bucket).  ntask[g] = ceil(count/KMAX);
// toff = exclusive scan of ntask.  One lane per task; a bucket's value is the partial of its
// first task once k_msm_combine has folded the partials of multi-task buckets into it.
// (KMAX, the largest task length, is defined with the table pipeline above)

HALO_DEV uint32_t scan_at(const uint32_t *__restrict__ in_block, const uint32_t *__restrict__ blockoff, uint32_t g) {
    return in_block[g] + blockoff[g >> 12];
}

// Tasks are processed in order of decreasing length so that the 64 lanes of a wave run chains of
// (almost) equal length: bucket sizes are Poisson distributed and a wave otherwise waits for its
// longest lane (~68 % lane efficiency at 32 points per bucket).  Counting sort over 65 length bins,
// aggregated per block in LDS so that only <= 65 global atomics per block are issued.
// meta[2 .. 2+65) = bin totals, meta[70 .. 70+65) = bin cursors.
HALO_DEV void task_locate(const uint32_t *__restrict__ toff, const uint32_t *__restrict__ tblockoff,
                          const uint32_t *__restrict__ counts, uint32_t total_buckets, uint32_t kmax, uint32_t t, uint32_t &g,
                          uint32_t &len) {
    uint32_t lo = 0, hi = total_buckets - 1;
    while (lo < hi) {
        uint32_t mid = (lo + hi + 1) >> 1;
        if (scan_at(toff, tblockoff, mid) <= t) lo = mid; else hi = mid - 1;
    }
    g = lo;
    uint32_t first = (t - scan_at(toff, tblockoff, g)) * kmax;
    len = counts[g] - first;
    if (len > kmax) len = kmax;
}
// Also: meta[0] = number of tasks; the multi-task buckets are listed for the combine kernel (by the lane that holds a
// bucket's first task): meta[1] of them with more than 8 tasks from the front of biglist, meta[140] with 2..8 from its end.
__global__ __launch_bounds__(256) void k_msm_task_bins(const uint32_t *__restrict__ ntask, const uint32_t *__restrict__ toff,
                                                       const uint32_t *__restrict__ tblockoff, const uint32_t *__restrict__ counts,
                                                       uint32_t total_buckets, uint32_t kmax, uint32_t *__restrict__ meta,
                                                       uint32_t *__restrict__ task_g, uint32_t *__restrict__ biglist) {
    __shared__ uint32_t bins[KMAX + 1];
    uint32_t ntasks = scan_at(toff, tblockoff, total_buckets - 1) + ntask[total_buckets - 1];
    if (blockIdx.x == 0 && threadIdx.x == 0) meta[0] = ntasks;
    if (blockIdx.x * 256 >= ntasks) return;  // the grid covers the worst case; blocks past the last task leave at once
    if (threadIdx.x <= KMAX) bins[threadIdx.x] = 0;
    __syncthreads();
    uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t < ntasks) {
        uint32_t g, len;
        task_locate(toff, tblockoff, counts, total_buckets, kmax, t, g, len);
        task_g[t] = g | ((KMAX - len) << 24);  // bucket id (< 2^24) and bin = 64 - len
        atomicAdd(&bins[KMAX - len], 1u);
        if (t == scan_at(toff, tblockoff, g)) {  // first task of its bucket
            uint32_t nt = ntask[g];
            if (nt > 8) biglist[atomicAdd(&meta[1], 1u)] = g;
            else if (nt > 1) biglist[total_buckets - 1 - atomicAdd(&meta[140], 1u)] = g;
        }
    }
    __syncthreads();
    if (threadIdx.x <= KMAX && bins[threadIdx.x]) atomicAdd(&meta[2 + threadIdx.x], bins[threadIdx.x]);
}
// order[pos] = {task id, first entry of the task in `sorted`, its length, that first entry}: everything the bucket kernel
// needs to start its gather, in ONE coalesced 16-byte load (it used to chase order -> task_g -> two scans + counts ->
// sorted -> point: five dependent loads per task before the first addition)
__global__ __launch_bounds__(256) void k_msm_task_order(const uint32_t *__restrict__ task_g, uint32_t *__restrict__ meta,
                                                        const uint32_t *__restrict__ sorted, const uint32_t *__restrict__ starts,
                                                        const uint32_t *__restrict__ blockoff, const uint32_t *__restrict__ counts,
                                                        const uint32_t *__restrict__ toff, const uint32_t *__restrict__ tblockoff,
                                                        uint32_t kmax, uint4 *__restrict__ order) {
    __shared__ uint32_t bins[KMAX + 1], base[KMAX + 1], tot[KMAX + 1];
    if (blockIdx.x * 256 >= meta[0]) return;
    if (threadIdx.x <= KMAX) {
        bins[threadIdx.x] = 0;
        tot[threadIdx.x] = meta[2 + threadIdx.x];  // global bin totals (65 values), read once per block
    }
    __syncthreads();
    uint32_t t = blockIdx.x * 256 + threadIdx.x;
    bool live = t < meta[0];
    uint32_t bin = 0, rank = 0;
    uint4 rec = make_uint4(0u, 0u, 0u, 0u);
    if (live) {
        uint32_t tg = task_g[t];
        bin = tg >> 24;
        rank = atomicAdd(&bins[bin], 1u);
        uint32_t g = tg & 0xFFFFFFu;
        uint32_t first = (t - scan_at(toff, tblockoff, g)) * kmax;
        uint32_t cnt = counts[g] - first;
        if (cnt > kmax) cnt = kmax;
        uint32_t st = scan_at(starts, blockoff, g) + first;
        rec = make_uint4(t, st, cnt, sorted[st]);
    }
    __syncthreads();
    if (threadIdx.x <= KMAX) {
        uint32_t start = 0;  // exclusive prefix of the totals
        for (uint32_t k = 0; k < threadIdx.x; k++) start += tot[k];
        base[threadIdx.x] = start + (bins[threadIdx.x] ? atomicAdd(&meta[70 + threadIdx.x], bins[threadIdx.x]) : 0u);
    }
    __syncthreads();
    if (live) order[base[bin] + rank] = rec;
}

__global__ __launch_bounds__(256) void k_msm_accumulate(const uint32_t *__restrict__ bases, const uint32_t *__restrict__ sorted,
                                                        const uint32_t *__restrict__ meta, const uint4 *__restrict__ order,
                                                        uint32_t *__restrict__ partial) {
    uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    if (tid >= meta[0]) return;
    uint4 rec = order[tid];
    uint32_t t = rec.x, st = rec.y, cnt = rec.z;
    XyzzN acc = xyzz_inf();
    // the next point's coordinates are fetched before the current mixed add is issued, and ITS index one add earlier
    // still: index -> gather is a dependent pair of loads, and with the index fetched in the same iteration the wave
    // sat in s_waitcnt for a whole memory latency per addition (17 % of the kernel's wave cycles parked:
    // profiles/r03_sq_msm.json).  The index load is unconditional, clamped to the task's last entry: a conditional one
    // made the compiler wait for it -- and for the gather just issued -- at the end of the branch.
    uint32_t e = rec.w;
    uint32_t last = cnt - 1;
    uint32_t e1 = sorted[st + (1 < last ? 1 : last)];
    // (the sign of the digit picks the stored y or -y of the entry by address: aff_load_signed)
    AffN nxt = aff_load_signed(bases + AFF_STRIDE * (size_t)(e & 0x7fffffffu), (e >> 31) != 0);
    for (uint32_t k = 0; k < cnt; k++) {
        AffN p = nxt;
        e = e1;
        if (k + 1 < cnt) nxt = aff_load_signed(bases + AFF_STRIDE * (size_t)(e & 0x7fffffffu), (e >> 31) != 0);
        e1 = sorted[st + (k + 2 < last ? k + 2 : last)];
        xyzz_madd(acc, p);
    }
    xyzz_store(partial + XYZZ_WORDS * (size_t)t, acc);
}

// Folds the partials of multi-task buckets into the first one.  Blocks [0, small_blocks): one lane per bucket with 2..8
// tasks (grid-stride over the tail of biglist); the others: one wave per bucket with more than 8 tasks.
__global__ __launch_bounds__(64) void k_msm_combine(const uint32_t *__restrict__ ntask, const uint32_t *__restrict__ toff,
                                                    const uint32_t *__restrict__ tblockoff, const uint32_t *__restrict__ meta,
                                                    const uint32_t *__restrict__ biglist, uint32_t total, uint32_t small_blocks,
                                                    uint32_t *__restrict__ partial) {
    uint32_t lane = threadIdx.x;
    if (blockIdx.x < small_blocks) {
        for (uint32_t b = blockIdx.x * 64 + lane; b < meta[140]; b += small_blocks * 64) {
            uint32_t g = biglist[total - 1 - b];
            uint32_t nt = ntask[g], t0 = scan_at(toff, tblockoff, g);
            XyzzN acc = xyzz_load(partial + XYZZ_WORDS * (size_t)t0);
#pragma unroll 1
            for (uint32_t j = 1; j < nt; j++) {
                XyzzN q = xyzz_load(partial + XYZZ_WORDS * (size_t)(t0 + j));
                xyzz_add(acc, q);
            }
            xyzz_store(partial + XYZZ_WORDS * (size_t)t0, acc);
        }
        return;
    }
    for (uint32_t b = blockIdx.x - small_blocks; b < meta[1]; b += gridDim.x - small_blocks) {
        uint32_t g = biglist[b];
        uint32_t nt = ntask[g], t0 = scan_at(toff, tblockoff, g);
        XyzzN acc = xyzz_inf();
#pragma unroll 1
        for (uint32_t j = lane; j < nt; j += 64) {
            XyzzN q = xyzz_load(partial + XYZZ_WORDS * (size_t)(t0 + j));
            xyzz_add(acc, q);
        }
#pragma unroll 1
        for (int off = 32; off >= 1; off >>= 1) {
            XyzzN o = xyzz_shfl(acc, (lane + off) & 63);
            if ((int)lane < off) xyzz_add(acc, o);
        }
        if (lane == 0) xyzz_store(partial + XYZZ_WORDS * (size_t)t0, acc);
    }
}

// value of bucket g after the combine pass
HALO_DEV XyzzN bucket_value(const uint32_t *__restrict__ partial, const uint32_t *__restrict__ ntask, const uint32_t *__restrict__ toff,
                            const uint32_t *__restrict__ tblockoff, uint32_t g) {
    if (ntask[g] == 0) return xyzz_inf();
    return xyzz_load(partial + XYZZ_WORDS * (size_t)scan_at(toff, tblockoff, g));
}

// ------------------------------------------------------------------------------ reduce
// Lane l holds S (sum of its buckets) and T (their sum weighted 1..L relative to the lane's
// first bucket).  Returns in lane 0: S_tot = sum_l S_l and T_tot = sum_l (T_l + l * 2^k * S_l).
// `park` (36 words per lane, LDS) holds T while S is scanned: with S, T, a shuffled copy and the temporaries of an
// addition live together the kernel needed 258 VGPRs, i.e. one wave per SIMD and no room next to a 256-register
// wave of k_msm_accumulate; forcing 256 made it spill, and that spill -- scratch inside a replayed hipGraph after the
// queue's scratch had been re-assigned -- is what faulted on ROCm 7.2 in round 1 (build gate:
// csrc/check_resources.py).
HALO_DEV void wave_weighted_sum(XyzzN &S, XyzzN &T, int k, uint32_t *park, int live = 64) {
    int lane = threadIdx.x & 63;
    {
        uint32_t *mine = park + lane;  // word j of lane l at park[64 * j + l]: conflict-free
#pragma unroll
        for (int j = 0; j < 9; j++) {
            mine[64 * j] = T.x.v[j]; mine[64 * (9 + j)] = T.y.v[j]; mine[64 * (18 + j)] = T.zz.v[j]; mine[64 * (27 + j)] = T.zzz.v[j];
        }
    }
    // inclusive suffix scan: S_l <- sum_{j >= l} S_j   (lanes >= live hold infinity: their steps are skipped)
#pragma unroll 1
    for (int off = 1; off < live; off <<= 1) {
        XyzzN o = xyzz_shfl(S, (lane + off) & 63);
        if (lane + off < 64) xyzz_add(S, o);
    }
    // sum_{l>=1} suffix_l = sum_l l * S_l
    XyzzN V = (lane >= 1) ? S : xyzz_inf();
#pragma unroll 1
    for (int i = 0; i < k; i++) V = xyzz_dbl(V);
    {
        const uint32_t *mine = park + lane;
#pragma unroll
        for (int j = 0; j < 9; j++) {
            T.x.v[j] = mine[64 * j]; T.y.v[j] = mine[64 * (9 + j)]; T.zz.v[j] = mine[64 * (18 + j)]; T.zzz.v[j] = mine[64 * (27 + j)];
        }
    }
    xyzz_add(T, V);
    int top = 32;
    while (top >= live && top > 1) top >>= 1;  // first offset that still pairs two live lanes
    if (live <= 1) top = 0;
#pragma unroll 1
    for (int off = top; off >= 1; off >>= 1) {
        XyzzN o = xyzz_shfl(T, (lane + off) & 63);
        if (lane < off) xyzz_add(T, o);
    }
}

// one wave per (window, segment of 64*L buckets)
// at most 256 VGPRs (see wave_weighted_sum): a wave of this kernel can share a SIMD with a wave of k_msm_accumulate
__global__ __launch_bounds__(64) void k_msm_reduce1(const uint32_t *__restrict__ partial, const uint32_t *__restrict__ ntask,
                                                    const uint32_t *__restrict__ toff, const uint32_t *__restrict__ tblockoff, uint32_t B,
                                                    uint32_t L, int logL, uint32_t nseg, uint32_t *__restrict__ seg) {
    __shared__ uint32_t park[36 * 64];
    uint32_t w = blockIdx.x / nseg, s = blockIdx.x % nseg;
    uint32_t lane = threadIdx.x;
    uint32_t first = s * 64 * L + lane * L;
    XyzzN run = xyzz_inf(), tot = xyzz_inf();
#pragma unroll 1
    for (int j = (int)L - 1; j >= 0; j--) {
        uint32_t idx = first + (uint32_t)j;
        XyzzN b = xyzz_inf();
        if (idx < B) b = bucket_value(partial, ntask, toff, tblockoff, w * B + idx);
        xyzz_add(run, b);
        xyzz_add(tot, run);
    }
    wave_weighted_sum(run, tot, logL, park);
    if (lane == 0) {
        uint32_t *o = seg + 2 * XYZZ_WORDS * ((size_t)w * nseg + s);
        xyzz_store(o, run);
        xyzz_store(o + XYZZ_WORDS, tot);
    }
}
// The segments of a window are combined by k_smsm_final (smsm.hip, quad-parallel): see quad_final_enqueue.

// Window sums of ONE set of 2^19 buckets (the c = 20 table plan) by rows and columns of the bucket index b = hi 2^10 + lo:
//     sum_b (b + 1) B_b  =  sum_lo (lo + 1) C_lo  +  2^10 sum_hi hi R_hi,     C_lo = sum_hi B_(hi,lo),  R_hi = sum_lo B_(hi,lo)
// C and R are PLAIN sums: every lane adds `per` buckets and a shuffle tree finishes the rows or columns of its wave -- for the
// c = 20 plan 16 + 6 dependent additions where the running-sum form (k_msm_reduce1: 16 in the lane, then a wave-wide weighted
// sum of 16 more) takes 32, for the same 1024 waves.  The weights come afterwards, over rows + columns points instead of all
// buckets (k_rc_mid, quad-parallel, then ~70 additions on the host).
//   shape            buckets per set      rows x columns   per lane   lanes per row / column
//   c = 20, 1 set         2^19              512 x 1024        16            64 / 32
//   c = 17, 8 sets        2^16              256 x 256         16            16 / 16      (batches of the small-key plan:
//   c = 17, 4 sets        2^16              256 x 256          8            32 / 32       the chip's 65536 lanes read
//   c = 17, 2 sets        2^16              256 x 256          4            64 / 64       every bucket twice)
//   c = 17, 1 set         2^16              256 x 256          4            64 / 64      (512 waves: 4 + 6 additions deep)
// Per set: blocks [0, nb) take columns, [nb, 2 nb) rows, nb = buckets / (64 per).  ent (per set): columns, then rows.
struct RcShape { int lg_rows = 0, lg_cols = 0, per = 0; };
static inline uint32_t rc_points(const RcShape &r) { return r.per ? 2u * ((1u << r.lg_rows) + (1u << r.lg_cols)) / 64u : 0u; }  // (S, T) pairs x 2, per set
__global__ __launch_bounds__(64) void k_msm_reduce_rc(const uint32_t *__restrict__ partial, const uint32_t *__restrict__ ntask,
                                                      const uint32_t *__restrict__ toff, const uint32_t *__restrict__ tblockoff,
                                                      RcShape sh, uint32_t *__restrict__ ent) {
    const uint32_t lane = threadIdx.x;
    const uint32_t rows = 1u << sh.lg_rows, cols = 1u << sh.lg_cols, per = (uint32_t)sh.per;
    const uint32_t nb = (rows << sh.lg_cols) / (64 * per);
    const uint32_t set = blockIdx.x / (2 * nb), r = blockIdx.x % (2 * nb);
    const bool row = r >= nb;
    uint32_t g0, stride, width, slot;
    if (row) {
        width = cols / per;                                   // lanes per row: lo = sub + width i (coalesced)
        uint32_t hi = (r - nb) * (64 / width) + lane / width;
        g0 = (hi << sh.lg_cols) + (lane & (width - 1)); stride = width; slot = cols + hi;
    } else {
        width = rows / per;                                   // lanes per column: hi = sub + width i
        uint32_t col = r * (64 / width) + lane / width;
        g0 = ((lane & (width - 1)) << sh.lg_cols) + col; stride = width << sh.lg_cols; slot = col;
    }
    g0 += set * (rows << sh.lg_cols);
    XyzzN acc = xyzz_inf();
    XyzzN b = bucket_value(partial, ntask, toff, tblockoff, g0);
#pragma unroll 1
    for (uint32_t i = 0; i < per; i++) {
        XyzzN nb2 = xyzz_inf();
        if (i + 1 < per) nb2 = bucket_value(partial, ntask, toff, tblockoff, g0 + (i + 1) * stride);  // in flight during the addition
        xyzz_add(acc, b);
        b = nb2;
    }
    const uint32_t sub = lane & (width - 1);
#pragma unroll 1
    for (uint32_t off = width >> 1; off >= 1; off >>= 1) {
        XyzzN o = xyzz_shfl(acc, (int)((lane + off) & 63));
        if (sub < off) xyzz_add(acc, o);
    }
    if (sub == 0) xyzz_store(ent + XYZZ_WORDS * ((size_t)set * (rows + cols) + slot), acc);
}

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

// ------------------------------------------------------------------------------ workspace
static size_t max_counts() { return (size_t)16 * 32768; }  // c = 16 is the largest W*B over c in [4, 16]

struct WorkspaceNeed {
    size_t n, counts, sorted, tasks, hist, windows;
};
static void workspace_release(MsmWorkspace &ws) {
    if (debug_trace()) fprintf(stderr, "[halo] workspace release %p\n", (void *)&ws);
    uint64_t *p64[] = {ws.d_canon, ws.d_winsum};
    uint32_t *p32[] = {ws.d_buckets, ws.d_seg, ws.d_counts, ws.d_starts, ws.d_hist, ws.d_blockoff, ws.d_sorted, ws.d_presort, ws.d_ntask, ws.d_toff,
                       ws.d_tblockoff, ws.d_biglist, ws.d_meta, ws.d_task_g, ws.d_order};
    for (auto p : p64) (void)hipFree(p);
    for (auto p : p32) (void)hipFree(p);
    (void)hipFree(ws.d_fine16);
    if (ws.h_winsum) (void)hipHostFree(ws.h_winsum);
    if (ws.h_done) (void)hipHostFree(ws.h_done);
    for (auto &g : ws.graphs) if (g.exec) (void)hipGraphExecDestroy(g.exec);
    ws = MsmWorkspace();
}
static int workspace_alloc_buffers(MsmWorkspace &ws, const WorkspaceNeed &need);
static int workspace_alloc(MsmWorkspace &ws, const WorkspaceNeed &need) {
    int rc = workspace_alloc_buffers(ws, need);
    if (rc) workspace_release(ws);  // a failed allocation part-way leaves nothing behind
    return rc;
}
static int workspace_alloc_buffers(MsmWorkspace &ws, const WorkspaceNeed &need) {
    ws.cap_n = need.n;
    ws.cap_counts = need.counts;
    ws.cap_sorted = need.sorted;
    ws.cap_tasks = need.tasks;
    ws.cap_hist = need.hist;
    ws.cap_windows = need.windows;
    // the LDS histograms need up to 128 KiB of dynamic LDS per block (160 KiB per CU on gfx950)
    HALO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_msm_hist), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    HALO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_msm_scatter), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    HALO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_msm_fine_sort<true>), hipFuncAttributeMaxDynamicSharedMemorySize, FINE_STAGE * 4));
    HALO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_msm_fine_sort<false>), hipFuncAttributeMaxDynamicSharedMemorySize, FINE_STAGE * 4));
    HALO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_tmsm_fine_sort), hipFuncAttributeMaxDynamicSharedMemorySize, TBL_STAGE * 4));
    { int rc = smsm_prepare(); if (rc) return rc; }
    HALO_HIP(hipMalloc(&ws.d_canon, ws.cap_sorted * 2 + 64));  // u16 digits, n * W of them
    HALO_HIP(hipMalloc(&ws.d_hist, ws.cap_hist * 4));          // [w][chunk][b]
    HALO_HIP(hipMalloc(&ws.d_counts, ws.cap_counts * 4));
    HALO_HIP(hipMalloc(&ws.d_starts, ws.cap_counts * 4));
    HALO_HIP(hipMalloc(&ws.d_blockoff, 1024 * 4));
    HALO_HIP(hipMalloc(&ws.d_sorted, ws.cap_sorted * 4));
    HALO_HIP(hipMalloc(&ws.d_presort, ws.cap_sorted * 4));  // coarse runs of the two-level sort
    HALO_HIP(hipMalloc(&ws.d_buckets, ws.cap_tasks * XYZZ_WORDS * 4));
    HALO_HIP(hipMalloc(&ws.d_ntask, ws.cap_counts * 4));
    HALO_HIP(hipMalloc(&ws.d_toff, ws.cap_counts * 4));
    HALO_HIP(hipMalloc(&ws.d_tblockoff, 1024 * 4));
    HALO_HIP(hipMalloc(&ws.d_biglist, ws.cap_counts * 4));
    HALO_HIP(hipMalloc(&ws.d_meta, 1024));
    HALO_HIP(hipMalloc(&ws.d_task_g, ws.cap_tasks * 4));
    HALO_HIP(hipMalloc(&ws.d_order, ws.cap_tasks * 16));  // uint4 per task (k_msm_task_order)
    HALO_HIP(hipMalloc(&ws.d_seg, ws.cap_windows * 64 * 2 * XYZZ_WORDS * 4));
    HALO_HIP(hipMalloc(&ws.d_winsum, ws.cap_windows * 12 * 8));
    // (the host reads both while the kernel that writes them is still running: fine-grained coherent, said explicitly)
    HALO_HIP(hipHostMalloc(&ws.h_winsum, ws.cap_windows * 12 * 8, hipHostMallocCoherent));
    HALO_HIP(hipHostMalloc(&ws.h_done, 64, hipHostMallocCoherent));
    *ws.h_done = 0;
    ws.done_expect = 0;
    if (debug_trace())  // address ranges, so that a faulting address can be mapped to a buffer
        fprintf(stderr, "[halo] workspace %p: digits=[%p,+%zu) sorted=[%p,+%zu) presort=[%p,+%zu) partials=[%p,+%zu) hist=[%p,+%zu) counts=%p starts=%p ntask=%p toff=%p task_g=%p order=%p seg=%p winsum=%p\n",
                (void *)&ws, (void *)ws.d_canon, ws.cap_sorted * 2 + 64, (void *)ws.d_sorted, ws.cap_sorted * 4, (void *)ws.d_presort, ws.cap_sorted * 4,
                (void *)ws.d_buckets, ws.cap_tasks * XYZZ_WORDS * 4, (void *)ws.d_hist, ws.cap_hist * 4, (void *)ws.d_counts, (void *)ws.d_starts,
                (void *)ws.d_ntask, (void *)ws.d_toff, (void *)ws.d_task_g, (void *)ws.d_order, (void *)ws.d_seg, (void *)ws.d_winsum);
    return HALO_OK;
}
// capacity for any single MSM of up to n points, whatever the window size
int msm_workspace_alloc(halo_ctx *ctx, size_t n, int slot) {
    if (n < 64) n = 64;
    WorkspaceNeed need;
    need.n = n;
    // sorted entries: n * W; the automatic plan has W <= 32 for n >= 4096 (c >= 8) and W <= 64 below
    need.sorted = n >= 4096 ? n * 32 : n * 64;
    need.counts = max_counts();
    // (a key of the c = 20 table plan: room for the TWO bucket sets of a tagged launch from the start -- growing the workspace at
    // the first tagged launch cost the first open of a context a second of hipFree / hipMalloc)
    if (n >= ((size_t)1 << 20) && need.counts < ((size_t)1 << 20)) need.counts = (size_t)1 << 20;
    {   // tasks: one per non-empty bucket plus entries / kmax (kmax = 16 only below 2^18 points, W <= 32 there)
        size_t small = need.sorted < ((size_t)1 << 23) ? need.sorted : ((size_t)1 << 23);
        size_t extra = need.sorted / KMAX > small / 16 ? need.sorted / KMAX : small / 16;
        need.tasks = need.counts + extra + 1;
    }
    need.hist = (size_t)256 * 32768 + need.counts;  // W * nchunks <= 256 blocks of B <= 32768 counters
    need.windows = 64;
    {   // the table pipeline runs a large MSM in pieces and keeps 2 x 16 window sums per piece
        size_t pieces = (n + TBL_PIECE - 1) / TBL_PIECE;
        if (48 * pieces > need.windows) need.windows = 48 * pieces;  // (or 24 (S, T) pairs per piece: k_msm_reduce_rc)
        if (need.windows < 128) need.windows = 128;                  // (a batch of 8 of the small-key plan: 8 x 8 pairs)
    }
    alloc_epoch_bump(ctx);
    return workspace_alloc(ctx->wss[slot], need);
}
void msm_workspace_free(halo_ctx *ctx) {
    alloc_epoch_bump(ctx);
    table_detach(ctx);
    for (int slot = 0; slot < HALO_SLOTS; ++slot) workspace_release(ctx->wss[slot]);
}

// ------------------------------------------------------------------------------ driver
int msm_run(halo_ctx *ctx, const uint32_t *d_bases, const uint64_t *d_scalars, bool mont, size_t n, host::Point *out) {
    // a multi-device context: a large MSM over its own key goes to the shards (multi.hip); short ones are not worth the fan-out
    if (n >= ((size_t)1 << 16) && multi_takes(ctx, d_bases, n))
        return multi_run(ctx, (size_t)(d_bases - ctx->d_bases) / AFF_STRIDE, n, d_scalars, mont, out);
    BorrowScope scope(ctx);  // synchronous: a large MSM may alternate its pieces over slot 1's workspace (tmsm_enqueue_launches)
    int rc = msm_enqueue(ctx, 0, d_bases, d_scalars, mont, n);
    if (rc) return rc;
    return msm_finish(ctx, 0, out);
}

int msm_enqueue_launches(halo_ctx *ctx, MsmWorkspace &ws, const uint32_t *d_bases, const MsmBatch &members, bool mont, size_t n, int partner);
static int table_build(halo_ctx *ctx);
static bool table_eligible(const halo_ctx *ctx, const uint32_t *d_bases, const MsmBatch &members, size_t n);

struct StreamGuard {  // the launch macro uses ctx->stream
    halo_ctx *ctx;
    hipStream_t saved;
    StreamGuard(halo_ctx *c, hipStream_t s) : ctx(c), saved(c->stream) { c->stream = s; }
    ~StreamGuard() { ctx->stream = saved; }
};

static uint32_t msm_kmax(const halo_ctx *ctx, size_t n) {
    const int small_env = tuning().smsm_kmax;  // development override
    if (ctx->task_len > 0) return (uint32_t)ctx->task_len;
    if (small_env > 0 && n <= ((size_t)1 << 16)) return (uint32_t)small_env;
    const int late_env = tuning().late_kmax;  // development override
    if (n <= ((size_t)1 << 14)) return late_env > 0 ? (uint32_t)late_env : 8u;  // the IPA's late rounds: the chain is the round's latency (measured: 16 -> 8: -0.15 ms per open)
    return n >= ((size_t)1 << 18) ? KMAX : 16u;
}

// what a batch of `count` MSMs of n points needs beyond the slot's current capacity (0 = fits)
// window bits of a launch: the context's forced value, else the caller's hint for these scalars, else the size-based table
static int launch_c(const halo_ctx *ctx, const MsmBatch &members) { return ctx->window_bits > 0 ? ctx->window_bits : members.c_hint; }
static bool batch_need(const halo_ctx *ctx, const MsmWorkspace &ws, size_t n, int count, WorkspaceNeed &need, int c_hint) {
    MsmPlan p = msm_plan(n, ctx->window_bits > 0 ? ctx->window_bits : c_hint);
    size_t Wt = (size_t)p.W * count, total = Wt * p.B, sorted = n * Wt;
    if (count > 1 && ctx->n < ((size_t)1 << 20) && ctx->n >= ((size_t)1 << 17)) {  // room for a batch through the small-key table plan
        size_t sets = 1;
        while (sets < (size_t)count) sets <<= 1;
        if (Wt < 32 * sets) Wt = 32 * sets;          // 2 x 16 window sums per bucket set
        if (sorted < n * 30 * (size_t)count) sorted = n * 30 * (size_t)count;  // 15 rows of u32 digits per member (d_canon holds 2 bytes per entry)
        if (total < sets << 16) total = sets << 16;
    }
    size_t hist = (Wt > 256 ? Wt : 256) * (size_t)p.B;
    size_t tasks = total + sorted / msm_kmax(ctx, n) + 1;
    bool grow = n > ws.cap_n || total > ws.cap_counts || sorted > ws.cap_sorted || hist > ws.cap_hist || Wt > ws.cap_windows ||
                tasks > ws.cap_tasks;
    need.n = n > ws.cap_n ? n : ws.cap_n;
    need.counts = total > ws.cap_counts ? total : ws.cap_counts;
    need.sorted = sorted > ws.cap_sorted ? sorted : ws.cap_sorted;
    need.hist = hist > ws.cap_hist ? hist : ws.cap_hist;
    need.windows = Wt > ws.cap_windows ? Wt : ws.cap_windows;
    need.tasks = tasks > ws.cap_tasks ? tasks : ws.cap_tasks;
    return grow;
}

int msm_enqueue(halo_ctx *ctx, int slot, const uint32_t *d_bases, const uint64_t *d_scalars, bool mont, size_t n) {
    MsmBatch one;
    one.count = 1;
    one.scalars[0] = d_scalars;
    return msm_enqueue_batch(ctx, slot, d_bases, one, mont, n);
}

int msm_enqueue_batch(halo_ctx *ctx, int slot, const uint32_t *d_bases, const MsmBatch &members, bool mont, size_t n) {
    if (slot < 0 || slot >= HALO_SLOTS) { set_error("msm: slot out of range"); return HALO_E_ARG; }
    if (members.count < 1 || members.count > MSM_MAX_BATCH) { set_error("msm: batch size must be in [1, 8]"); return HALO_E_ARG; }
    if (!ctx->wss[slot].d_counts) {
        int rc = msm_workspace_alloc(ctx, ctx->wss[0].cap_n, slot);
        if (rc) return rc;
    }
    MsmWorkspace &ws = ctx->wss[slot];
    if (ws.in_flight) { set_error("msm: slot already has an MSM in flight"); return HALO_E_ARG; }
    ws.plan = MsmPlan{0, 0, 0, msm_outputs(members), 0, 0};
    if (members.tagged && (members.count != 1 || mont || !msm_tagged_ready(ctx, d_bases, n))) {
        set_error("msm: a tagged launch is one canonical scalar array over the context's key with the c = 20 table in place (msm_tagged_ready)");
        return HALO_E_ARG;
    }
    if (members.parts < 1 || members.part < 0 || members.part >= members.parts) { set_error("msm: window shard out of range"); return HALO_E_ARG; }
    if (n == 0) { ws.in_flight = true; return HALO_OK; }
    if (members.parts > 1) {  // a shard that owns no window (more shards than windows) contributes the point at infinity
        MsmPlan p = msm_plan(n, launch_c(ctx, members));
        if (p.W * members.part / members.parts == p.W * (members.part + 1) / members.parts) { ws.in_flight = true; return HALO_OK; }
    }
    {
        // A batch lays the members' windows side by side and a forced task length multiplies the tasks: grow
        // this slot's workspace when the launch needs more room than a single automatic-plan MSM of the
        // context's size (the slot is idle here and its stream is drained).
        MsmPlan p = msm_plan(n, launch_c(ctx, members));
        if ((size_t)p.W * members.count * p.B > ((size_t)1 << 22)) {
            set_error("msm: batch too large for this window size (windows * batch * buckets <= 2^22)");
            return HALO_E_ARG;
        }
        WorkspaceNeed need;
        if ((members.count > 1 || ctx->task_len > 0) && batch_need(ctx, ws, n, members.count, need, members.c_hint)) {
            alloc_epoch_bump(ctx);
            workspace_release(ws);
            int rc = workspace_alloc(ws, need);
            if (rc) return rc;
        }
        if (members.tagged) {  // two sets of 2^19 buckets: per-bucket arrays of 2^20, a task per non-empty bucket and per kmax entries beyond
            const TblPlan tp = ctx->tbl;
            size_t counts = 2 * (size_t)tp.B, kmax = ctx->task_len > 0 ? (size_t)ctx->task_len : KMAX;
            size_t tasks = counts + (size_t)tp.W * n / kmax + 1;
            if (counts > ws.cap_counts || tasks > ws.cap_tasks) {
                need.n = ws.cap_n; need.sorted = ws.cap_sorted; need.hist = ws.cap_hist; need.windows = ws.cap_windows;
                need.counts = counts > ws.cap_counts ? counts : ws.cap_counts;
                need.tasks = tasks > ws.cap_tasks ? tasks : ws.cap_tasks;
                alloc_epoch_bump(ctx);
                workspace_release(ws);
                int rc = workspace_alloc(ws, need);
                if (rc) return rc;
            }
        }
    }
    StreamGuard guard(ctx, ctx->streams[slot]);
    if (!ctx->d_table && table_eligible(ctx, d_bases, members, n)) {
        int rc = table_build(ctx);
        if (rc) return rc;
    }
    // A table MSM of more than TBL_PIECE points runs as consecutive pieces.  Inside a synchronous call (msm_run, pcdl::check: the
    // caller has nothing else in flight) the pieces ALTERNATE over this slot's workspace and stream and the neighbouring slot's:
    // piece k + 1 sorts and accumulates while piece k's window sums -- latency chains on a few hundred waves -- finish.
    int partner = -1;
    if (ctx->may_borrow > 0 && members.count == 1 && ctx->d_table && ctx->tbl.c == 20 && n > TBL_PIECE && table_eligible(ctx, d_bases, members, n) &&
        tuning().piece_alternate) {
        int cand = slot ^ 1;
        if (!ctx->wss[cand].in_flight) {
            if (!ctx->wss[cand].d_counts || ctx->wss[cand].cap_n < ws.cap_n) {
                if (ctx->wss[cand].d_counts) { alloc_epoch_bump(ctx); workspace_release(ctx->wss[cand]); }
                if (msm_workspace_alloc(ctx, ws.cap_n, cand) == HALO_OK) partner = cand;
                else (void)hipGetLastError();  // no room for a second workspace: the pieces run one after the other, same result
            } else partner = cand;
        }
        for (int k = 0; k < 2 && partner >= 0; ++k)
            if (!ctx->ev_piece[slot][k] && hipEventCreateWithFlags(&ctx->ev_piece[slot][k], hipEventDisableTiming) != hipSuccess) partner = -1;
    }
    // The launch sequence below is fixed for a given (bases, scalars, n, form, window): the second
    // time the same key arrives it is captured into a hipGraph, afterwards one graph launch replaces
    // ~25 kernel launches (host launch cost matters for the small MSMs of the IPA rounds and of a
    // rank's share of a sharded MSM).  Event profiling needs the individual launches.
    MsmWorkspace::GraphKey key;
    key.bases = d_bases; key.members = members; key.n = n; key.mont = mont ? 1 : 0; key.c = ctx->window_bits; key.span = ctx->reduce_span + 1024 * ctx->task_len + 65536 * (ctx->sort_two_level + 1) + 262144 * (ctx->small_path + 1) + 1048576 * (ctx->table_mode + 1) + 4194304 * (partner + 1);
    bool graphs = ctx->use_graphs && !ctx->prof.on;
    // A graph is kept while this context has not allocated or freed device memory since it was instantiated (its own
    // workspaces, table, IPA buffers: first use only -- the opens of a loop allocate nothing, so their graphs survive), up to
    // MsmWorkspace::GRAPHS keys per slot.  A replay launches exactly the kernels, grids and arguments a fresh
    // enqueue with this key would.  The GPU memory fault of round 1 ("graph REPLAY ... n=262144") was a kernel with
    // SCRATCH inside a replayed graph (k_msm_reduce1: 256 VGPRs, 12 B/lane of spill) after the queue's scratch had been
    // re-assigned -- not a stale pointer: csrc/check_resources.py now fails the build if any kernel uses scratch.
    // (msm_wait returns when the last kernel has PUBLISHED, which is before the graph's last node has retired: the stream is
    // drained before one of its executable graphs is destroyed -- rare paths, both of them)
    auto drain = [&]() {
        (void)hipStreamSynchronize(ctx->streams[slot]);
        if (partner >= 0) (void)hipStreamSynchronize(ctx->streams[partner]);
    };
    if (ws.graph_epoch != ctx->alloc_epoch) {
        bool any = false;
        for (auto &g : ws.graphs) any = any || g.exec;
        if (any) drain();
        for (auto &g : ws.graphs) {
            if (g.exec) (void)hipGraphExecDestroy(g.exec);
            g = MsmWorkspace::CachedGraph();
        }
        ws.graph_epoch = ctx->alloc_epoch;
    }
    const int cache_n = tuning().graph_cache;  // development switch: HALO_GRAPH_CACHE=1 is the single graph per slot of rounds 1-3
    static_assert(MsmWorkspace::GRAPHS == 8, "tuning.hip clamps HALO_GRAPH_CACHE to 8");
    MsmWorkspace::CachedGraph *hit = nullptr;
    for (int k = 0; k < cache_n; ++k)
        if (ws.graphs[k].exec && key == ws.graphs[k].key) hit = &ws.graphs[k];
    if (graphs && hit) {
        if (debug_trace()) fprintf(stderr, "[halo] graph REPLAY ctx=%p slot=%d n=%zu\n", (void *)ctx, slot, n);
        HALO_HIP(hipGraphLaunch(hit->exec, ctx->streams[slot]));
        hit->used = ++ws.graph_clock;
        ws.plan = hit->plan;
        ws.done_expect += (uint32_t)ws.plan.publishers;
        ws.in_flight = true;
        if (partner >= 0) { ws.borrowed = partner; ctx->wss[partner].in_flight = true; ctx->wss[partner].lent_from = slot; }
        return HALO_OK;
    }
    bool capture = false;
    if (graphs) {
        for (int k = 0; k < cache_n; ++k) capture = capture || key == ws.seen[k];
        if (!capture) { ws.seen[ws.seen_at] = key; ws.seen_at = (ws.seen_at + 1) % cache_n; }
    }
    if (debug_trace()) fprintf(stderr, "[halo] msm enqueue ctx=%p slot=%d n=%zu batch=%d part=%d/%d capture=%d\n", (void *)ctx, slot, n, members.count, members.part, members.parts, (int)capture);
    if (capture) HALO_HIP(hipStreamBeginCapture(ctx->streams[slot], hipStreamCaptureModeRelaxed));
    // The launch's last kernel publishes its sums itself (smsm.hip publish(): pinned buffer + pinned counter) unless the event
    // profiler brackets every launch or HALO_DIRECT_RESULTS=0 asks for the copy + stream wait of rounds 1-3 (development switch).
    const bool direct = tuning().direct_results;
    ctx->sink_done = direct && !ctx->prof.on ? ws.h_done : nullptr;
    ctx->sink_publishers = 0;
    int rc = msm_enqueue_launches(ctx, ws, d_bases, members, mont, n, partner);
    ctx->sink_done = nullptr;
    if (!rc) ws.plan.publishers = ctx->sink_publishers;
    if (capture) {
        hipGraph_t graph = nullptr;
        hipError_t e = hipStreamEndCapture(ctx->streams[slot], &graph);
        if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        if (e != hipSuccess) return hip_fail(e, "hipStreamEndCapture");
        MsmWorkspace::CachedGraph *victim = &ws.graphs[0];
        for (int k = 0; k < cache_n; ++k)
            if (!ws.graphs[k].exec) { victim = &ws.graphs[k]; break; }
            else if (ws.graphs[k].used < victim->used) victim = &ws.graphs[k];
        if (victim->exec) { drain(); (void)hipGraphExecDestroy(victim->exec); *victim = MsmWorkspace::CachedGraph(); }
        e = hipGraphInstantiate(&victim->exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess) { victim->exec = nullptr; return hip_fail(e, "hipGraphInstantiate"); }
        victim->key = key;
        victim->plan = ws.plan;
        victim->used = ++ws.graph_clock;
        HALO_HIP(hipGraphLaunch(victim->exec, ctx->streams[slot]));
    }
    if (rc) {
        // part of the sequence may be on its way and will publish: the counter is brought back in step before anyone waits on it
        (void)hipStreamSynchronize(ctx->streams[slot]);
        if (partner >= 0) (void)hipStreamSynchronize(ctx->streams[partner]);
        (void)hipGetLastError();
        ws.done_expect = *(volatile uint32_t *)ws.h_done;
        return rc;
    }
    ws.done_expect += (uint32_t)ws.plan.publishers;
    ws.in_flight = true;
    if (partner >= 0) { ws.borrowed = partner; ctx->wss[partner].in_flight = true; ctx->wss[partner].lent_from = slot; }
    return HALO_OK;
}

// T[w][i] = 2^(c w) G_i over the whole key, built window by window on the context's first table MSM (one-off: W - 1 passes
// of c doublings and an inversion per point, ~10 ms at n = 2^20)
static int table_build(halo_ctx *ctx) {
    if (ctx->d_table) return HALO_OK;
    // a table that could not be had is tried again after table_backoff more eligible MSMs (64, 128, ... 4096), not never:
    // the memory may have come back, the budget may have been raised
    {   // a clone of this context (halo_ctx_clone) may have built the table already, or be building it right now
        std::lock_guard<std::mutex> lk(ctx->share->mu);
        if (ctx->share->d_table) {
            alloc_epoch_bump(ctx);
            ctx->tbl = ctx->share->tbl;
            ctx->d_table = ctx->share->d_table;
            ctx->table_status = 2;
            return HALO_OK;
        }
        if (ctx->share->table_busy) return HALO_OK;  // (this MSM takes the table-free pipeline; the next one looks again)
        if (ctx->table_calls + 1 < ctx->table_retry_at) { ++ctx->table_calls; return HALO_OK; }
        ctx->share->table_busy = true;
    }
    struct Busy { KeyShare *k; ~Busy() { std::lock_guard<std::mutex> lk(k->mu); k->table_busy = false; } } busy{ctx->share.get()};
    ++ctx->table_calls;
    size_t n = ctx->n;
    TblPlan tp = table_plan(n);
    const size_t bytes = (size_t)tp.W * n * 128;
    auto later = [ctx, bytes](int status, const char *why) {
        ctx->table_status = status;
        ctx->table_retry_at = ctx->table_calls + ctx->table_backoff;
        if (ctx->table_backoff < 4096) ctx->table_backoff *= 2;
        if (!ctx->table_said)
            fprintf(stderr, "[halo] fixed-base table of %zu bytes not built (%s): the table-free pipeline runs, same results (halo_ctx_info 6; tried again later)\n", bytes, why);
        ctx->table_said = true;
    };
    if (!table_budget_reserve(ctx, bytes)) { later(3, "over the budget for optional memory, halo_set_memory_budget"); return HALO_OK; }
    alloc_epoch_bump(ctx);
    // built into a local pointer and published (d_table + tbl together) only after the last step has succeeded: a
    // half-built table is never visible to table_eligible / tmsm_enqueue_piece
    uint32_t *tbl = nullptr;
    hipError_t e = dev_hooks().table_fail ? hipErrorOutOfMemory : hipMalloc(&tbl, bytes);  // (development library's hook: the failure path)
    if (e == hipSuccess) {
        if (debug_trace()) fprintf(stderr, "[halo] table ctx=%p c=%d [%p, +%zu)\n", (void *)ctx, tp.c, (void *)tbl, bytes);
        e = hipMemcpyAsync(tbl, ctx->d_bases, n * 128, hipMemcpyDeviceToDevice, ctx->stream);
        for (int w = 1; w < tp.W && e == hipSuccess; ++w) {
            HALO_LAUNCH(ctx, "k_table_step", k_table_step, dim3((unsigned)(((n + TBL_E - 1) / TBL_E + 255) / 256)), dim3(256), 0,
                        tbl + (size_t)(w - 1) * n * AFF_STRIDE, (uint32_t)n, tp.c, tbl + (size_t)w * n * AFF_STRIDE);
            e = hipGetLastError();
        }
        hipError_t e2 = hipStreamSynchronize(ctx->stream);
        if (e == hipSuccess) e = e2;
    }
    if (e != hipSuccess) {
        // No table, no problem: the general pipeline needs no table memory and gives the same point.
        (void)hipGetLastError();
        if (tbl) (void)hipFree(tbl);
        table_budget_release(ctx, bytes);
        later(4, hipGetErrorString(e));
        return HALO_OK;
    }
    {
        std::lock_guard<std::mutex> lk(ctx->share->mu);
        ctx->share->tbl = tp;
        ctx->share->d_table = tbl;
    }
    ctx->tbl = tp;
    ctx->d_table = tbl;
    ctx->table_status = 2;
    return HALO_OK;
}
// A tagged launch (MsmBatch::tagged) has no table-free form: the caller asks first and keeps its two plain launches otherwise
// (no table yet -- the first MSM over the key builds it --, table mode off, a forced window size, the small-key plan).
bool msm_tagged_ready(const halo_ctx *ctx, const uint32_t *d_bases, size_t n) {
    const bool off = !tuning().tagged;  // development switch: never
    if (off || !ctx->d_table || ctx->tbl.c != 20) return false;
    MsmBatch one;
    one.tagged = true;
    return table_eligible(ctx, d_bases, one, n);
}
// halo_set_table_mode(ctx, 0): the table's memory goes back to the device (its launches have drained: every slot is idle)
int table_release(halo_ctx *ctx) {
    if (!ctx->d_table) return HALO_OK;
    for (int k = 0; k < HALO_SLOTS; ++k) {
        if (ctx->wss[k].in_flight) { set_error("table mode: an MSM is in flight on this context"); return HALO_E_ARG; }
        HALO_HIP(hipStreamSynchronize(ctx->streams[k]));
    }
    alloc_epoch_bump(ctx);  // cached launch graphs name the table
    table_detach(ctx);
    return HALO_OK;
}
// this context stops using the table; the memory goes back when no clone uses it either (each user's view is its own d_table:
// a user that still holds one is counted by looking at the share's other users -- conservatively: freed by the last user of the key)
void table_detach(halo_ctx *ctx) {
    if (!ctx->d_table) return;
    bool free_it = false;
    {
        std::lock_guard<std::mutex> lk(ctx->share->mu);
        if (ctx->share->users == 1 && ctx->share->d_table == ctx->d_table) { ctx->share->d_table = nullptr; ctx->share->tbl = TblPlan{}; free_it = true; }
    }
    if (free_it) {
        (void)hipFree(ctx->d_table);
        table_budget_release(ctx, (size_t)ctx->tbl.W * ctx->n * 128);
    }
    ctx->d_table = nullptr;
    ctx->tbl = TblPlan{};
}
// can this launch take the table pipeline?  One MSM, all windows, over a stretch of the context's own key -- of at least 2^20
// points, or at least half of a smaller key (that plan's coarse ranges are sized for the key) -- indices within 31 bits.
static bool table_eligible(const halo_ctx *ctx, const uint32_t *d_bases, const MsmBatch &members, size_t n) {
    if (ctx->table_mode == 0 || ctx->window_bits != 0 || members.parts != 1) return false;
    TblPlan tp = table_plan(ctx->n);
    if (members.tagged && (tp.c != 20 || n > TBL_PIECE || members.count != 1)) return false;  // (one piece, 2 x 512 coarse ranges)
    if (members.count != 1) {  // batches: small-key plan only, members over the same points, count * ranges coarse ranges at most 512
        uint32_t cpow = 1;
        while ((int)cpow < members.count) cpow <<= 1;
        if (tp.c == 20 || cpow * tp.ranges > TBL_MAX_RANGES) return false;
        for (int b = 1; b < members.count; ++b)
            if (members.base_off[b] != members.base_off[0]) return false;
    }
    size_t least = tp.c == 20 ? ((size_t)1 << 20) : ((size_t)1 << 17);
    if (ctx->n < least || (n < least && !(members.sub && tp.c == 20 && n >= 4096)) || (tp.c != 20 && 2 * n < ctx->n) || n % 4 != 0 || (size_t)tp.W * ctx->n >= ((size_t)1 << 31)) return false;
    return d_bases >= ctx->d_bases && d_bases + AFF_STRIDE * n <= ctx->d_bases + AFF_STRIDE * ctx->n;
}
static int tmsm_enqueue_piece(halo_ctx *ctx, MsmWorkspace &ws, const uint32_t *d_bases, const MsmBatch &members, size_t soff, bool mont, size_t n, int piece,
                              uint64_t *h_dst);
// the shape of the row / column window sums (k_msm_reduce_rc) for a launch of `sets` bucket sets of B buckets each; per = 0: none
static RcShape table_rc_shape(int c, uint32_t B, uint32_t sets) {
    const bool off = !tuning().reduce_rc;  // development switch: the older form
    RcShape r;
    if (off) return r;
    if (c == 20 && (sets == 1 || sets == 2) && B == (1u << 19)) { r.lg_rows = 9; r.lg_cols = 10; r.per = 16; }  // (2 sets: a tagged launch, 2048 waves)
    else if (c == 17 && B == (1u << 16) && (sets == 1 || sets == 2 || sets == 4 || sets == 8)) { r.lg_rows = 8; r.lg_cols = 8; r.per = sets == 1 ? 4 : (int)(2 * sets); }
    return r;
}
// A batch is about throughput: its window sums take 2^15-bucket virtual windows (8 buckets per lane) like the large plan --
// with 2^12 (one bucket per lane: the short chain a single MSM wants) the wave-wide step of k_msm_reduce1 cost as many
// instructions as the bucket kernel itself.
static TblPlan table_launch_plan(const halo_ctx *ctx, int count) {
    TblPlan tp = ctx->tbl;
    if (count > 1 && tp.vw_bits < 15 && tp.B >= (1u << 15)) { tp.vw_bits = 15; tp.vw = tp.B >> 15; }
    return tp;
}
static int tmsm_enqueue_launches(halo_ctx *ctx, MsmWorkspace &ws, const uint32_t *d_bases, const MsmBatch &members, bool mont, size_t n, int partner) {
    const TblPlan tp = table_launch_plan(ctx, members.count);
    size_t pieces = tp.c == 20 ? (n + TBL_PIECE - 1) / TBL_PIECE : 1;
    uint32_t cpow = 1;  // a batch (small-key plan, one piece) lays its members' bucket sets side by side: a power of two of them
    while ((int)cpow < msm_outputs(members)) cpow <<= 1;
    const RcShape rcs = table_rc_shape(tp.c, tp.B, cpow);
    if (2 * tp.vw * pieces * cpow > ws.cap_windows || rc_points(rcs) * cpow * pieces > ws.cap_windows) {
        set_error("msm: table plan exceeds workspace");
        return HALO_E_ARG;
    }
    size_t len = ((n + pieces - 1) / pieces + 3) / 4 * 4, off = 0;
    // `partner` >= 0: the odd pieces run on that slot's workspace and stream (forked off this stream here, joined below) and
    // leave their window sums in THIS slot's pinned buffer, where msm_combine_member adds the pieces up.
    if (pieces < 2) partner = -1;
    int slot = (int)(&ws - ctx->wss);
    hipStream_t mine = ctx->stream;
    if (partner >= 0) {
        if (2 * tp.vw * cpow > ctx->wss[partner].cap_windows || rc_points(rcs) * cpow > ctx->wss[partner].cap_windows) partner = -1;
    }
    if (partner >= 0) {
        HALO_HIP(hipEventRecord(ctx->ev_piece[slot][0], mine));
        HALO_HIP(hipStreamWaitEvent(ctx->streams[partner], ctx->ev_piece[slot][0], 0));
    }
    for (size_t k = 0; k < pieces; ++k, off += len) {
        size_t m = off + len <= n ? len : n - off;  // (n and len are multiples of 4)
        bool alt = partner >= 0 && (k & 1);
        StreamGuard on(ctx, alt ? ctx->streams[partner] : mine);
        int rc = tmsm_enqueue_piece(ctx, alt ? ctx->wss[partner] : ws, d_bases + AFF_STRIDE * off, members, off, mont, m, (int)k, ws.h_winsum);
        if (rc) return rc;
    }
    if (partner >= 0) {
        HALO_HIP(hipEventRecord(ctx->ev_piece[slot][1], ctx->streams[partner]));
        HALO_HIP(hipStreamWaitEvent(mine, ctx->ev_piece[slot][1], 0));
    }
    MsmPlan p;
    p.c = tp.c; p.W = tp.W; p.B = tp.B; p.batch = msm_outputs(members); p.w0 = 0; p.w1 = tp.W; p.table_vw = (int)tp.vw; p.table_vw_bits = tp.vw_bits;
    p.table_pieces = (int)pieces;
    p.table_sets = (int)cpow;
    p.table_rc = rcs.per != 0;
    p.table_rc_lg_rows = rcs.lg_rows;
    p.table_rc_lg_cols = rcs.lg_cols;
    ws.plan = p;
    return HALO_OK;
}
// one piece: window sums to slot `piece` of d_winsum / h_winsum (sets * vw weighted sums, then sets * vw plain sums)
static int tmsm_enqueue_piece(halo_ctx *ctx, MsmWorkspace &ws, const uint32_t *d_bases, const MsmBatch &members, size_t soff, bool mont, size_t n, int piece,
                              uint64_t *h_dst) {
    TblPlan tp = table_launch_plan(ctx, members.count);
    uint32_t cpow = 1;
    while ((int)cpow < msm_outputs(members)) cpow <<= 1;
    size_t entries = (size_t)tp.W * n * members.count;
    if (n > ws.cap_n || entries > ws.cap_sorted || (size_t)tp.B * cpow > ws.cap_counts) { set_error("msm: table plan exceeds workspace"); return HALO_E_ARG; }
    if (!ws.d_fine16) {
        alloc_epoch_bump(ctx);
        HALO_HIP(hipMalloc(&ws.d_fine16, ws.cap_sorted * 2));
    }
    hipStream_t s = ctx->stream;
    uint32_t base_off = (uint32_t)((d_bases - ctx->d_bases) / AFF_STRIDE);
    uint32_t *d_digits = reinterpret_cast<uint32_t *>(ws.d_canon);  // 4 * W n bytes per member <= 2 * cap_sorted
    TblScalars srcs{};
    for (int b = 0; b < members.count; ++b) srcs.p[b] = members.scalars[b] + 4 * soff;
    HALO_LAUNCH(ctx, "k_tmsm_recode", k_tmsm_recode, dim3((unsigned)((n + 255) / 256), (unsigned)members.count), dim3(256), 0, srcs, mont ? 1 : 0,
                members.tagged ? 1 : 0, (uint32_t)n, tp, d_digits, ws.d_meta, ws.d_blockoff, ws.d_tblockoff);
    // from here on: ONE MSM of rows = count * W digit rows over sets * B buckets
    const uint32_t rows = (uint32_t)tp.W * (uint32_t)members.count;
    tp.B *= cpow; tp.ranges *= cpow; tp.vw *= cpow;
    uint32_t nchunks = 256u / rows;  // 19 (17) chunks per window: about one block per CU
    uint32_t chunk_len = (uint32_t)((n + nchunks - 1) / nchunks);
    chunk_len = (chunk_len + 3) / 4 * 4;
    dim3 gridc((unsigned)(rows * nchunks)), b1024(1024), b256(256);
    uint32_t *chist = ws.d_hist, *cstart = ws.d_hist + (size_t)rows * nchunks * tp.ranges;  // <= 255 * 512 + 513 words <= cap_hist
    // chain bound per lane of the bucket kernel: the W n additions over the chip's 2048 x 64 lanes, in one round
    uint32_t kmax = KMAX;
    if (ctx->task_len > 0) kmax = (uint32_t)ctx->task_len;
    else if (entries <= (size_t)16 * 131072) kmax = 16;
    else if (entries <= (size_t)32 * 131072) kmax = 32;
    const bool wide = tp.ranges > TBL_MAX_RANGES;  // (two bucket sets of the c = 20 plan: 1024 coarse ranges)
    if (wide) HALO_LAUNCH(ctx, "k_tmsm_coarse_hist2", k_tmsm_coarse_hist2, gridc, b1024, 0, d_digits, (uint32_t)n, nchunks, chunk_len, tp, chist);
    else HALO_LAUNCH(ctx, "k_tmsm_coarse_hist", k_tmsm_coarse_hist, gridc, b1024, 0, d_digits, (uint32_t)n, nchunks, chunk_len, tp, chist);
    uint32_t *rtotal = cstart + tp.ranges + 1;
    HALO_LAUNCH(ctx, "k_tmsm_scan_chunks", k_tmsm_scan_chunks, dim3(tp.ranges), b256, 0, chist, rows * nchunks, tp.ranges, rtotal);
    HALO_LAUNCH(ctx, "k_tmsm_scan_ranges", k_tmsm_scan_ranges, dim3(1), dim3(tp.ranges), 0, rtotal, cstart);
    if (wide) HALO_LAUNCH(ctx, "k_tmsm_coarse_scatter2", k_tmsm_coarse_scatter2, gridc, b1024, 0, d_digits, (uint32_t)n, nchunks, chunk_len, chist, cstart,
                          (uint32_t)ctx->n, base_off, tp, ws.d_presort, ws.d_fine16);
    else HALO_LAUNCH(ctx, "k_tmsm_coarse_scatter", k_tmsm_coarse_scatter, gridc, b1024, 0, d_digits, (uint32_t)n, nchunks, chunk_len, chist, cstart,
                     (uint32_t)ctx->n, base_off, tp, ws.d_presort, ws.d_fine16);
    HALO_LAUNCH(ctx, "k_tmsm_fine_sort", k_tmsm_fine_sort, dim3(tp.ranges), b1024, TBL_STAGE * 4, ws.d_presort, ws.d_fine16, cstart, kmax, tp, ws.d_counts,
                ws.d_starts, ws.d_ntask, ws.d_toff, ws.d_task_g, ws.d_biglist, ws.d_meta, ws.d_sorted);
    uint32_t total = tp.B;
    size_t max_tasks = (size_t)total + entries / kmax + 1;
    if (max_tasks > ws.cap_tasks) max_tasks = ws.cap_tasks;
    dim3 gridt((unsigned)((max_tasks + 255) / 256));
    HALO_LAUNCH(ctx, "k_msm_task_order", k_msm_task_order, gridt, b256, 0, ws.d_task_g, ws.d_meta, ws.d_sorted, ws.d_starts, ws.d_blockoff, ws.d_counts,
                ws.d_toff, ws.d_tblockoff, kmax, reinterpret_cast<uint4 *>(ws.d_order));
    HALO_LAUNCH(ctx, "k_msm_accumulate", k_msm_accumulate, gridt, b256, 0, ctx->d_table, ws.d_sorted, ws.d_meta, reinterpret_cast<const uint4 *>(ws.d_order),
                ws.d_buckets);
    HALO_LAUNCH(ctx, "k_msm_combine", k_msm_combine, dim3(512 + 1024), dim3(64), 0, ws.d_ntask, ws.d_toff, ws.d_tblockoff, ws.d_meta, ws.d_biglist,
                total, 512u, ws.d_buckets);
    // window sums: the buckets as tp.vw virtual windows of 2^vw_bits, 64 segments each (c = 20: 8 buckets per lane; c = 17: 1)
    const RcShape rcs = table_rc_shape(tp.c, tp.B / cpow, cpow);
    if (rcs.per) {
        // rows and columns of the bucket index (k_msm_reduce_rc): rows + columns entries per set, then blocks of 64 entries ->
        // (S, T) pairs, combined on the host (msm_combine_member)
        const uint32_t pts = rc_points(rcs) * cpow, nb = (tp.B / cpow) / (64u * (uint32_t)rcs.per);
        uint64_t *d_rc = ws.d_winsum + (size_t)piece * pts * 12;
        HALO_LAUNCH(ctx, "k_msm_reduce_rc", k_msm_reduce_rc, dim3(cpow * 2 * nb), dim3(64), 0, ws.d_buckets, ws.d_ntask, ws.d_toff, ws.d_tblockoff, rcs, ws.d_seg);
        uint64_t *h_rc = h_dst + (size_t)piece * pts * 12;
        int rc = rc_mid_enqueue(ctx, ws.d_seg, pts / 2, ctx->sink_done ? h_rc : d_rc, ctx->sink_done, ws.d_meta + 255);
        if (rc) return rc;
        HALO_HIP(hipGetLastError());
        if (!ctx->sink_done) HALO_HIP(hipMemcpyAsync(h_rc, d_rc, (size_t)pts * 96, hipMemcpyDeviceToHost, s));
        return HALO_OK;
    }
    uint64_t *d_out = ws.d_winsum + (size_t)piece * 2 * tp.vw * 12;
    uint32_t vwB = 1u << tp.vw_bits;
    uint32_t L = vwB / 4096 ? vwB / 4096 : 1, nseg = vwB / (64 * L);
    if (tp.vw_bits == 15 && (ctx->reduce_span == 16 || ctx->reduce_span == 32 || ctx->reduce_span == 64)) { L = (uint32_t)ctx->reduce_span; nseg = 512 / L; }
    int logL = 0;
    while ((1u << logL) < L) logL++;
    HALO_LAUNCH(ctx, "k_msm_reduce1", k_msm_reduce1, dim3(tp.vw * nseg), dim3(64), 0, ws.d_buckets, ws.d_ntask, ws.d_toff, ws.d_tblockoff, vwB, L, logL,
                nseg, ws.d_seg);
    {
        uint64_t *out = ctx->sink_done ? h_dst + (size_t)piece * 2 * tp.vw * 12 : d_out;
        int rc = quad_final_enqueue(ctx, ws, tp.vw, nseg, logL + 6, out, out + 12 * tp.vw, nullptr, ctx->sink_done);
        if (rc) return rc;
    }
    HALO_HIP(hipGetLastError());
    if (!ctx->sink_done) HALO_HIP(hipMemcpyAsync(h_dst + (size_t)piece * 2 * tp.vw * 12, d_out, (size_t)2 * tp.vw * 96, hipMemcpyDeviceToHost, s));
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

int msm_finish(halo_ctx *ctx, int slot, host::Point *out) { return msm_finish_batch(ctx, slot, out, 1); }

int msm_wait(halo_ctx *ctx, int slot, int count) {
    if (slot < 0 || slot >= HALO_SLOTS || !ctx->wss[slot].in_flight) { set_error("msm: nothing in flight on this slot"); return HALO_E_ARG; }
    MsmWorkspace &ws = ctx->wss[slot];
    if (ws.plan.batch != count) { set_error("msm: this slot holds a batch of a different size"); return HALO_E_ARG; }
    ws.in_flight = false;
    if (ws.borrowed >= 0) {  // (the neighbour's launches were joined into this slot's stream: the wait below covers them)
        ctx->wss[ws.borrowed].in_flight = false;
        ctx->wss[ws.borrowed].lent_from = -1;
        ws.borrowed = -1;
    }
    if (ws.plan.W == 0) return HALO_OK;  // n == 0, or a window shard without windows
    if (ws.plan.publishers > 0) {
        // the last kernel's blocks count themselves in as they hand over their sums (smsm.hip publish()): poll the pinned counter
        // instead of waiting for the stream -- and look at the stream now and then, in case a launch has died
        // Waiting costs a core only briefly: the first tuning().spin_us microseconds (50) poll with the spin-wait hint -- the late
        // rounds of an open end within that --, then every poll is followed by sched_yield() (free when nobody else wants the
        // core, and a host running many provers shares it), and a launch that outlasts 2 ms (n >= 2^21, a shard's stretch) sleeps
        // 50 us between polls.
        volatile uint32_t *f = ws.h_done;
        const auto t_begin = std::chrono::steady_clock::now();
        const long spin_us = tuning().spin_us;
        int phase = 0;  // 0 spin, 1 yield, 2 sleep
        for (uint32_t spins = 1;; ++spins) {
            if ((int32_t)(*f - ws.done_expect) >= 0) break;
            if ((spins & (phase == 0 ? 0x3fffu : 0xffu)) == 0 || phase == 2) {
                hipError_t e = hipStreamQuery(ctx->streams[slot]);
                if (e == hipSuccess) {
                    if ((int32_t)(*f - ws.done_expect) >= 0) break;
                    ws.done_expect = *f;
                    set_error("msm: the launch sequence ended without publishing its results");
                    return HALO_E_DEVICE;
                }
                if (e != hipErrorNotReady) return hip_fail(e, "hipStreamQuery");
            }
            if (phase == 0) {
                host::cpu_relax();
                if ((spins & 0x3fu) == 0 &&
                    std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_begin).count() >= spin_us) phase = 1;
            } else if (phase == 1) {
                std::this_thread::yield();
                if ((spins & 0x3fu) == 0 &&
                    std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_begin).count() >= 2000) phase = 2;
            } else {
                std::this_thread::sleep_for(std::chrono::microseconds(50));
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        if (ctx->prof.on) HALO_HIP(hipStreamSynchronize(ctx->streams[slot]));
    } else {
        HALO_HIP(hipStreamSynchronize(ctx->streams[slot]));
    }
    bool others = false;
    for (int k = 0; k < HALO_SLOTS; ++k) others = others || ctx->wss[k].in_flight;
    if (ctx->prof.on && !others) ctx->prof.collect();
    return HALO_OK;
}
// one member of a batched launch (any thread: reads the slot's pinned window sums only)
void msm_combine_member(halo_ctx *ctx, int slot, int b, host::Point *out) {
    const MsmWorkspace &ws = ctx->wss[slot];
    MsmPlan p = ws.plan;
    *out = host::Point::infinity();
    if (p.W == 0) return;
    if (p.table_vw > 0 && p.table_rc) {
        // per piece and set: pairs (S_q, T_q) = (sum E, sum (Q + 1) E) over 64 entries each, the columns' first, then the rows'.
        //   sum_lo (lo + 1) C_lo = sum_q T_q + 64 sum_q q S_q;   sum_hi (hi + 1) R_hi likewise
        //   sum_b (b + 1) B_b = [columns] + 2^lg_cols ([rows] - sum_hi R_hi)
        const int cblocks = (1 << p.table_rc_lg_cols) / 64, rblocks = (1 << p.table_rc_lg_rows) / 64;
        const size_t set_pts = 2 * (size_t)(cblocks + rblocks), piece_pts = set_pts * (size_t)p.table_sets;
        auto weighted = [&](int first, int count, host::Point *plain) {
            host::Point t = host::Point::infinity(), run = host::Point::infinity(), tot = host::Point::infinity();
            for (int q = count - 1; q >= 0; --q) {
                host::Point sq = host::Point::infinity();
                for (int k = 0; k < p.table_pieces; ++k) {
                    const uint64_t *w = ws.h_winsum + 12 * ((size_t)k * piece_pts + (size_t)b * set_pts + 2 * (size_t)(first + q));
                    sq = sq + host::Point::load(w);
                    t = t + host::Point::load(w + 12);
                }
                run = run + sq;
                if (q >= 1) tot = tot + run;  // tot = sum_q q S_q by running sums
            }
            for (int k = 0; k < 6 && !tot.is_inf(); ++k) tot = tot.dbl();
            if (plain) *plain = run;
            return t + tot;
        };
        host::Point s_rows;
        host::Point cols = weighted(0, cblocks, nullptr);
        host::Point rows = weighted(cblocks, rblocks, &s_rows) - s_rows;
        for (int k = 0; k < p.table_rc_lg_cols && !rows.is_inf(); ++k) rows = rows.dbl();
        *out = cols + rows;
        return;
    }
    if (p.table_vw > 0) {
        // virtual window v holds the buckets v 2^b + 1 .. (v + 1) 2^b (b = table_vw_bits): sum_v [ T_v + v 2^b S_v ];
        // a piece (large MSM) or a batch stores sets * V weighted sums, then sets * V plain sums; member b owns set b
        int V = p.table_vw, A = p.table_sets * V;
        host::Point acc = host::Point::infinity(), run = host::Point::infinity(), tot = host::Point::infinity();
        for (int k = 0; k < p.table_pieces; ++k)  // the pieces of a large MSM add up window by window
            for (int v = 0; v < V; ++v) acc = acc + host::Point::load(ws.h_winsum + 12 * ((size_t)k * 2 * A + (size_t)b * V + v));
        for (int v = V - 1; v >= 1; --v) {  // tot = sum_v v S_v by running sums
            for (int k = 0; k < p.table_pieces; ++k) run = run + host::Point::load(ws.h_winsum + 12 * ((size_t)k * 2 * A + A + (size_t)b * V + v));
            tot = tot + run;
        }
        for (int k = 0; k < p.table_vw_bits && !tot.is_inf(); ++k) tot = tot.dbl();
        *out = acc + tot;
        return;
    }
    int Wm = p.w1 - p.w0;
    host::Point acc = host::Point::infinity();
    for (int w = Wm - 1; w >= 0; --w) {
        if (!acc.is_inf())
            for (int k = 0; k < p.c; ++k) acc = acc.dbl();
        acc = acc + host::Point::load(ws.h_winsum + 12 * ((size_t)b * Wm + w));
    }
    if (!acc.is_inf())
        for (int k = 0; k < p.c * p.w0; ++k) acc = acc.dbl();  // a window shard's weight 2^(c * w0)
    *out = acc;
}
// Horner over the window sums the slot's last launch left in pinned memory (call after msm_wait)
void msm_combine(halo_ctx *ctx, int slot, host::Point *out, int count) {
    const MsmWorkspace &ws = ctx->wss[slot];
    MsmPlan p = ws.plan;
    for (int b = 0; b < count; ++b) out[b] = host::Point::infinity();
    if (p.W == 0) return;
    for (int b = 0; b < count; ++b) msm_combine_member(ctx, slot, b, &out[b]);
}
int msm_finish_batch(halo_ctx *ctx, int slot, host::Point *out, int count) {
    for (int b = 0; b < count; ++b) out[b] = host::Point::infinity();
    int rc = msm_wait(ctx, slot, count);
    if (rc) return rc;
    msm_combine(ctx, slot, out, count);
    return HALO_OK;
}

}  // namespace halo
