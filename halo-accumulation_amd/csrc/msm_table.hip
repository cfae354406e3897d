// The fixed-base-table pipeline of the MSM (K1/K2 over the context's own key): table plans, the table's build, its recode and
// two-level sort kernels and its launch sequence in pieces.  Behind the sort it shares msm_buckets.hip with the general pipeline.
#include <atomic>
#include <cstring>
#include <thread>

#include "msm_kernels.hpp"

namespace halo {

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

int msm_table_prepare() {
    HALO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_tmsm_fine_sort), hipFuncAttributeMaxDynamicSharedMemorySize, TBL_STAGE * 4));
    return HALO_OK;
}

static int tmsm_enqueue_piece(halo_ctx *ctx, MsmWorkspace &ws, const uint32_t *d_bases, const MsmBatch &members, size_t soff, bool mont, size_t n, int piece,
                              uint64_t *h_dst);
// T[w][i] = 2^(c w) G_i over the whole key, built window by window on the context's first table MSM (one-off: W - 1 passes
// of c doublings and an inversion per point, ~10 ms at n = 2^20)
int table_build(halo_ctx *ctx) {
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
bool table_eligible(const halo_ctx *ctx, const uint32_t *d_bases, const MsmBatch &members, size_t n) {
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
int tmsm_enqueue_launches(halo_ctx *ctx, MsmWorkspace &ws, const uint32_t *d_bases, const MsmBatch &members, bool mont, size_t n, int partner) {
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

}  // namespace halo
