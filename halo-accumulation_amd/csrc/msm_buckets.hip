// What the sorted MSM pipelines share behind their sort: the scans, the task lists, the bucket kernel (k_msm_accumulate, the
// dominant kernel), the combine of multi-task buckets and the window sums (running-sum form k_msm_reduce1, row / column form
// k_msm_reduce_rc).  Launched from msm_general.hip and msm_table.hip (declarations: msm_kernels.hpp).
#include <atomic>
#include <cstring>
#include <thread>

#include "msm_kernels.hpp"

namespace halo {

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

}  // namespace halo
