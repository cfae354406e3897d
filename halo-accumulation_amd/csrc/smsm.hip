// Small MSMs (n <= 2^16 points): the 2 lg n L/R products of every pcdl::open (pcdl.rs:203-208), the alpha-weighted
// point_dot of acc.rs:178, commitments of short keys.  At this size an MSM is a latency chain, not a throughput
// problem: the general pipeline of msm.hip spends it in ~22 dependent launches and ~45 serial point additions in
// the window sums (0.33 - 0.6 ms whatever n is).  Here the same Pippenger MSM is 4 launches:
//
//   k_msm_recode        (msm.hip)  signed digits, u16 [window][i]
//   k_smsm_sort         one block per window: counting sort of the window's digits with the whole histogram in
//                       LDS (<= 2^14 counters), point references grouped by bucket, buckets cut into tasks of
//                       <= kmax entries (balanced), task ids reserved with one atomic per window
//   k_smsm_accumulate   one lane per task: XYZZ mixed additions (the same inner loop as k_msm_accumulate)
//   k_smsm_reduce       sum_k k * B_k per window with QUAD-PARALLEL point additions (curve_quad.hpp): a block of
//                       64 quads reduces 64 * L buckets (running sums per quad, suffix scan and tree over the
//                       quads through LDS); the last block of a window to finish combines the window's segments
//                       (ticket counter + __threadfence) and writes the window sum.
// Results are bit-identical to the general pipeline (same digits, same group law; addition order differs, the
// normalised result does not).
#include "curve_quad.hpp"
#include "internal.hpp"

namespace halo {

constexpr uint32_t DIGIT_NONE16 = 0xFFFFu;  // msm.hip: zero digit
constexpr uint32_t SMSM_HEAVY = 4;          // buckets with more partials than this are summed by the whole block first
constexpr uint32_t WAVE_TASK = 512;         // entries one wave task covers at most (8 per lane, then a 6-step shuffle tree)
constexpr uint32_t TASK_IS_WAVE = 0xFFFFFFFFu;

// Tasks of a bucket with c entries.  Up to 4 * kmax entries: lane tasks of <= kmax entries (one lane runs the chain).
// More (the top window, whose few real scalar bits put n / 2^bits entries into each bucket; skewed scalars): wave tasks
// of <= 4096 entries -- 64 lanes run chains of <= 64 and fold them with a shuffle tree -- so that a bucket never leaves
// more than a handful of partials to the window-sum kernel.
HALO_DEV uint32_t bucket_tasks(uint32_t c, uint32_t kmax, uint32_t wave_task) { return c <= 4 * kmax ? (c + kmax - 1) / kmax : (c + wave_task - 1) / wave_task; }


// ------------------------------------------------------------------------------ sort + tasks
// Grid = Wt * R blocks: block (w, r) sorts the buckets [r * RB, (r + 1) * RB) of window w, RB = B / R.  Every block reads
// all n digits of its window (2 n bytes, L2-resident), counts the digits that fall into LOWER ranges in registers and
// those of its own range with LDS atomics: the atomics are what one CU is slow at, so a window is spread over R CUs.
// meta[0] = number of tasks (ids reserved with one atomicAdd per block), meta[1] = number of wave tasks (wt: slot, start,
// length per entry); both zeroed by k_msm_recode.
// The window that holds the scalars' top bits is special: only top_span of its B buckets can be hit at all (the scalar
// has 256 - c (W - 1) bits left) and scalars below the group order only reach the first top_rb * R of them, so its R
// blocks split THAT stretch (top_rb buckets each, the last one takes the rest up to top_span) instead of idling while
// block 0 sorts the whole window.
__global__ __launch_bounds__(1024) void k_smsm_sort(const uint16_t *__restrict__ digits, uint32_t n, uint32_t B, uint32_t R, uint32_t kmax, uint32_t wave_task,
                                                    uint32_t top_w, uint32_t Wm, uint32_t top_span, uint32_t top_rb,
                                                    uint32_t base_off, uint32_t *__restrict__ sorted, uint32_t *__restrict__ bk_first,
                                                    uint32_t *__restrict__ bk_nt, uint32_t *__restrict__ task_rec, uint32_t *__restrict__ wt,
                                                    uint32_t *__restrict__ order, uint32_t *__restrict__ meta) {
    // One private histogram per wave: the top window keeps only a few scalar bits, so all n of its digits fall into ~100
    // buckets, and 16 waves hammering the same LDS words serialise (that block set the kernel's duration).
    extern __shared__ uint32_t lds[];  // hist[16][RB] | scan_a[1024] | scan_b[1024] | misc[2]
    uint32_t RB = B / R;
    uint32_t *hist = lds + (threadIdx.x >> 6) * RB, *sa = lds + 16 * RB, *sb = sa + 1024, *sc = sb + 1024, *misc = sc + 1024, *lbin = misc + 4;  // lbin[0..65): tasks per length, then cursors
    uint32_t w = blockIdx.x / R, r = blockIdx.x % R, tid = threadIdx.x;
    uint32_t lo_b = r * RB, hi_b = lo_b + RB;
    if (w % Wm == top_w) {  // (a batch lays its members' Wm windows side by side: one top window per member)
        lo_b = r * top_rb;
        hi_b = r == R - 1 ? top_span : lo_b + top_rb;
        if (lo_b > top_span) lo_b = top_span;
        if (hi_b > top_span) hi_b = top_span;
        // nothing lands past top_span: the reduce kernel still reads those buckets' task counts
        if (r == R - 1)
            for (uint32_t b = top_span + tid; b < B; b += 1024) { bk_first[w * B + b] = 0; bk_nt[w * B + b] = 0; }
    }
    uint32_t NB = hi_b - lo_b;  // buckets this block owns (<= RB)
    const uint16_t *dg = digits + (size_t)w * n;
#ifdef SMSM_TIMING
    uint64_t tm[8]; int tmi = 0;
#define TMARK() do { __syncthreads(); tm[tmi++] = wall_clock64(); } while (0)
#else
#define TMARK() do {} while (0)
#endif
    TMARK();
    for (uint32_t b = tid; b < 16 * RB; b += 1024) lds[b] = 0;
    if (tid == 0) misc[1] = 0;
    if (tid < 72) lbin[tid] = 0;
    __syncthreads();
    uint32_t below = 0;
    bool vec = (n & 7u) == 0;  // digit rows stay 16-byte aligned: eight digits per load, all loads of a pass in flight at once
    // n <= 2^16: a thread's digits are at most 8 loads of 8 digits, all issued before the first is used (both passes)
    uint4 dq[8];
    if (vec) {
#pragma unroll
        for (int it = 0; it < 8; it++) {
            uint32_t i = 8 * tid + (uint32_t)it * 8192;
            dq[it] = i < n ? *reinterpret_cast<const uint4 *>(dg + i) : make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
        }
#pragma unroll
        for (int it = 0; it < 8; it++) {
            uint32_t v[4] = {dq[it].x, dq[it].y, dq[it].z, dq[it].w};
#pragma unroll
            for (int k = 0; k < 8; k++) {
                uint32_t d = (v[k >> 1] >> (16 * (k & 1))) & 0xFFFFu, b = d & 0x7FFFu;
                if (d != DIGIT_NONE16) {
                    if (b < lo_b) below++;
                    else if (b < hi_b) atomicAdd(&hist[b - lo_b], 1u);
                }
            }
        }
    } else {
        for (uint32_t i = tid; i < n; i += 1024) {
            uint32_t d = dg[i], b = d & 0x7FFFu;
            if (d != DIGIT_NONE16) {
                if (b < lo_b) below++;
                else if (b < hi_b) atomicAdd(&hist[b - lo_b], 1u);
            }
        }
    }
    if (below) atomicAdd(&misc[1], below);
    __syncthreads();
    TMARK();
    // thread t owns the buckets [t * per, (t + 1) * per) of this block's range
    uint32_t per = (NB + 1023) / 1024, b0 = tid * per;
    uint32_t cnt_sum = 0, task_sum = 0, wave_sum = 0;
    for (uint32_t k = 0; k < per; k++) {
        uint32_t b = b0 + k;
        if (b < NB) {
            uint32_t c = 0;
            for (uint32_t wv = 0; wv < 16; wv++) c += lds[wv * RB + b];
            cnt_sum += c;
            uint32_t nt = bucket_tasks(c, kmax, wave_task);
            task_sum += nt;
            if (c > 4 * kmax) wave_sum += nt;
        }
    }
    sa[tid] = cnt_sum;
    sb[tid] = task_sum;
    sc[tid] = wave_sum;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
        uint32_t va = tid >= off ? sa[tid - off] : 0u, vb = tid >= off ? sb[tid - off] : 0u, vc = tid >= off ? sc[tid - off] : 0u;
        __syncthreads();
        sa[tid] += va;
        sb[tid] += vb;
        sc[tid] += vc;
        __syncthreads();
    }
    TMARK();
    if (tid == 1023) {  // one reservation per block: task ids [base, base + total) and wave-task entries
        misc[0] = atomicAdd(&meta[0], sb[1023]);
        misc[2] = sc[1023] ? atomicAdd(&meta[1], sc[1023]) : 0u;
    }
    __syncthreads();
    uint32_t pos = (uint32_t)((size_t)w * n) + misc[1] + sa[tid] - cnt_sum;  // absolute position of the first entry of bucket b0
    uint32_t tfirst = misc[0] + sb[tid] - task_sum;
    uint32_t wfirst = misc[2] + sc[tid] - wave_sum;
    TMARK();
    for (uint32_t k = 0; k < per; k++) {
        uint32_t b = b0 + k;
        if (b >= NB) break;
        uint32_t c = 0;
        for (uint32_t wv = 0; wv < 16; wv++) {  // the waves' counts become their write cursors inside the bucket
            uint32_t t = lds[wv * RB + b];
            lds[wv * RB + b] = pos + c;
            c += t;
        }
        uint32_t nt = bucket_tasks(c, kmax, wave_task);
        uint32_t g = w * B + lo_b + b;
        bk_first[g] = tfirst;
        bk_nt[g] = nt;
        if (nt) {  // balanced split: lengths differ by at most one
            uint32_t q = c / nt, rem = c % nt, p = pos;
            bool wave = c > 4 * kmax;
            uint32_t wi = wfirst;
            if (wave) wfirst += nt;
            for (uint32_t j = 0; j < nt; j++) {
                uint32_t len = q + (j < rem ? 1u : 0u);
                if (wave) {
                    task_rec[tfirst + j] = TASK_IS_WAVE;
                    wt[3 * (wi + j)] = tfirst + j;
                    wt[3 * (wi + j) + 1] = p;
                    wt[3 * (wi + j) + 2] = len;
                } else {
                    task_rec[tfirst + j] = p | ((len - 1) << 26);
                    atomicAdd(&lbin[kmax - len], 1u);
                }
                p += len;
            }
        }
        pos += c;
        tfirst += nt;
    }
    __syncthreads();
    // Execution order of this block's tasks: by decreasing length, so that the 64 lanes of a wave of k_smsm_accumulate run
    // chains of (almost) equal length (bucket sizes are Poisson distributed: a wave otherwise waits for its longest lane).
    // order[base + k] = slot of the k-th task; the wave tasks' slots fill the tail with TASK_IS_WAVE (those lanes leave).
    if (tid == 0) {
        uint32_t run = 0;
        for (uint32_t k = 0; k <= kmax; k++) { uint32_t t = lbin[k]; lbin[k] = run; run += t; }
        misc[3] = run;  // lane tasks of this block
    }
    __syncthreads();
    {
        uint32_t tf = misc[0] + sb[tid] - task_sum;
        for (uint32_t k = 0; k < per; k++) {
            uint32_t b = b0 + k;
            if (b >= NB) break;
            uint32_t g = w * B + lo_b + b, nt = bk_nt[g];
            if (nt && task_rec[tf] != TASK_IS_WAVE) {
                for (uint32_t j = 0; j < nt; j++) {
                    uint32_t len = (task_rec[tf + j] >> 26) + 1;
                    order[misc[0] + atomicAdd(&lbin[kmax - len], 1u)] = tf + j;
                }
            }
            tf += nt;
        }
        for (uint32_t k = misc[3] + tid; k < sb[1023]; k += 1024) order[misc[0] + k] = TASK_IS_WAVE;
    }
    TMARK();
    if (vec) {
#pragma unroll
        for (int it = 0; it < 8; it++) {
            uint32_t i = 8 * tid + (uint32_t)it * 8192;
            uint32_t v[4] = {dq[it].x, dq[it].y, dq[it].z, dq[it].w};
            uint32_t at[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                uint32_t d = (v[k >> 1] >> (16 * (k & 1))) & 0xFFFFu, b = d & 0x7FFFu;
                at[k] = (d != DIGIT_NONE16 && b >= lo_b && b < hi_b) ? atomicAdd(&hist[b - lo_b], 1u) : 0xFFFFFFFFu;
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                uint32_t d = (v[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
                if (at[k] != 0xFFFFFFFFu) sorted[at[k]] = (i + k + base_off) | ((d >> 15) << 31);
            }
        }
    } else {
        for (uint32_t i = tid; i < n; i += 1024) {
            uint32_t d = dg[i], b = d & 0x7FFFu;
            if (d != DIGIT_NONE16 && b >= lo_b && b < hi_b) {
                uint32_t at = atomicAdd(&hist[b - lo_b], 1u);
                sorted[at] = (i + base_off) | ((d >> 15) << 31);
            }
        }
    }
    TMARK();
#ifdef SMSM_TIMING
    if (tid == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - R) && n == 65536)
        printf("sort block %u: zero+pass1 %llu, ownership+scan %llu, reserve %llu, records %llu, pass2 %llu (x10 ns)\n", blockIdx.x,
               (unsigned long long)(tm[1] - tm[0]), (unsigned long long)(tm[2] - tm[1]), (unsigned long long)(tm[3] - tm[2]),
               (unsigned long long)(tm[4] - tm[3]), (unsigned long long)(tm[5] - tm[4]));
#endif
}

// ------------------------------------------------------------------------------ accumulate
// blocks [0, wave_blocks): one wave per wave task (4 per block) -- the long ones start first; the others: one lane per task
__global__ __launch_bounds__(256) void k_smsm_accumulate(const uint32_t *__restrict__ bases, const uint32_t *__restrict__ sorted,
                                                         const uint32_t *__restrict__ task_rec, const uint32_t *__restrict__ wt,
                                                         const uint32_t *__restrict__ order, const uint32_t *__restrict__ meta,
                                                         uint32_t wave_blocks, uint32_t *__restrict__ partial) {
    uint32_t slot, st, cnt;
    bool wave = blockIdx.x < wave_blocks;
    uint32_t lane = threadIdx.x & 63;
    if (!wave) {
        uint32_t t = (blockIdx.x - wave_blocks) * 256 + threadIdx.x;
        if (t >= meta[0]) return;
        slot = order[t];
        if (slot == TASK_IS_WAVE) return;
        uint32_t rec = task_rec[slot];
        st = rec & 0x3FFFFFFu;
        cnt = (rec >> 26) + 1;
    } else {
        uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 6);
        if (i >= meta[1]) return;
        slot = wt[3 * i];
        uint32_t start = wt[3 * i + 1], len = wt[3 * i + 2];
        uint32_t per = (len + 63) / 64, lo = lane * per;
        st = start + lo;
        cnt = lo >= len ? 0u : (len - lo < per ? len - lo : per);
    }
    XyzzN acc = xyzz_inf();
    if (cnt) {
        // index two additions ahead, point one ahead (see k_msm_accumulate: the index -> gather pair is dependent)
        uint32_t e = sorted[st];
        uint32_t last = cnt - 1;
        uint32_t e1 = sorted[st + (1 < last ? 1 : last)];
        AffN nxt = aff_load_signed(bases + AFF_STRIDE * (size_t)(e & 0x7fffffffu), (e >> 31) != 0);
        for (uint32_t k = 0; k < cnt; k++) {
            AffN p = nxt;
            e = e1;
            if (k + 1 < cnt) nxt = aff_load_signed(bases + AFF_STRIDE * (size_t)(e & 0x7fffffffu), (e >> 31) != 0);
            e1 = sorted[st + (k + 2 < last ? k + 2 : last)];
            xyzz_madd(acc, p);
        }
    }
    if (wave) {
#pragma unroll 1
        for (int off = 32; off >= 1; off >>= 1) {
            XyzzN o = xyzz_shfl(acc, (int)((lane + off) & 63));
            if ((int)lane < off) xyzz_add(acc, o);
        }
        if (lane != 0) return;
    }
    xyzz_store(partial + XYZZ_WORDS * (size_t)slot, acc);
}

// ------------------------------------------------------------------------------ window sums
// LDS exchange between the 64 quads of a block: quad Q publishes a point, every quad reads the one `off` quads up
// (infinity past the end).  Lane ql writes coordinate ql; all four lanes read the whole point.
HALO_DEV XyzzN block_shift_down(const XyzzN &p, uint32_t *xch, int Q, int ql, int off, int nquads) {
    __syncthreads();
    xyzz_store_quad(xch + XYZZ_WORDS * Q, p, ql);
    __syncthreads();
    int src = Q + off;
    XyzzN o = xyzz_load(xch + XYZZ_WORDS * (src < nquads ? src : Q));
    return xyzz_select(src < nquads, o, xyzz_inf());
}
// Register relief: a point a quad does not need for a while waits in LDS (a quad add keeps ~200 registers busy; three
// live points next to it spill).  Lane ql stores coordinate ql, all four lanes read the whole point back.
// The four lanes of a quad belong to one wave and LDS operations of a wave complete in order, so no barrier is needed
// between a put and the matching get -- but the compiler must not move a lane's reads of the other lanes' words above
// its own write: hence the fences (they cost an s_waitcnt).
HALO_DEV void park_put(uint32_t *park, int Q, int ql, const XyzzN &p) {
    xyzz_store_quad(park + XYZZ_WORDS * Q, p, ql);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
}
HALO_DEV XyzzN park_get(const uint32_t *park, int Q) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return xyzz_load(park + XYZZ_WORDS * Q);
}

// The last kernel of a launch sequence hands its results to the host itself: the sums are written straight into pinned host
// memory; the writing blocks count themselves in on a device word (zeroed by the launch's recode kernel) and the last of
// `total` adds one to a pinned counter (release, system scope) that msm_wait polls -- no copy kernel behind the last kernel
// and no stream wait on the host.  (One system-scope atomic per launch: one per BLOCK, 52-104 of them on one host word that
// the host is polling, cost 30 us per round.)
HALO_DEV void publish(uint32_t *done, uint32_t *ticket, uint32_t total) {
    if (!done) return;
    __threadfence_system();  // this block's sums are in host memory before it counts itself in
    if (__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u == total) {
        __threadfence_system();
        __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Quad Q holds S (sum of its unit; in registers) and T (its unit's weighted sum, weights 1.. relative to the unit's
// first bucket; parked in parkT); a unit spans 2^k buckets.  Afterwards quad 0 holds S = sum_Q S_Q (returned in S) and
// T = sum_Q (T_Q + Q 2^k S_Q) (returned in T).  `live` = number of quads that hold anything, rounded up to a power of two.
HALO_DEV void block_weighted_sum(XyzzN &S, XyzzN &T, int k, uint32_t *xch, uint32_t *parkS, uint32_t *parkT, int Q, int ql, int live) {
    // inclusive suffix scan of S
#pragma unroll 1
    for (int off = 1; off < live; off <<= 1) {
        XyzzN o = block_shift_down(S, xch, Q, ql, off, live);
        xyzz_add_quad(S, o, ql);
    }
    park_put(parkS, Q, ql, S);
    // sum_{Q >= 1} suffix_Q = sum_Q Q S_Q
    XyzzN V = xyzz_select(Q >= 1, S, xyzz_inf());
#pragma unroll 1
    for (int i = 0; i < k; i++) V = xyzz_dbl_quad(V, ql);
    T = park_get(parkT, Q);
    xyzz_add_quad(T, V, ql);
#pragma unroll 1
    for (int off = live >> 1; off >= 1; off >>= 1) {
        XyzzN o = block_shift_down(T, xch, Q, ql, off, live);
        XyzzN t = T;
        xyzz_add_quad(t, o, ql);
        T = xyzz_select(Q < off, t, T);
    }
    S = park_get(parkS, Q);
}

// grid = Wt * nseg blocks of 64 quads; block (w, s) reduces the buckets [s * 64 L, (s + 1) * 64 L) of window w.
// seg: Wt * nseg records of (S, T); done: one ticket counter per window (zeroed by k_msm_recode).
template <bool FUSED>
__global__ __launch_bounds__(256, 2) void k_smsm_reduce(uint32_t *__restrict__ partial, const uint32_t *__restrict__ bk_first,
                                                        const uint32_t *__restrict__ bk_nt, uint32_t B, uint32_t L, int logL, uint32_t nseg,
                                                        int live_seg, uint32_t *__restrict__ seg, uint32_t *__restrict__ done,
                                                        uint64_t *__restrict__ winsum, uint32_t *__restrict__ published) {
    __shared__ uint32_t xch[64 * XYZZ_WORDS], parkS[64 * XYZZ_WORDS], parkT[64 * XYZZ_WORDS];
    __shared__ uint32_t heavy[64], heavy_n, ticket;
    uint32_t w = blockIdx.x / nseg, s = blockIdx.x % nseg;
    int tid = threadIdx.x, ql = tid & 3, Q = tid >> 2;
    uint32_t first = s * 64 * L + (uint32_t)Q * L;
    if (tid == 0) heavy_n = 0;
    __syncthreads();
    // ---- heavy buckets (more than SMSM_HEAVY partials: skewed scalars only): summed by all 64 quads, result left in
    // the bucket's first partial.  Listed first; at most 64 per block are pre-summed, the rest take the serial loop.
    for (uint32_t j = 0; j < L; j++) {
        uint32_t idx = first + j;
        if (ql == 0 && idx < B && bk_nt[w * B + idx] > SMSM_HEAVY) {
            uint32_t at = atomicAdd(&heavy_n, 1u);
            if (at < 64) heavy[at] = w * B + idx;
        }
    }
    __syncthreads();
    uint32_t nheavy = heavy_n < 64 ? heavy_n : 64;
#pragma unroll 1
    for (uint32_t h = 0; h < nheavy; h++) {
        uint32_t g = heavy[h], nt = bk_nt[g], t0 = bk_first[g];
        XyzzN acc = xyzz_inf();
#pragma unroll 1
        for (uint32_t j = (uint32_t)Q; j < ((nt + 63) & ~63u); j += 64) {
            XyzzN o = xyzz_load(partial + XYZZ_WORDS * (size_t)(t0 + (j < nt ? j : 0)));
            xyzz_add_quad(acc, xyzz_select(j < nt, o, xyzz_inf()), ql);
        }
#pragma unroll 1
        for (int off = 32; off >= 1; off >>= 1) {
            XyzzN o = block_shift_down(acc, xch, Q, ql, off, 64);
            XyzzN t = acc;
            xyzz_add_quad(t, o, ql);
            acc = xyzz_select(Q < off, t, acc);
        }
        if (Q == 0) xyzz_store_quad(partial + XYZZ_WORDS * (size_t)t0, acc, ql);
        __syncthreads();
    }
    // ---- running sums over this quad's L buckets: run in registers, tot parked (only this quad touches its slot)
    XyzzN run = xyzz_inf();
    park_put(parkT, Q, ql, xyzz_inf());
#pragma unroll 1
    for (int j = (int)L - 1; j >= 0; j--) {
        uint32_t idx = first + (uint32_t)j;
        uint32_t nt = 0, t0 = 0;
        if (idx < B) {
            uint32_t g = w * B + idx;
            nt = bk_nt[g];
            t0 = bk_first[g];
            if (nt > SMSM_HEAVY) {  // pre-summed above?
                for (uint32_t h = 0; h < nheavy; h++)
                    if (heavy[h] == g) nt = 1;
            }
        }
        // the wave walks to its largest partial count so that the quad operations stay wave-uniform
        uint32_t ntmax = nt;
#pragma unroll
        for (int off = 32; off >= 4; off >>= 1) {
            uint32_t o = (uint32_t)__shfl_xor((int)ntmax, off, 64);
            ntmax = o > ntmax ? o : ntmax;
        }
        XyzzN b = xyzz_inf();
        if (nt) b = xyzz_load(partial + XYZZ_WORDS * (size_t)t0);
#pragma unroll 1
        for (uint32_t k = 1; k < ntmax; k++) {
            XyzzN o = xyzz_inf();
            if (k < nt) o = xyzz_load(partial + XYZZ_WORDS * (size_t)(t0 + k));
            xyzz_add_quad(b, o, ql);
        }
        xyzz_add_quad(run, b, ql);
        XyzzN tot = park_get(parkT, Q);
        xyzz_add_quad(tot, run, ql);
        park_put(parkT, Q, ql, tot);
    }
    XyzzN tot;
    block_weighted_sum(run, tot, logL, xch, parkS, parkT, Q, ql, 64);
    uint32_t *rec = seg + 2 * XYZZ_WORDS * ((size_t)w * nseg + s);
    if (nseg > 1) {
        if (Q == 0) {
            xyzz_store_quad(rec, run, ql);
            xyzz_store_quad(rec + XYZZ_WORDS, tot, ql);
        }
        if (!FUSED) return;  // k_smsm_final combines the segments
        // the last block of this window to get here combines the window's segments
        __threadfence();
        __syncthreads();
        if (tid == 0) ticket = atomicAdd(&done[w], 1u);
        __syncthreads();
        if (ticket != nseg - 1) return;
        __threadfence();
        XyzzN S = xyzz_inf(), T = xyzz_inf();
        if ((uint32_t)Q < nseg) {
            const uint32_t *o = seg + 2 * XYZZ_WORDS * ((size_t)w * nseg + (uint32_t)Q);
            S = xyzz_load(o);
            T = xyzz_load(o + XYZZ_WORDS);
        }
        park_put(parkT, Q, ql, T);
        block_weighted_sum(S, T, logL + 6, xch, parkS, parkT, Q, ql, live_seg);
        tot = T;
    }
    if (tid == 0) {
        xyzz_store_jac_words(winsum + 12 * (size_t)w, tot);
        publish(published, done + 191, gridDim.x / nseg);  // (done = d_meta + 64: word 255 of the launch's small state)
    }
}

// one block per window: the segments' (S, T) -> the window sum
__global__ __launch_bounds__(256, 2) void k_smsm_final(const uint32_t *__restrict__ seg, uint32_t nseg, int k, int live_seg, uint64_t *__restrict__ winsum,
                                                       uint64_t *__restrict__ winsum_plain, uint32_t *__restrict__ published, uint32_t *__restrict__ ticket) {
    __shared__ uint32_t xch[64 * XYZZ_WORDS], parkS[64 * XYZZ_WORDS], parkT[64 * XYZZ_WORDS];
    uint32_t w = blockIdx.x;
    int tid = threadIdx.x, ql = tid & 3, Q = tid >> 2;
    XyzzN S = xyzz_inf(), T = xyzz_inf();
    if ((uint32_t)Q < nseg) {
        const uint32_t *o = seg + 2 * XYZZ_WORDS * ((size_t)w * nseg + (uint32_t)Q);
        S = xyzz_load(o);
        T = xyzz_load(o + XYZZ_WORDS);
    }
    park_put(parkT, Q, ql, T);
    block_weighted_sum(S, T, k, xch, parkS, parkT, Q, ql, live_seg);
    if (tid == 0) {
        xyzz_store_jac_words(winsum + 12 * (size_t)w, T);
        if (winsum_plain) xyzz_store_jac_words(winsum_plain + 12 * (size_t)w, S);  // the unweighted sum (table pipeline)
        publish(published, ticket, gridDim.x);
    }
}

// ------------------------------------------------------------------------------ launch sequence
// Called from msm_enqueue_launches (msm.hip) after k_msm_recode; digits are in ws.d_canon.  Buffer reuse: d_sorted
// (entries), d_starts (first task of a bucket), d_counts (tasks of a bucket), d_task_g (task records), d_buckets
// (partials), d_seg, d_meta[0] (task count), d_meta[64 + w] (tickets), d_winsum.
int smsm_enqueue(halo_ctx *ctx, MsmWorkspace &ws, const uint32_t *d_bases, uint32_t base_off, size_t n, const MsmPlan &p, uint32_t Wt,
                 uint32_t kmax) {
    const uint16_t *d_digits = reinterpret_cast<const uint16_t *>(ws.d_canon);
    // entries per wave task: 8 per lane where the launch is throughput-bound, 2 per lane for the latency-bound small ones
    const int wt_env = tuning().smsm_wave_task;  // development override
    uint32_t wave_task = wt_env > 0 ? (uint32_t)wt_env : WAVE_TASK;
    // bucket ranges per window: enough blocks that one CU issues at most ~8K LDS atomics per pass
    uint32_t R = 1;
    while (R < 8 && p.B / (2 * R) >= 64 && n / R > 8192) R <<= 1;
    while (p.B / R > 2048) R <<= 1;  // 16 private histograms of B / R counters share the CU's LDS
    size_t lds = (16 * ((size_t)p.B / R) + 3072 + 4 + 72) * 4;
    // the top window (present in this launch unless a window shard stops short of it)
    uint32_t Wm = Wt / (uint32_t)p.batch;  // windows per member
    uint32_t top_w = p.w1 == p.W ? Wm - 1 : 0xFFFFFFFFu, top_span = p.B, top_rb = p.B / R;
    uint32_t spread_bit = 0;
    if (msm_spread(p, &spread_bit)) top_w = 0xFFFFFFFFu;  // the recode has spread the top digits over the whole window: a window like the others
    else {
        int top_bits = 256 - p.c * (p.W - 1);  // scalar bits the last window sees (any 256-bit input)
        if (top_bits < p.c) {
            uint32_t safe = 1u << top_bits;                         // magnitudes 1 .. 2^top_bits
            if (safe < top_span) top_span = safe;
            uint32_t hot = top_bits > 2 ? (1u << (top_bits - 2)) + 2 : top_span;  // scalars below r ~ 2^254: two bits fewer (+ carry)
            if (hot > top_span) hot = top_span;
            top_rb = (hot + R - 1) / R;
            if (top_span - (R - 1) * top_rb > p.B / R) top_rb = (top_span + R - 1) / R;  // the last block's stretch must fit its histogram
        }
    }
    HALO_LAUNCH(ctx, "k_smsm_sort", k_smsm_sort, dim3(Wt * R), dim3(1024), lds, d_digits, (uint32_t)n, p.B, R, kmax, wave_task, top_w, Wm, top_span, top_rb,
                base_off, ws.d_sorted, ws.d_starts, ws.d_counts, ws.d_task_g, ws.d_order, ws.d_biglist, ws.d_meta);
    size_t max_tasks = (size_t)Wt * p.B + n * (size_t)Wt / kmax + 1;
    if (max_tasks > ws.cap_tasks || max_tasks > ws.cap_counts) { set_error("msm: small-path tasks exceed the workspace"); return HALO_E_ARG; }
    // wave tasks: one per bucket of more than 4 kmax entries plus one per further WAVE_TASK entries (3 words each in d_order)
    size_t max_wave = (size_t)Wt * n / (4 * (size_t)kmax) + (size_t)Wt * n / wave_task + 1;
    if (3 * max_wave > ws.cap_tasks) { set_error("msm: small-path wave tasks exceed the workspace"); return HALO_E_ARG; }
    unsigned lane_blocks = (unsigned)((max_tasks + 255) / 256), wave_blocks = (unsigned)((max_wave + 3) / 4);
    HALO_LAUNCH(ctx, "k_smsm_accumulate", k_smsm_accumulate, dim3(lane_blocks + wave_blocks), dim3(256), 0, d_bases, ws.d_sorted, ws.d_task_g,
                ws.d_order, ws.d_biglist, ws.d_meta, wave_blocks, ws.d_buckets);
    // 64 quads per block, L buckets per quad, at most 16 segments per window
    uint32_t L = 1;
    while ((size_t)64 * L * 16 < p.B) L <<= 1;
    {
        // ... and no more blocks than give every SIMD one wave (4 waves per block): with 1.6 waves per SIMD the SIMDs that
        // hold two ran every step of the chain at half speed, and the kernel waits for them
        const int waves_env = tuning().smsm_waves;  // development switch (0: off)
        while (waves_env > 0 && L < 64 && (size_t)Wt * ((p.B + 64 * L - 1) / (64 * L)) * 4 > (size_t)waves_env && (size_t)64 * L < p.B) L <<= 1;
    }
    if (ctx->reduce_span > 0) {
        L = (uint32_t)ctx->reduce_span;
        while ((size_t)64 * L * 64 < p.B) L <<= 1;  // the final stage holds at most 64 segments
    }
    uint32_t nseg = (uint32_t)((p.B + 64 * L - 1) / (64 * L));
    int logL = 0, live_seg = 1;
    while ((1u << logL) < L) logL++;
    while ((uint32_t)live_seg < nseg) live_seg <<= 1;
    // (the fused form's per-window tickets are done[w] = d_meta[64 + w]: d_meta has 256 words, all zeroed by the recode, and word
    // 255 is the publish ticket -- more than 191 windows keep the two-launch form)
    const bool fused = tuning().smsm_fused && Wt <= 191;
    // (one thread per window writes its sum: straight to the slot's pinned buffer when the launch publishes, see publish())
    uint64_t *out = ctx->sink_done ? ws.h_winsum : ws.d_winsum;
    if (fused || nseg == 1) {
        HALO_LAUNCH(ctx, "k_smsm_reduce", k_smsm_reduce<true>, dim3(Wt * nseg), dim3(256), 0, ws.d_buckets, ws.d_starts, ws.d_counts, p.B, L, logL, nseg,
                    live_seg, ws.d_seg, ws.d_meta + 64, out, ctx->sink_done);
    } else {
        HALO_LAUNCH(ctx, "k_smsm_reduce", k_smsm_reduce<false>, dim3(Wt * nseg), dim3(256), 0, ws.d_buckets, ws.d_starts, ws.d_counts, p.B, L, logL, nseg,
                    live_seg, ws.d_seg, ws.d_meta + 64, out, (uint32_t *)nullptr);
        HALO_LAUNCH(ctx, "k_smsm_final", k_smsm_final, dim3(Wt), dim3(256), 0, ws.d_seg, nseg, logL + 6, live_seg, out, (uint64_t *)nullptr, ctx->sink_done, ws.d_meta + 255);
    }
    if (ctx->sink_done) ctx->sink_publishers += 1;
    return HALO_OK;
}

// The last step of the sorted pipelines of msm.hip as well: the (S, T) records k_msm_reduce1 left per segment (nseg <= 64
// per window, 2^k buckets each) -> window sums.  A chain of ~20 dependent point operations run by Wt blocks: the quad
// form takes 80 us where one wave per window took 130.
int quad_final_enqueue(halo_ctx *ctx, MsmWorkspace &ws, uint32_t Wt, uint32_t nseg, int k, uint64_t *winsum, uint64_t *winsum_plain,
                       const uint32_t *seg, uint32_t *done) {
    int live_seg = 1;
    while ((uint32_t)live_seg < nseg) live_seg <<= 1;
    HALO_LAUNCH(ctx, "k_smsm_final", k_smsm_final, dim3(Wt), dim3(256), 0, seg ? seg : ws.d_seg, nseg, k, live_seg, winsum, winsum_plain, done, ws.d_meta + 255);
    if (done) ctx->sink_publishers += 1;
    return HALO_OK;
}

// Last device step of the row / column window sums (msm.hip k_msm_reduce_rc): block w takes 64 consecutive entries E_0..E_63
// (single points) and leaves (S, T) = (sum E_Q, sum (Q + 1) E_Q) as Jacobian words at winsum[2 w], winsum[2 w + 1]: the same
// quad-parallel weighted sum as k_smsm_final, fed with S = T = E.  The 24 pairs are combined on the host (msm_combine_member:
// ~70 additions; a second launch for them was 66 us of latency chain).
__global__ __launch_bounds__(256, 2) void k_rc_mid(const uint32_t *__restrict__ ent, uint64_t *__restrict__ winsum, uint32_t *__restrict__ published, uint32_t *__restrict__ ticket) {
    __shared__ uint32_t xch[64 * XYZZ_WORDS], parkS[64 * XYZZ_WORDS], parkT[64 * XYZZ_WORDS];
    uint32_t w = blockIdx.x;
    int tid = threadIdx.x, ql = tid & 3, Q = tid >> 2;
    XyzzN S = xyzz_load(ent + XYZZ_WORDS * ((size_t)w * 64 + (uint32_t)Q));
    XyzzN T = S;
    park_put(parkT, Q, ql, T);
    block_weighted_sum(S, T, 0, xch, parkS, parkT, Q, ql, 64);
    if (tid == 0) {
        xyzz_store_jac_words(winsum + 24 * (size_t)w, S);
        xyzz_store_jac_words(winsum + 24 * (size_t)w + 12, T);
        publish(published, ticket, gridDim.x);
    }
}
int rc_mid_enqueue(halo_ctx *ctx, const uint32_t *entries, uint32_t blocks, uint64_t *winsum, uint32_t *done, uint32_t *ticket) {
    HALO_LAUNCH(ctx, "k_rc_mid", k_rc_mid, dim3(blocks), dim3(256), 0, entries, winsum, done, ticket);
    if (done) ctx->sink_publishers += 1;
    return HALO_OK;
}

int smsm_prepare() {
    HALO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_smsm_sort), hipFuncAttributeMaxDynamicSharedMemorySize, (16 * 2048 + 3072 + 4 + 72) * 4));
    return HALO_OK;
}

}  // namespace halo
