// Shared between the translation units of the sorted MSM pipelines (msm_general.hip, msm_table.hip, msm_buckets.hip, msm_driver.hip,
// urs.hip): constants, the small argument structs of the kernels, the kernels one unit defines and another launches, and the
// host functions that cross unit boundaries.  (smsm.hip, the small pipeline, shares only internal.hpp.)
#pragma once
#include "curve_quad.hpp"
#include "internal.hpp"

namespace halo {

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

constexpr uint32_t FINE_STAGE = 36864;             // entries staged in LDS by the fine pass: 144 KiB of the CU's 160 KiB
constexpr uint32_t TBL_MAX_RANGES = 512;
// An MSM of more points than this runs as consecutive pieces on its stream (c = 20 plan): a coarse range of a piece then
// holds <= 13 * 1310720 / 512 = 33 k entries and fits the fine sort's LDS stage, buckets keep ~26-33 entries (one task
// each) -- at 2^21 .. 2^24 points in one piece the fine sort placed entries straight to HBM (0.36 ms at 2^21, 8.5 ms at
// 2^24) and k_msm_combine folded 2 .. 8 partials per bucket.  The pieces' window sums are added on the host.
constexpr size_t TBL_PIECE = 1310720;
constexpr uint32_t TDIGIT_NONE = 0xFFFFFFFFu;
constexpr int TBL_E = 4;  // points per lane of the kernels that share one inversion (k_table_step, k_batch_to_affine)
constexpr uint32_t TBL_STAGE = 35840;  // entries staged in LDS: 140 KiB next to 20 KiB of counters

struct MemberScalars { const uint64_t *p[MSM_MAX_BATCH]; };
struct MemberOffsets { uint32_t v[MSM_MAX_BATCH]; };
struct TblScalars { const uint64_t *p[MSM_MAX_BATCH]; };
struct RcShape { int lg_rows = 0, lg_cols = 0, per = 0; };
static inline uint32_t rc_points(const RcShape &r) { return r.per ? 2u * ((1u << r.lg_rows) + (1u << r.lg_cols)) / 64u : 0u; }  // (S, T) pairs x 2, per set
HALO_DEV uint32_t scan_at(const uint32_t *__restrict__ in_block, const uint32_t *__restrict__ blockoff, uint32_t g) {
    return in_block[g] + blockoff[g >> 12];
}

// ---- msm_buckets.hip: everything behind the sort, common to the general and the table pipeline
__global__ __launch_bounds__(256) void k_scan_blocks(const uint32_t *__restrict__ in, uint32_t total, uint32_t *__restrict__ out,
                                                     uint32_t *__restrict__ blocksum);
__global__ __launch_bounds__(1024) void k_scan_top(uint32_t *blocksum, uint32_t nblocks);
__global__ __launch_bounds__(256) void k_msm_task_bins(const uint32_t *__restrict__ ntask, const uint32_t *__restrict__ toff,
                                                       const uint32_t *__restrict__ tblockoff, const uint32_t *__restrict__ counts,
                                                       uint32_t total_buckets, uint32_t kmax, uint32_t *__restrict__ meta,
                                                       uint32_t *__restrict__ task_g, uint32_t *__restrict__ biglist);
__global__ __launch_bounds__(256) void k_msm_task_order(const uint32_t *__restrict__ task_g, uint32_t *__restrict__ meta,
                                                        const uint32_t *__restrict__ sorted, const uint32_t *__restrict__ starts,
                                                        const uint32_t *__restrict__ blockoff, const uint32_t *__restrict__ counts,
                                                        const uint32_t *__restrict__ toff, const uint32_t *__restrict__ tblockoff,
                                                        uint32_t kmax, uint4 *__restrict__ order);
__global__ __launch_bounds__(256) void k_msm_accumulate(const uint32_t *__restrict__ bases, const uint32_t *__restrict__ sorted,
                                                        const uint32_t *__restrict__ meta, const uint4 *__restrict__ order,
                                                        uint32_t *__restrict__ partial);
__global__ __launch_bounds__(64) void k_msm_combine(const uint32_t *__restrict__ ntask, const uint32_t *__restrict__ toff,
                                                    const uint32_t *__restrict__ tblockoff, const uint32_t *__restrict__ meta,
                                                    const uint32_t *__restrict__ biglist, uint32_t total, uint32_t small_blocks,
                                                    uint32_t *__restrict__ partial);
__global__ __launch_bounds__(64) void k_msm_reduce1(const uint32_t *__restrict__ partial, const uint32_t *__restrict__ ntask,
                                                    const uint32_t *__restrict__ toff, const uint32_t *__restrict__ tblockoff, uint32_t B,
                                                    uint32_t L, int logL, uint32_t nseg, uint32_t *__restrict__ seg);
__global__ __launch_bounds__(64) void k_msm_reduce_rc(const uint32_t *__restrict__ partial, const uint32_t *__restrict__ ntask,
                                                      const uint32_t *__restrict__ toff, const uint32_t *__restrict__ tblockoff,
                                                      RcShape sh, uint32_t *__restrict__ ent);

// ---- host functions that cross unit boundaries
struct StreamGuard {  // the launch macro uses ctx->stream
    halo_ctx *ctx;
    hipStream_t saved;
    StreamGuard(halo_ctx *c, hipStream_t s) : ctx(c), saved(c->stream) { c->stream = s; }
    ~StreamGuard() { ctx->stream = saved; }
};
uint32_t msm_kmax(const halo_ctx *ctx, size_t n);                       // msm_driver.hip: task length of the bucket kernel for an MSM of n points
int launch_c(const halo_ctx *ctx, const MsmBatch &members);            // msm_driver.hip: window bits of a launch
int msm_enqueue_launches(halo_ctx *ctx, MsmWorkspace &ws, const uint32_t *d_bases, const MsmBatch &members, bool mont, size_t n, int partner);  // msm_general.hip
int msm_general_prepare();                                              // msm_general.hip: dynamic LDS sizes of its sort kernels
int table_build(halo_ctx *ctx);                                         // msm_table.hip
bool table_eligible(const halo_ctx *ctx, const uint32_t *d_bases, const MsmBatch &members, size_t n);
int tmsm_enqueue_launches(halo_ctx *ctx, MsmWorkspace &ws, const uint32_t *d_bases, const MsmBatch &members, bool mont, size_t n, int partner);
int msm_table_prepare();

}  // namespace halo
