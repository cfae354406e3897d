/* Development interface of the MI355X MSM / IPA library: libhalo_hip_dev.so (csrc/dev.hip).
 *
 * NOT part of the drop-in boundary (include/halo_accumulation.h, libhalo_hip.so): a production host neither links nor loads
 * this library.  It links against libhalo_hip.so (one copy of the library's state per process) and adds
 *   - the per-context experiment knobs behind the measurements of DESIGN.md / HISTORY.md (every setting gives bit-identical
 *     results; the GPU suite holds them against each other),
 *   - the primitive test hooks of tests/test_gpu_parity.py (one field / group operation per lane),
 *   - halo_bench_fr_kernel for the profiler,
 *   - halo_dev_hook: fault injectors and forced test paths.  The product library has no other way to switch them on: it reads
 *     no HALO_TEST_* variable (csrc/tuning.hpp).
 */
#ifndef HALO_ACCUMULATION_DEV_H
#define HALO_ACCUMULATION_DEV_H
#include "halo_accumulation.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Fault injectors / forced paths, process-wide.  name: "table_fail" (value != 0: the allocation of a fixed-base or fold table
 * reports out-of-memory), "force_peer_copy" (multi-device contexts stage device scalars through a peer copy even on one GPU),
 * "shard_fail_rank" + "shard_fail_at" (the rank with that offset fails locally before collective number `at` of a sharded open:
 * 0 = the share of p(z), 1.. = the rounds, then the tail; at = -2: in a sharded check), "reset" (all off). */
int halo_dev_hook(const char *name, long value);
/* What the library read from the environment at its first use (csrc/tuning.hip), by field: "host_split_set", "host_pieces", "host_split0".."host_split3",
 * "fold_table_after", "graph_cache", "pow_e", "spin_us", "graphs", "memory_budget" (MiB, -1 unset), "trace", "tagged"; -1 for an
 * unknown name.  Host only. */
long halo_dev_tuning(const char *name);

/* `reps` back-to-back launches of one bandwidth-side Fr kernel over n elements of the context's scratch memory (no host
 * round trip in between): which = 0 k_powers, 1 k_poly_eval_partial, 2 k_dot2_partial (one pair of vectors), 3 k_dot2_partial
 * (the two pairs of an IPA round, m = n / 2), 4 k_h_coeffs, 5 k_fold_scalars (m = n / 2), 6 k_axpy.  For rocprofv3 / the
 * event profiler: steady-state kernel durations, the figures of bench.py's hbm_kernels block. */
int halo_bench_fr_kernel(halo_ctx *ctx, int which, size_t n, int reps);

/* replay cached hipGraphs of the MSM launch sequence when the same shape repeats (default on) */
int halo_set_graphs(halo_ctx *ctx, int on);  /* also: environment HALO_GRAPHS=0 at context creation; HALO_TRACE=1 logs every launch */

/* IPA tuning: key size at which halo_ipa_* stops folding G and switches to MSMs over the fixed
 * folded key (default 2^14; 0 or 1 = always fold).  Results are identical either way. */
int halo_set_ipa_switch(halo_ctx *ctx, size_t size);

/* IPA tuning: 2 (default) folds G every other round, two halvings at once with one shared doubling chain, the rounds in
 * between taking L, R from MSMs over the unfolded key; 1 folds G every round.  Results are identical either way. */
int halo_set_fold_levels(halo_ctx *ctx, int levels);

/* IPA tuning: a two-level fold of a key of at most 2^18 points (a latency chain on one wave per SIMD) can run on the context's
 * fourth stream BESIDE the next two rounds, which then take their L, R from the key it reads.  -1 (default): in opens of at
 * most 2^18 points, where it pays (9.3 -> 9.0 ms at 2^18; at 2^20 the rounds over the larger key lose more than the hidden
 * fold returns: 15.7 -> 16.4 ms); 0: every fold in line; 1: wherever possible.  Environment
 * HALO_FOLD_ASYNC=-1/0/1 at context creation.  Results are identical either way. */
int halo_set_fold_async(halo_ctx *ctx, int mode);

/* verifier tuning: 1 (default) = succinct checks of >= 64 instances on the device, 0 = always the host thread pool */
int halo_set_batch_verify(halo_ctx *ctx, int on);

/* MSM tuning: window bits (0 = automatic) */
int halo_set_window_bits(halo_ctx *ctx, int c);

/* MSM tuning: buckets per lane in the window-sum kernel (0 = automatic, else a power of two) */
int halo_set_reduce_span(halo_ctx *ctx, int span);

/* MSM tuning: bucket sort in one pass (0), in two (coarse runs, then a fine sort per run: 1, where the shape allows),
 * or chosen by size (-1, default: two levels from n = 2^17) */
int halo_set_sort_mode(halo_ctx *ctx, int mode);

/* MSM tuning: the 4-launch pipeline for MSMs of up to 2^16 points (sort per window in LDS, quad-parallel window sums):
 * -1 automatic (default), 0 never (the general pipeline at every size).  Results are identical. */
int halo_set_small_path(halo_ctx *ctx, int mode);

/* MSM tuning: longest chain of mixed additions one lane runs in the bucket kernel (0 = automatic; 8, 16, 32, 64) */
int halo_set_task_len(halo_ctx *ctx, int len);

/* ---- primitive hooks used by the parity tests (elementwise over n) ----------------------- */
/* host-only: base-2 expansion of the fold scalar over the Eisenstein units (host_math.hpp glv_digits):
 * out[i] = digit code of 2^i (0 none, 1..3 = +lambda^0..2, 4..6 = -lambda^0..2), *n = number of digits */
int halo_test_glv_digits(const uint64_t xi[4], uint8_t out[144], int *n);
/* host-only: the comb digits of the table fold (foldtab.hip): s = k1 + k2 lambda, out[0..22) = signed base-64 digits of k1
 * (each in [-32, 32]), out[22..44) = those of k2 */
int halo_test_fold_digits(const uint64_t s[4], int8_t out[44]);
int halo_test_field_op(halo_ctx *ctx, int field /*0 Fq, 1 Fr*/, int op /*0 mul,1 add,2 sub,3 inv,4 from_mont,5 to_mont*/,
                       const uint64_t *a, const uint64_t *b, size_t n, uint64_t *out);
/* op 0: jacobian(a) + jacobian(b) via XYZZ add; 1: a + affine b (mixed); 2: double a; 3: a * scalar b (4 limbs);
 * 4, 5, 6: the quad-parallel forms (curve_quad.cuh): a + b, 2a, a + b with every fourth b replaced by a */
int halo_test_point_op(halo_ctx *ctx, int op, const uint64_t *a_jac, const uint64_t *b, size_t n, uint64_t *out_jac);

#ifdef __cplusplus
}
#endif
#endif
