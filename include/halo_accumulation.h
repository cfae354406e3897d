/* halo_accumulation.h -- C ABI of the MI355X (gfx950) MSM / IPA hot path.
 *
 * Drop-in boundary for rasmus-kirk/halo-accumulation's `group.rs` function set plus the
 * fold loop of `pcdl::open`.  The reference has no FFI seam of its
 * own; each entry point below names the reference code (file:line under code/src/) that a
 * Rust shim would replace with a call to it (the shim is shown in INTEGRATION.md).
 *
 * Data conventions -- exactly what arkworks keeps in memory, so a shim passes `x.0.0`:
 *   scalar (Fr)   4 x u64, little-endian limbs, Montgomery form (R = 2^256)
 *   affine base   8 x u64 = x limbs | y limbs (Fq Montgomery).  (0,0) = point at infinity
 *   Jacobian pt  12 x u64 = X | Y | Z (Fq Montgomery), what `Projective::new_unchecked`
 *                takes (consts.rs:13-21).  Z = 0 = infinity.
 *   Every point WRITTEN by the library is normalised: (x, y, 1), or (1, 1, 0) for infinity,
 *   so outputs are bit-reproducible run to run.
 *
 * Errors: 0 = ok, negative = HALO_E_*; halo_last_error() returns a thread-local message.
 * A shim maps HALO_E_ASSERT to panic! (the reference's assert!, pcdl.rs:102-104,130-132,
 * pedersen.rs:7-12) and HALO_E_REJECT to anyhow::Error (its ensure!, pcdl.rs:261-262,
 * 307-310,339; acc.rs:152-155,169,237-240).
 *
 * Ownership: inputs are borrowed for the call; outputs go to caller buffers; only ctx / ipa
 * handles are library-owned.  A ctx is bound to one HIP device (halo_ctx_create_multi: one per shard) and its streams:
 * calls on the same ctx must not overlap; different ctxs are independent.  No global init.
 * There is NO CPU fallback: without a gfx950 device every compute entry point fails with
 * HALO_E_DEVICE.
 */
#ifndef HALO_ACCUMULATION_H
#define HALO_ACCUMULATION_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HALO_OK 0
#define HALO_E_ASSERT (-1) /* reference would panic (assert!) */
#define HALO_E_REJECT (-2) /* reference would return Err (ensure!) */
#define HALO_E_ARG (-3)    /* bad pointer / size for this library */
#define HALO_E_DEVICE (-4) /* HIP / RCCL failure, or no GPU */

typedef struct halo_ctx halo_ctx;
typedef struct halo_ipa halo_ipa;

const char *halo_last_error(void);
int halo_device_count(void);

/* ---- context: the commitment key, device resident (consts.rs:23-24,68: N, GS) ---------- */
/* Upload n affine bases (n x 8 limbs). */
int halo_ctx_create(int device, const uint64_t *bases_affine, size_t n, halo_ctx **out);
/* Derive G_i = [SHA3-256(genesis || LE64(first_index + i)) mod r] * (-1, 2) on the device
 * (main.rs:18-45; GS[i] of consts.rs is first_index = 2).  Hashing on host, scalar-mul in HIP. */
int halo_ctx_create_urs(int device, uint64_t first_index, size_t n, halo_ctx **out);
/* same with G_j = hash(first_index + j * stride): a rank's cyclic shard of the key (stride = world size) */
int halo_ctx_create_urs_strided(int device, uint64_t first_index, uint64_t stride, size_t n, halo_ctx **out);
/* Multi-device contexts (one process, n_dev GPUs of one node; a device id may repeat).  The handle is a full context on
 * devices[0] over the whole key -- every entry point works on it -- that also owns one shard context per device over that
 * device's index block of the key (block k = [k n / n_dev, (k + 1) n / n_dev), derived or uploaded once).  halo_msm,
 * halo_msm_dev, the begin/end halves of both and the library's own MSMs over the key (commit, check, h_commit) of at least
 * 2^16 points are cut along the block boundaries: each shard runs its stretch on its own device and stream, the partial
 * points are added on the host in block order -- the same normalised point as on one device.  Device-resident scalars
 * are read in place on their own GPU and copied peer-to-peer (xGMI) to the others; host scalars go to each device over its
 * own PCIe link.  halo_msm_dev_batch_begin/_end fan out the same way (every shard runs its stretch of all members as one batched
 * launch); the window-shard form (part / parts != 0 / 1) stays on devices[0]. */
int halo_ctx_create_multi(const int *devices, int n_dev, const uint64_t *bases_affine, size_t n, halo_ctx **out);
int halo_ctx_create_urs_multi(const int *devices, int n_dev, uint64_t first_index, size_t n, halo_ctx **out);
int halo_ctx_devices(const halo_ctx *ctx); /* shards of a multi-device context, 1 for a plain one */
/* A second context over the SAME resident key, for another host thread (the reference is single-threaded; a service that runs
 * several provers side by side is not): it shares `ctx`'s key and -- whichever of the contexts builds them -- the fixed-base MSM
 * table and the fold table (immutable, reference-counted: freed with the last context over the key), and owns its streams,
 * workspaces, scratch and IPA buffers: ~2.3 GB at 2^20 points instead of the 37 GB of an independent context with both
 * tables.  Contexts over one key are as independent as any two contexts: calls on ONE context must not overlap, calls on
 * different ones may (two open + check pairs in flight on a context and its clone: ~14 ms per pair against ~17 ms one at a time).  The
 * tuning knobs are copied at this moment.  Not for multi-device contexts.  halo_set_table_mode(ctx, 0) /
 * halo_set_fold_table(ctx, 0) on one of them stop THAT context's use of the table; the memory goes back when no context
 * over the key is left to use it. */
int halo_ctx_clone(halo_ctx *ctx, halo_ctx **out);
void halo_ctx_destroy(halo_ctx *ctx);
size_t halo_ctx_size(const halo_ctx *ctx);
/* copy bases [off, off+n) back to the host (n x 8 limbs) */
int halo_ctx_read_bases(halo_ctx *ctx, size_t off, size_t n, uint64_t *out_affine);
/* device pointer of the base table and the ctx's hipStream_t (for callers holding device memory) */
void *halo_ctx_bases_dev(halo_ctx *ctx);
void *halo_ctx_stream(halo_ctx *ctx);
/* S = hash(0), H = hash(1) of main.rs:35-45 as normalised Jacobian limbs (host computed) */
int halo_public_points(uint64_t S_out[12], uint64_t H_out[12]);

/* ---- group.rs ------------------------------------------------------------------------- */
/* point_dot_affine (group.rs:24-26): sum_i scalars[i] * GS[off + i], Pippenger in HIP.
 * scalars_are_mont = 1: arkworks' in-memory Fr (Montgomery limbs); 0: plain little-endian integers below 2^255
 * (canonical Fr values; an unreduced value in [r, 2^255) is taken as it is, which gives the same point).
 * `scalars` is host memory (pageable is fine) and is read only during the call.  Synchronous; over the context's fixed-base table
 * an MSM of >= 2^19 points runs as two index stretches on slots 0 and 1 (four on slots 0..3 from 2^21 points on), every later
 * stretch's copy under the earlier ones' kernels (these slots must be idle, else one copy + one launch sequence on slot 0; environment HALO_HOST_SPLIT). */
int halo_msm(halo_ctx *ctx, size_t off, size_t n, const uint64_t *scalars, int scalars_are_mont, uint64_t out_jac[12]);
/* Asynchronous halves of halo_msm (scalars in HOST memory): begin() copies the scalars to the device on the slot's own
 * stream and enqueues the launch sequence behind the copy, end() waits and combines.  With two or more slots a caller
 * overlaps the copy of the next MSM (32 MiB at n = 2^20: PCIe time) with the kernels of the current one.  The scalars must
 * stay untouched until end() returns, whatever memory they live in: begin() only ENQUEUES the copy (hipMemcpyAsync from
 * the caller's pointer; the runtime may pin a pageable range in place and DMA from it later rather than stage it). */
int halo_msm_begin(halo_ctx *ctx, int slot, size_t off, size_t n, const uint64_t *scalars, int scalars_are_mont);
int halo_msm_end(halo_ctx *ctx, int slot, uint64_t out_jac[12]);
/* same, scalars already in device memory (n x 4 limbs, 32-byte aligned device pointer) */
int halo_msm_dev(halo_ctx *ctx, size_t off, size_t n, const void *d_scalars, int scalars_are_mont, uint64_t out_jac[12]);
/* Asynchronous halves of halo_msm_dev, for overlapping independent MSMs: `slot` (0..3) selects
 * one of the context's four workspaces/streams (allocated on first use); begin() only enqueues, end() waits and combines.
 * At most one MSM per slot in flight; the scalars must stay untouched until end(). */
int halo_msm_dev_begin(halo_ctx *ctx, int slot, size_t off, size_t n, const void *d_scalars, int scalars_are_mont);
int halo_msm_dev_end(halo_ctx *ctx, int slot, uint64_t out_jac[12]);
/* Window shard `part` of `parts`: the same MSM restricted to the scalar windows [part*W/parts, (part+1)*W/parts)
 * of the W Pippenger windows, already weighted by its power of two -- the results of all parts (any order,
 * any GPU holding the same bases and scalars) add up to halo_msm_dev's point (halo_point_sum).  A GPU then
 * does 1/parts of the bucket work at the bucket efficiency of the full-size MSM, which index-sharding an
 * n <= 2^21 MSM does not.  end() is halo_msm_dev_end. */
int halo_msm_dev_begin_part(halo_ctx *ctx, int slot, size_t off, size_t n, const void *d_scalars, int scalars_are_mont, int part,
                            int parts);
/* Batched form: `batch` (1..8) independent MSMs over the same bases G[off .. off+n), one resident scalar
 * array each (d_scalars[b]), issued as ONE launch sequence on `slot`: the members' windows go through the
 * sort / bucket / window-sum kernels side by side, which fills the GPU where one launch alone is
 * latency-bound (a rank's window shard of a sharded MSM).  part/parts as in halo_msm_dev_begin_part (0, 1 =
 * whole MSMs).  end() writes batch x 12 limbs; member b's result is bit-identical to the unbatched call on
 * d_scalars[b].  The slot's workspace grows on first use. */
int halo_msm_dev_batch_begin(halo_ctx *ctx, int slot, size_t off, size_t n, const void *const *d_scalars, size_t batch,
                             int scalars_are_mont, int part, int parts);
int halo_msm_dev_batch_end(halo_ctx *ctx, int slot, size_t batch, uint64_t *out_jac);
/* point_dot (group.rs:18-21): arbitrary Jacobian points (m x 12 limbs); they are brought to
 * affine on the device (Montgomery batch inversion: one field inversion per four points; the reference runs m). */
int halo_msm_points(halo_ctx *ctx, const uint64_t *pts_jac, const uint64_t *scalars, size_t m, uint64_t out_jac[12]);
/* point_dot_affine (group.rs:24-26) for bases that are NOT a stretch of the context's key: pedersen::commit
 * (pedersen.rs:6-20) is public API over any `&[PallasAffine]`, and the reference's own test_homomorphism_property
 * (pedersen.rs:30-63) is free to pass other generators.  bases_affine = m x 8 limbs, uploaded for this call
 * ((0,0) = infinity); more generators than the context holds points run as consecutive chunks.  The general
 * (table-free) Pippenger pipeline runs. */
int halo_msm_affine(halo_ctx *ctx, const uint64_t *bases_affine, const uint64_t *scalars, size_t m, int scalars_are_mont,
                    uint64_t out_jac[12]);
/* scalar_dot (group.rs:13-15) */
int halo_scalar_dot(halo_ctx *ctx, const uint64_t *xs, const uint64_t *ys, size_t m, uint64_t out[4]);
/* construct_powers (group.rs:29-37): [1, z, ..., z^(n-1)] */
int halo_powers(halo_ctx *ctx, const uint64_t z[4], size_t n, uint64_t *out);
/* DensePolynomial::evaluate as called at pcdl.rs:135 */
int halo_poly_eval(halo_ctx *ctx, const uint64_t *coeffs, size_t len, const uint64_t z[4], uint64_t out[4]);

/* ---- pcdl.rs: h(X) -------------------------------------------------------------------- */
/* HPoly::get_poly().coeffs (pcdl.rs:56-77): 2^lg_n coefficients from xis[0..lg_n] (xis[0] unused) */
int halo_h_coeffs(halo_ctx *ctx, const uint64_t *xis, size_t lg_n, uint64_t *out);
/* pedersen::commit(None, GS[0..2^lg_n], h.get_poly().coeffs) of pcdl.rs:338, fused on device */
int halo_h_commit(halo_ctx *ctx, const uint64_t *xis, size_t lg_n, uint64_t out_jac[12]);
/* HPoly::eval (pcdl.rs:79-91) for m polynomials at one point (acc.rs:102-104) */
int halo_h_eval_batch(halo_ctx *ctx, const uint64_t *xis, size_t m, size_t lg_n, const uint64_t z[4], uint64_t *out);
/* AccumulatedHPolys::get_poly (acc.rs:85-94): out = h0 (2 coeffs) + sum_i alphas[i] * h_i(X) */
int halo_h_accumulate(halo_ctx *ctx, const uint64_t *h0, const uint64_t *xis, const uint64_t *alphas, size_t m,
                      size_t lg_n, uint64_t *out);

/* ---- pcdl.rs:183-231: the IPA halving loop, state device resident ------------------------ */
/* G = GS[0..n), c = coeffs zero-padded to n, z-powers of z (pcdl.rs:183-186) */
int halo_ipa_begin(halo_ctx *ctx, size_t n, const uint64_t *coeffs, size_t len, const uint64_t z[4], halo_ipa **out);
/* ---- sharded open: rank r of P holds the cyclic shard i = r (mod P) of G, c, z-powers ----
 * local state over n_local = n / P elements with z-vector z^(offset + j * stride) */
int halo_ipa_begin_strided(halo_ctx *ctx, size_t n_local, const uint64_t *coeffs_local, size_t len, const uint64_t z[4],
                           uint64_t stride, uint64_t offset, halo_ipa **out);
/* state from explicit c and z vectors (the last lg P rounds on the gathered P elements) */
int halo_ipa_begin_vectors(halo_ctx *ctx, size_t n, const uint64_t *c_vec, const uint64_t *z_vec, halo_ipa **out);
/* <c, z> of the current state (a shard's share of p(z) before the first round) */
int halo_ipa_dot_cz(halo_ipa *st, uint64_t out[4]);
/* this shard's <c_r, G_l>, <c_l, G_r> (no H' term) and dots = <c_r, z_l> | <c_l, z_r> */
int halo_ipa_round_lr_partial(halo_ipa *st, uint64_t L[12], uint64_t R[12], uint64_t dots[8]);
/* hiding branch (pcdl.rs:137-164): this shard's slice of p_bar = q (X - z), q = `deg` scalars of the stream
 * after rng_state, and its share of <p_bar, G> */
int halo_ipa_hiding_partial(halo_ipa *st, uint64_t rng_state, size_t deg, const uint64_t z[4], uint64_t stride, uint64_t offset,
                            uint64_t Cbar_part[12]);
/* c <- c + alpha p_bar on this shard */
int halo_ipa_apply_hiding(halo_ipa *st, const uint64_t alpha[4]);
/* host step of the hiding branch: C_bar, alpha, w', C' from the gathered parts; advances *rng_state past q and w_bar */
int halo_open_hiding_combine(const uint64_t C[12], const uint64_t z[4], const uint64_t *v_parts, const uint64_t *Cbar_parts, size_t P,
                             const uint64_t w[4], uint64_t *rng_state, size_t deg, uint64_t Cbar[12], uint64_t alpha[4],
                             uint64_t w_prime[4], uint64_t C_prime[12]);
/* halo_ipa_finish plus z[0] */
int halo_ipa_finish_z(halo_ipa *st, uint64_t U[12], uint64_t c[4], uint64_t z0[4]);
/* host steps between collectives: v = sum of the shards' <c, z>, xi_0 = rho_0(C, z, v), H' = xi_0 H (pcdl.rs:135,180-181) */
int halo_open_start(const uint64_t C[12], const uint64_t z[4], const uint64_t *v_parts, size_t P, uint64_t v_out[4], uint64_t xi0[4],
                    uint64_t Hp_out[12]);
/* parts = P records (L 12 | R 12 | dot_l 4 | dot_r 4) in rank order -> L, R with the H' terms, the next
 * challenge rho_0(xi_prev, L, R) and its inverse (pcdl.rs:203-213) */
int halo_open_combine(const uint64_t *parts, size_t P, const uint64_t Hp[12], const uint64_t xi_prev[4], uint64_t L[12], uint64_t R[12],
                      uint64_t xi[4], uint64_t xi_inv[4]);
/* the last lg P rounds of a sharded open, on the host: after lg(n / P) rounds every rank holds ONE element of G, c and z;
 * recs = the P gathered records (G_i Jacobian 12 | c_i 4 | z_i 4) in index order, P a power of two <= 64.  Runs pcdl.rs:195-227
 * over them (challenges chained from xi_prev) and returns the lg P pairs L, R (12 words each, round order), U and c
 * (pcdl.rs:230-231).  Identical on every rank. */
int halo_open_tail(const uint64_t *recs, size_t P, const uint64_t Hp[12], const uint64_t xi_prev[4], uint64_t *Ls, uint64_t *Rs,
                   uint64_t U[12], uint64_t c_out[4]);
/* L = <c_r, G_l> + <c_r, z_l> H', R = <c_l, G_r> + <c_l, z_r> H'   (pcdl.rs:203-208) */
int halo_ipa_round_lr(halo_ipa *st, const uint64_t H_prime[12], uint64_t L[12], uint64_t R[12]);
/* G, c, z folds with the challenge the host hashed from (xi_prev, L, R)   (pcdl.rs:216-224) */
int halo_ipa_round_fold(halo_ipa *st, const uint64_t xi[4], const uint64_t xi_inv[4]);
/* U = G[0], c = c[0]   (pcdl.rs:230-231) */
int halo_ipa_finish(halo_ipa *st, uint64_t U[12], uint64_t c[4]);
void halo_ipa_destroy(halo_ipa *st);
size_t halo_ipa_len(const halo_ipa *st);

/* ---- pcdl / acc level: the reference's public functions, host glue in C++, linear work in HIP
 *
 * Flat layouts (u64 words), lg = log2(d + 1):
 *   EvalProof (pcdl.rs:22-30): [0] hiding flag | [1] lg | Ls lg*12 | Rs lg*12 | U 12 | c 4 | C_bar 12 | w' 4
 *   Instance  (acc.rs:21-28) : C 12 | d 1 | z 4 | v 4 | EvalProof
 *   Accumulator (acc.rs:43-59): Instance fields (C_bar, d, z, v, pi) | h0 8 (two Fr) | U0 12 | w 4
 * `rng_state` is a SplitMix64 state (the reference takes `rng: &mut R`): scalars are 4 draws,
 * little-endian, reduced mod r; the state is advanced exactly as a sequential stream would be. */
size_t halo_proof_words(size_t lg_n);
size_t halo_instance_words(size_t lg_n);
size_t halo_accumulator_words(size_t lg_n);
/* pedersen::commit (pedersen.rs:6-20) over GS[0..n_bases) */
int halo_pedersen_commit(halo_ctx *ctx, const uint64_t *w /*nullable*/, size_t n_bases, const uint64_t *ms, size_t n_ms,
                         uint64_t out[12]);
/* pedersen::commit (pedersen.rs:6-20) over generators the caller passes (n_bases x 8 limbs; not the context's key):
 * the signature takes any `&[PallasAffine]` (pedersen.rs:6), this is that case; halo_msm_affine underneath */
int halo_pedersen_commit_affine(halo_ctx *ctx, const uint64_t *w /*nullable*/, const uint64_t *bases_affine, size_t n_bases,
                                const uint64_t *ms, size_t n_ms, uint64_t out[12]);
/* pcdl::commit (pcdl.rs:99-110) */
int halo_pcdl_commit(halo_ctx *ctx, const uint64_t *coeffs, size_t len, size_t d, const uint64_t *w /*nullable*/, uint64_t out[12]);
/* pcdl::open (pcdl.rs:120-242) */
int halo_pcdl_open(halo_ctx *ctx, uint64_t *rng_state, const uint64_t *coeffs, size_t len, const uint64_t C[12], size_t d,
                   const uint64_t z[4], const uint64_t *w /*nullable*/, uint64_t *proof_out);
/* The same two for a polynomial already resident in device memory (len x 4 limbs, Montgomery; len = p.degree() + 1,
 * i.e. the top coefficient is non-zero as ark-poly's DensePolynomial guarantees).  The coefficients are read, not
 * modified.  No host-to-device transfer of the polynomial: this is the form whose rate bench.py reports. */
int halo_pcdl_commit_dev(halo_ctx *ctx, const void *d_coeffs, size_t len, size_t d, const uint64_t *w /*nullable*/, uint64_t out[12]);
int halo_pcdl_open_dev(halo_ctx *ctx, uint64_t *rng_state, const void *d_coeffs, size_t len, const uint64_t C[12], size_t d,
                       const uint64_t z[4], const uint64_t *w /*nullable*/, uint64_t *proof_out);
/* pcdl::succinct_check (pcdl.rs:252-314): xis_out (lg+1) x 4, U_out */
int halo_pcdl_succinct_check(halo_ctx *ctx, const uint64_t C[12], size_t d, const uint64_t z[4], const uint64_t v[4],
                             const uint64_t *proof, uint64_t *xis_out, uint64_t U_out[12]);
/* m succinct checks at once (acc.rs:158-170 runs them in a loop): instances = m Instance blobs of degree bound d.  From 64
 * instances on the m relations are evaluated in two device launches (h_i(z_i) for all i; the 2 lg n + 2 scalar multiples
 * of every relation, one wave per instance), the transcripts on a pool of host threads.  status[i] (nullable) = 0 or
 * HALO_E_REJECT per instance; xis_out m x (lg+1) x 4 and U_out m x 12 (nullable) as halo_pcdl_succinct_check. */
int halo_pcdl_succinct_check_batch(halo_ctx *ctx, size_t d, const uint64_t *instances, size_t m, uint64_t *xis_out, uint64_t *U_out,
                                   int *status);
/* pcdl::check (pcdl.rs:323-342) */
int halo_pcdl_check(halo_ctx *ctx, const uint64_t C[12], size_t d, const uint64_t z[4], const uint64_t v[4], const uint64_t *proof);
/* One rank's half of pcdl::check over a key sharded cyclically (halo_ctx_create_urs_strided: point i on rank i mod stride,
 * offset = the rank): the succinct check (pcdl.rs:333, the same on every rank; d + 1 may be up to stride * the shard's size)
 * and this rank's share of CM.Commit(ck, h) (pcdl.rs:338) over its own points.  U_out = the proof's U after the succinct
 * check, part_out = the share.  The caller adds the shares of all ranks in rank order (halo_point_sum) and accepts iff the
 * sum equals U (pcdl.rs:339); errors as halo_pcdl_check. */
int halo_pcdl_check_partial(halo_ctx *ctx, const uint64_t C[12], size_t d, const uint64_t z[4], const uint64_t v[4], const uint64_t *proof,
                            uint64_t stride, uint64_t offset, uint64_t U_out[12], uint64_t part_out[12]);

/* ---- sharded open / check in one call each --------------------------------------------------------------
 * The caller's all-gather: every rank contributes `words` u64 from `send`, `recv` receives P x words in rank order; returns
 * 0 on success.  Collectives stay outside the library (RCCL through the host language, MPI, ...): this is the only hook.
 *
 * Failure safety: the number and order of collectives of a call depend only on the arguments all ranks pass alike (P, d,
 * hiding or not).  Every record carries one status word behind its payload (the sizes below are payload + 1): a rank on
 * which something fails between two collectives (device allocation, launch, copy, a per-rank argument) does NOT return --
 * it enters the next collective with its error code there, and all P ranks return that code (the first non-zero one in rank
 * order) at the same collective.  No rank waits in an all-gather its peers never enter.  If the callback itself returns
 * non-zero the fabric has failed: the call returns HALO_E_ARG and the caller must abort its process group. */
typedef int (*halo_allgather_fn)(void *user, const uint64_t *send, size_t words, uint64_t *recv);
/* pcdl::open (pcdl.rs:120-242) with G, c and the z-powers placed cyclically: ctx = this rank's shard of the key
 * (halo_ctx_create_urs_strided with stride = P ranks, first_index + offset), coeffs_local = c[offset], c[offset + P], ...
 * (len_local of them), deg = p.degree() of the WHOLE polynomial (hiding branch only), w = commitment randomness or NULL,
 * *rng_state as halo_pcdl_open (the same on every rank, advanced alike).  Collectives, in order: the shares of p(z) (4 + 1
 * words; hiding: 16 + 1), per round L | R | dot_l | dot_r (32 + 1 words), the P last elements (20 + 1 words, P > 1 only); no
 * vector exchange; the last lg P rounds run on the host from the P gathered elements.  proof_out and v_out = p(z) are the same on
 * every rank and equal what halo_pcdl_open returns on one GPU. */
int halo_pcdl_open_sharded(halo_ctx *ctx, uint64_t stride, uint64_t offset, uint64_t *rng_state, const uint64_t *coeffs_local, size_t len_local,
                           size_t deg, const uint64_t C[12], size_t d, const uint64_t z[4], const uint64_t *w /*nullable*/,
                           halo_allgather_fn allgather, void *user, uint64_t *proof_out, uint64_t v_out[4]);
/* pcdl::check (pcdl.rs:323-342) over the same shards: halo_pcdl_check_partial, one all-gather of 12 + 1 words, the sum compared
 * with U; HALO_E_REJECT -- or the code of a rank whose half failed -- on every rank alike */
int halo_pcdl_check_sharded(halo_ctx *ctx, uint64_t stride, uint64_t offset, const uint64_t C[12], size_t d, const uint64_t z[4],
                            const uint64_t v[4], const uint64_t *proof, halo_allgather_fn allgather, void *user);
/* acc::prover / verifier / decider (acc.rs:190-255); instances = m contiguous Instance blobs */
int halo_acc_prover(halo_ctx *ctx, uint64_t *rng_state, size_t d, const uint64_t *instances, size_t m, uint64_t *acc_out);
int halo_acc_verifier(halo_ctx *ctx, size_t d, const uint64_t *instances, size_t m, const uint64_t *acc);
int halo_acc_decider(halo_ctx *ctx, const uint64_t *acc);
/* benches/acc.rs:15-29 random_instance: the workload generator of the reference's benchmark */
int halo_random_instance(halo_ctx *ctx, uint64_t *rng_state, size_t d, uint64_t *instance_out);

/* ---- wire format (host only; no device needed) ----------------------------------------------
 * EvalProof (pcdl.rs:22-30), Instance (acc.rs:21-28) and Accumulator (acc.rs:43-59) in the byte layout a derived
 * ark-serialize `CanonicalSerialize` (compressed) writes: fields in declaration order; Fr = 32 bytes LE canonical;
 * point = 33 bytes (x LE, byte 32: bit 7 = y is the larger of {y, -y}, bit 6 = infinity); usize = u64 LE; Vec = u64 LE
 * length + elements; Option = 1 byte tag + value.  decode() validates canonical scalars, flag bits and curve membership
 * and returns HALO_E_REJECT on malformed input; blobs are the flat u64 layouts above. */
size_t halo_proof_encoded_size(size_t lg_n, int hiding);
size_t halo_instance_encoded_size(size_t lg_n, int hiding);
size_t halo_accumulator_encoded_size(size_t lg_n); /* upper bound */
int halo_proof_encode(const uint64_t *proof, uint8_t *out, size_t cap, size_t *len);
int halo_proof_decode(const uint8_t *in, size_t len, uint64_t *proof_out, size_t cap_words, size_t *lg_n);
int halo_instance_encode(const uint64_t *instance, uint8_t *out, size_t cap, size_t *len);
int halo_instance_decode(const uint8_t *in, size_t len, uint64_t *instance_out, size_t cap_words, size_t *lg_n);
int halo_accumulator_encode(const uint64_t *acc, uint8_t *out, size_t cap, size_t *len);
int halo_accumulator_decode(const uint8_t *in, size_t len, uint64_t *acc_out, size_t cap_words, size_t *lg_n);

/* ---- measurement and resource policy ------------------------------------------------------ */
/* (Experiment knobs, the primitive test hooks, halo_bench_fr_kernel and the fault injectors are NOT part of this library:
 * include/halo_accumulation_dev.h, libhalo_hip_dev.so.  Environment switches: csrc/tuning.hpp) */
/* on = 1: every kernel launch on this ctx is bracketed by hipEvents on the ctx stream;
 * on = 2: only the dominant kernels (k_msm_accumulate, k_fold_points); 0: off. */
int halo_prof_enable(halo_ctx *ctx, int on);
int halo_prof_reset(halo_ctx *ctx);
/* number of distinct kernels seen; name/total ms/launch count of entry i */
int halo_prof_count(halo_ctx *ctx);
int halo_prof_get(halo_ctx *ctx, int i, const char **name, double *total_ms, long *launches);
/* n uniform scalars (Montgomery limbs) of the SplitMix64 stream, written to DEVICE memory: the
 * synthetic-input generator of the benchmarks (element i = draws 4i+1..4i+4 after *rng_state,
 * little-endian, reduced mod r); *rng_state advances as a sequential stream would */
int halo_rng_scalars_dev(halo_ctx *ctx, uint64_t *rng_state, size_t n, void *d_out);
/* sum of k Jacobian points in index order, on the host (combine step of the sharded MSM) */
int halo_point_sum(const uint64_t *pts_jac, size_t k, uint64_t out[12]);
/* IPA tuning: comb table for the first fold of an open whose size is the context's key (E[w][d][i] = d 64^w G_i: 22
 * windows x 32 multiples x 64 bytes per point of the upper three quarters of the key = 33 KiB x n: 35.4 GB at n = 2^20, plus a
 * 4.6 GB temporary while it is built; ~0.2 s to build; the scalars are split with the curve's endomorphism, 2 x 22 entries per
 * scalar multiple; an open + check at 2^20 takes ~2.6 ms less with it, so the build pays for itself after ~64 opens).
 * -1 (default), contexts of 2^18 .. 2^21 points: nothing happens during the first 7 full-size opens over a key (counted over
 * the context and its clones) -- no memory is requested or reserved; the 8th asks for the table's memory on a helper thread
 * (40 GB at 2^20: 0.5 ms .. 2 s of hipMalloc depending on what the driver has at hand) and the first later open that finds it
 * there builds the table (environment HALO_FOLD_TABLE_AFTER=k replaces the 8; 0 = never automatically).  A caller that knows
 * it will open many times says 1: allocated and built at the next full-size open; 0: never, a table already built
 * (or requested) is released.  The table is OPTIONAL memory and subject to halo_set_memory_budget below: over the budget the
 * generic fold kernel runs (halo_ctx_info(ctx, 5) says so) and the table is considered again later.  Results are identical. */
int halo_set_fold_table(halo_ctx *ctx, int mode);
/* Budget for OPTIONAL device memory -- the MSM fixed-base table (halo_set_table_mode) and the fold table above: memory the
 * library takes on its own to be faster, never to be correct.  The budget is per DEVICE and process-wide: all contexts of
 * this process on ctx's device together never hold more optional memory than `bytes`, and no single request takes more than
 * half of what hipMemGetInfo reports free at that moment.  Default: one sixth of the device's total memory (48 GB of an
 * MI355X's 288 GB: room for the tables of ONE context over 2^20 points, 37.1 GB + the 4.6 GB temporary -- a second context
 * on the same device then runs without a fold table unless the caller raises the budget); environment
 * HALO_MEMORY_BUDGET=<bytes>[K|M|G] replaces the default for the whole process.  0 = no optional memory at all.  Lowering
 * the budget below what is held releases nothing by itself (halo_set_table_mode(ctx, 0) / halo_set_fold_table(ctx, 0) do).
 * A table that was refused or whose allocation failed is tried again later (after 8, 16, ... more opens / 64, 128, ... more
 * MSMs) and at once after this call. */
int halo_set_memory_budget(halo_ctx *ctx, size_t bytes);
/* what: 0 = bytes of the MSM fixed-base table, 1 = bytes of the fold table, 2 = microseconds the fold table took to build,
 * 3 = the budget for optional memory on this context's device, 4 = bytes of it in use (all contexts of this process on
 * that device), 5 = status of the fold table, 6 = status of the MSM table: 0 nothing yet, 1 memory requested, 2 built,
 * 3 over the budget, 4 allocation failed (tried again later), 5 switched off */
size_t halo_ctx_info(const halo_ctx *ctx, int what);
/* MSM tuning: fixed-base tables.  -1 (default): an MSM of n >= 2^20 points over the context's own key uses the table
 * T[w][i] = 2^(20 w) G_i (13 x 128 bytes per point of the key, built on the first such MSM): 13 instead of 16 mixed
 * additions per point and one shared set of 2^19 buckets.  A context whose key has 2^17 .. 2^20 - 1 points builds
 * T[w][i] = 2^(17 w) G_i instead (15 x 128 bytes per point, 15 additions, 2^16 buckets) for MSMs over at least half of
 * its key.  0: never (no table memory).  Optional memory: subject to halo_set_memory_budget; a table that cannot be had
 * (budget, allocation) is not an error -- the table-free pipeline runs (halo_ctx_info(ctx, 6)) and the table is tried again
 * later.  Results are identical. */
int halo_set_table_mode(halo_ctx *ctx, int mode);


#ifdef __cplusplus
}
#endif
#endif
