/* CPU restatement of the halo-accumulation MSM / IPA path -- TEST INFRASTRUCTURE.
 *
 * Nothing in oracle/ is part of the product.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load liborc (built from this file), and only
 * as the checker / the timed CPU baseline ("kind": "port").
 *
 * Data conventions (identical to the product's C ABI, include/halo_accumulation.h):
 *   scalar  (Fr) : 4 x u64 little-endian limbs, Montgomery form, R = 2^256
 *   affine  base : 8 x u64 = x limbs then y limbs (Fq Montgomery); (0,0) = infinity
 *   jacobian pt  : 12 x u64 = X, Y, Z (Fq Montgomery); Z = 0 is infinity
 *   All points WRITTEN by these functions are normalised: (x, y, 1) or (1, 1, 0).
 *
 * Flat proof layout (u64 words), lg = log2(d+1):
 *   [0] hiding flag  [1] lg  [2 ..) Ls lg*12 | Rs lg*12 | U 12 | c 4 | C_bar 12 | w' 4
 * Instance   : C 12 | d 1 | z 4 | v 4 | proof
 * Accumulator: instance | h0 8 (two Fr coefficients) | U0 12 | w 4
 */
#ifndef HALO_ORACLE_CPU_H
#define HALO_ORACLE_CPU_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_PROOF_WORDS(lg) (2u + 24u * (size_t)(lg) + 32u)
#define ORC_INSTANCE_WORDS(lg) (21u + ORC_PROOF_WORDS(lg))
#define ORC_ACC_WORDS(lg) (ORC_INSTANCE_WORDS(lg) + 24u)

typedef struct {
    uint64_t S[12];
    uint64_t H[12];
    const uint64_t *GS; /* N x 8 affine Montgomery */
    size_t N;
} orc_pp;

/* field / encoding helpers */
void orc_fr_to_mont(const uint64_t canon[4], uint64_t out[4]);
void orc_fr_from_mont(const uint64_t mont[4], uint64_t out[4]);
void orc_fq_to_mont(const uint64_t canon[4], uint64_t out[4]);
void orc_fq_from_mont(const uint64_t mont[4], uint64_t out[4]);
void orc_fr_mul(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]);
void orc_fr_add(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]);
int orc_fr_inv(const uint64_t a[4], uint64_t out[4]);
void orc_fq_mul(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]);
/* ns per Fq Montgomery product on this host (four independent chains); out = a value that depends on every product */
double orc_bench_fq_mul(size_t iters, uint64_t out[4]);
/* jacobian -> canonical affine bytes: 32 B x LE, 32 B y LE; returns 1 if infinity */
int orc_point_canonical(const uint64_t jac[12], uint8_t out64[64]);
void orc_point_add(const uint64_t a[12], const uint64_t b[12], uint64_t out[12]);
void orc_point_mul(const uint64_t p[12], const uint64_t k_mont[4], uint64_t out[12]);
void orc_affine_to_jac(const uint64_t aff[8], uint64_t out[12]);

/* SplitMix64 input generator */
uint64_t orc_rng_u64(uint64_t *state);
void orc_rng_scalar(uint64_t *state, uint64_t out_mont[4]);
void orc_rng_scalars(uint64_t *state, size_t n, uint64_t *out_mont);

/* sha3 + transcript (group.rs:41-89) */
void orc_sha3_256(const uint8_t *data, size_t len, uint8_t out[32]);
/* items: kinds[i] = 0 scalar (4 words), 1 jacobian point (12 words); tag 0 = rho_0, 1 = rho_1 */
void orc_rho(int tag, const int *kinds, const uint64_t *const *items, size_t n_items, uint64_t out_mont[4]);

/* main.rs:18-45 */
void orc_urs_scalar(uint64_t index, uint64_t out_mont[4]);
void orc_urs_point(uint64_t index, uint64_t out_jac[12]);
void orc_urs_affine(uint64_t first_index, size_t count, uint64_t *out_affine);
void orc_pp_init(orc_pp *pp, const uint64_t *gs_affine, size_t n);

/* group.rs */
void orc_scalar_dot(const uint64_t *xs, const uint64_t *ys, size_t m, uint64_t out[4]);
void orc_msm_affine(const uint64_t *bases, const uint64_t *scalars, size_t n, uint64_t out[12]);
void orc_msm_jac(const uint64_t *pts, const uint64_t *scalars, size_t m, uint64_t out[12]);
void orc_msm_naive(const uint64_t *bases, const uint64_t *scalars, size_t n, uint64_t out[12]);
void orc_powers(const uint64_t z[4], size_t n, uint64_t *out);

/* pedersen.rs:6-20 ; returns <0 on the reference's assert! */
int orc_pedersen_commit(const orc_pp *pp, const uint64_t *w, const uint64_t *bases, size_t n_bases,
                        const uint64_t *ms, size_t n_ms, uint64_t out[12]);

/* pcdl.rs */
void orc_h_coeffs(const uint64_t *xis, size_t lg_n, uint64_t *out /* 2^lg_n x 4 */);
void orc_h_eval(const uint64_t *xis, size_t lg_n, const uint64_t z[4], uint64_t out[4]);
void orc_poly_eval(const uint64_t *coeffs, size_t len, const uint64_t z[4], uint64_t out[4]);
int orc_pcdl_commit(const orc_pp *pp, const uint64_t *coeffs, size_t len, size_t d, const uint64_t *w, uint64_t out[12]);
int orc_pcdl_open(const orc_pp *pp, uint64_t *rng, const uint64_t *coeffs, size_t len, const uint64_t C[12],
                  size_t d, const uint64_t z[4], const uint64_t *w, uint64_t *proof_out);
int orc_pcdl_succinct_check(const orc_pp *pp, const uint64_t C[12], size_t d, const uint64_t z[4],
                            const uint64_t v[4], const uint64_t *proof, uint64_t *xis_out, uint64_t U_out[12]);
int orc_pcdl_check(const orc_pp *pp, const uint64_t C[12], size_t d, const uint64_t z[4], const uint64_t v[4],
                   const uint64_t *proof);
/* one IPA round of pcdl.rs:195-227 on explicit state (used for kernel-level parity) */
void orc_ipa_round_lr(const uint64_t *gs_jac, const uint64_t *cs, const uint64_t *zs, size_t m,
                      const uint64_t Hp[12], uint64_t L[12], uint64_t R[12]);
void orc_ipa_round_fold(uint64_t *gs_jac, uint64_t *cs, uint64_t *zs, size_t m, const uint64_t xi[4],
                        const uint64_t xi_inv[4]);

/* acc.rs ; instances = m contiguous ORC_INSTANCE_WORDS(lg) blobs */
int orc_acc_prover(const orc_pp *pp, uint64_t *rng, size_t d, const uint64_t *instances, size_t m, uint64_t *acc_out);
int orc_acc_verifier(const orc_pp *pp, size_t d, const uint64_t *instances, size_t m, const uint64_t *acc);
int orc_acc_decider(const orc_pp *pp, const uint64_t *acc);
/* benches/acc.rs:15-29 random_instance */
int orc_random_instance(const orc_pp *pp, uint64_t *rng, size_t d, uint64_t *instance_out);

const char *orc_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
