/* integration/harness.c -- the call sequence of the patched pcdl::open (integration/pcdl_rs.patch, INTEGRATION.md
 * section 3) and of the patched AccumulatedHPolys (integration/acc_rs.patch), written in plain C against include/halo_accumulation.h and compared with the library's own
 * halo_pcdl_open / verified with halo_pcdl_check.  What Rust keeps doing in the shim -- rho_0! and the inverse
 * (pcdl.rs:212-213) -- is done here by halo_open_start / halo_open_combine (the host steps the library exports for the
 * sharded open; with P = 1 they reduce to exactly that).
 *
 *   gcc -O2 -I../include harness.c -o harness -L../halo-accumulation_amd -lhalo_hip -Wl,-rpath,$PWD/../halo-accumulation_amd
 *   ./harness [lg_n = 12]        exit code 0 = both proofs identical and accepted
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "halo_accumulation.h"

#define CK(expr)                                                                    \
    do {                                                                            \
        int rc_ = (expr);                                                           \
        if (rc_ != HALO_OK) {                                                       \
            fprintf(stderr, "%s -> %d: %s\n", #expr, rc_, halo_last_error());       \
            return 2;                                                               \
        }                                                                           \
    } while (0)

int main(int argc, char **argv) {
    size_t lg = argc > 1 ? (size_t)atoi(argv[1]) : 12, n = (size_t)1 << lg, d = n - 1;
    halo_ctx *ctx = NULL;
    CK(halo_ctx_create_urs(0, 2, n, &ctx)); /* GS[0..n) by the main.rs:18-45 rule */

    /* a polynomial of degree d - 3 and an evaluation point from the library's input generator (SplitMix64) */
    size_t len = n - 3;
    uint64_t *coeffs = malloc((n + 1) * 32), z[4];
    {   /* rng_scalars writes device memory: the base table's pointer is not ours to use, so go through halo_powers-free
           host generation: halo_h_coeffs of random-looking challenges gives n arbitrary scalars without any device pointer */
        uint64_t xis[64 * 4];
        memset(xis, 0, sizeof xis);
        for (size_t i = 0; i <= lg; i++) { xis[4 * i] = 0x9E3779B97F4A7C15ull * (i + 1); xis[4 * i + 1] = 0xBF58476D1CE4E5B9ull ^ i; xis[4 * i + 3] = 0x1234567ull + i; }
        CK(halo_h_coeffs(ctx, xis, lg, coeffs));
        memcpy(z, coeffs + 4 * 5, 32);
    }
    uint64_t C[12], v[4];
    CK(halo_pcdl_commit(ctx, coeffs, len, d, NULL, C));
    CK(halo_poly_eval(ctx, coeffs, len, z, v));

    /* (a) the library's pcdl::open */
    size_t words = halo_proof_words(lg);
    uint64_t *pa = calloc(words, 8), *pb = calloc(words, 8), rng = 1;
    CK(halo_pcdl_open(ctx, &rng, coeffs, len, C, d, z, NULL, pa));

    /* (b) the shim's loop (non-hiding branch, pcdl.rs:165-231) */
    uint64_t v2[4], xi[4], Hp[12];
    CK(halo_open_start(C, z, v, 1, v2, xi, Hp)); /* xi_0 = rho_0(C, z, v), H' = xi_0 H */
    halo_ipa *st = NULL;
    CK(halo_ipa_begin(ctx, n, coeffs, len, z, &st));
    pb[0] = 0;
    pb[1] = lg;
    for (size_t round = 0; round < lg; round++) {
        uint64_t rec[32], L[12], R[12], xi_next[4], xi_inv[4];
        CK(halo_ipa_round_lr_partial(st, rec, rec + 12, rec + 24));       /* <c_r, G_l>, <c_l, G_r>, the two dots */
        CK(halo_open_combine(rec, 1, Hp, xi, L, R, xi_next, xi_inv));      /* + H' terms, rho_0(xi, L, R), inverse */
        memcpy(pb + 2 + 12 * round, L, 96);
        memcpy(pb + 2 + 12 * lg + 12 * round, R, 96);
        memcpy(xi, xi_next, 32);
        CK(halo_ipa_round_fold(st, xi_next, xi_inv));
    }
    uint64_t *tail = pb + 2 + 24 * lg;
    CK(halo_ipa_finish(st, tail, tail + 12));
    halo_ipa_destroy(st);
    memcpy(tail + 16, pa + 2 + 24 * lg + 16, 96); /* C_bar = None: the library's encoding of the point at infinity */

    int same = memcmp(pa, pb, words * 8) == 0;
    int ok_a = halo_pcdl_check(ctx, C, d, z, v, pa), ok_b = halo_pcdl_check(ctx, C, d, z, v, pb);
    /* and the wire format round trip of the shim's proof */
    size_t cap = halo_proof_encoded_size(lg, 0), blen = 0, lg_back = 0;
    uint8_t *bytes = malloc(cap);
    uint64_t *pc = calloc(words, 8);
    CK(halo_proof_encode(pb, bytes, cap, &blen));
    CK(halo_proof_decode(bytes, blen, pc, words, &lg_back));
    int wire = blen == cap && lg_back == lg && memcmp(pb, pc, words * 8) == 0;
    /* the shim's point_dot_affine (ffi.rs) has two routes: a slice of the static key by (offset, length) through halo_msm,
     * anything else with its generators through halo_msm_affine.  Same generators -> same point, whichever route. */
    size_t mk = n < 512 ? n : 512, koff = n - mk;
    uint64_t *gs = malloc(mk * 64), P1[12], P2[12];
    CK(halo_ctx_read_bases(ctx, koff, mk, gs));
    CK(halo_msm(ctx, koff, mk, coeffs, 1, P1));
    CK(halo_msm_affine(ctx, gs, coeffs, mk, 1, P2));
    int routes = memcmp(P1, P2, 96) == 0;
    free(gs);
    printf("point_dot_affine routes agree: %s\n", routes ? "yes" : "NO");
    same = same && routes;
    /* acc_rs.patch: AccumulatedHPolys::get_poly / eval (acc.rs:85-106) through ffi::h_accumulate / ffi::h_eval_batch.  Checked
     * with what the C ABI itself offers: the commitment is linear, so commit(h_0 + sum a_i h_i) must equal
     * commit(h_0) + sum a_i commit(h_i) (halo_h_commit per polynomial, the sum by halo_msm_points); and h_i(z) from the batch
     * must equal halo_poly_eval over h_i's expanded coefficients. */
    int acc_ok = 1;
    {
        enum { M = 3 };
        uint64_t xis[M * 64 * 4], alphas[M * 4], h0[8], one_z[8];
        memset(xis, 0, sizeof xis); memset(alphas, 0, sizeof alphas); memset(h0, 0, sizeof h0);
        for (size_t i = 0; i < M; i++) {
            for (size_t k = 0; k <= lg; k++) {
                uint64_t *x = xis + 4 * (i * (lg + 1) + k);
                x[0] = 0xD1B54A32D192ED03ull * (k + 1) + i; x[1] = 0x94D049BB133111EBull ^ (k << 8 | i); x[3] = 0x0F00000ull + 16 * k + i;
            }
            alphas[4 * i] = 0xA0761D6478BD642Full + i; alphas[4 * i + 2] = i + 1; alphas[4 * i + 3] = 0x2000000ull + i;
        }
        h0[0] = 7; h0[3] = 0x111111; h0[4] = 9; h0[7] = 0x222222;
        CK(halo_powers(ctx, z, 2, one_z)); /* one_z[0..4) = 1 in the library's scalar form */
        uint64_t *acc = malloc(n * 32), *hi = malloc(n * 32), pts[(M + 1) * 12], sc[(M + 1) * 4], lhs[12], rhs[12], ev[M * 4], ev1[4];
        CK(halo_h_accumulate(ctx, h0, xis, alphas, M, lg, acc));
        CK(halo_pcdl_commit(ctx, acc, n, d, NULL, lhs));
        CK(halo_pcdl_commit(ctx, h0, n < 2 ? n : 2, d, NULL, pts));
        memcpy(sc, one_z, 32);
        for (size_t i = 0; i < M; i++) {
            CK(halo_h_commit(ctx, xis + 4 * i * (lg + 1), lg, pts + 12 * (i + 1)));
            memcpy(sc + 4 * (i + 1), alphas + 4 * i, 32);
        }
        CK(halo_msm_points(ctx, pts, sc, M + 1, rhs));
        acc_ok = memcmp(lhs, rhs, 96) == 0;
        CK(halo_h_eval_batch(ctx, xis, M, lg, z, ev));
        for (size_t i = 0; i < M; i++) {
            CK(halo_h_coeffs(ctx, xis + 4 * i * (lg + 1), lg, hi));
            CK(halo_poly_eval(ctx, hi, n, z, ev1));
            acc_ok = acc_ok && memcmp(ev + 4 * i, ev1, 32) == 0;
        }
        free(acc); free(hi);
    }
    printf("acc.rs get_poly / eval through h_accumulate / h_eval_batch: %s\n", acc_ok ? "yes" : "NO");
    same = same && acc_ok;
    printf("lg_n=%zu  shim loop == halo_pcdl_open: %s   check(a)=%d check(b)=%d   wire round trip: %s (%zu bytes)\n", lg, same ? "yes" : "NO", ok_a,
           ok_b, wire ? "yes" : "NO", blen);
    halo_ctx_destroy(ctx);
    return same && ok_a == HALO_OK && ok_b == HALO_OK && wire ? 0 : 1;
}
