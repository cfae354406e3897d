/* CPU restatement of the halo-accumulation MSM / IPA path -- TEST INFRASTRUCTURE.
 * See halo_cpu.h for the rules on who may use this file and for data conventions.
 *
 * Single-threaded plain C restating the reference's CPU algorithm:
 *   code/src/group.rs, pedersen.rs, pcdl.rs, acc.rs, main.rs (file:line cited per function)
 * and, because the arithmetic itself lives in un-vendored crates absent from
 * /root/reference (ark-ec / ark-ff / ark-pallas / ark-poly / ark-serialize 0.5.0,
 * sha3 0.10.8 -- versions pinned by code/Cargo.lock), their published algorithms:
 *   - Fp Montgomery backend, 4 x 64-bit limbs, R = 2^256 (ark-ff 0.5 `MontBackend`)
 *   - short-Weierstrass Jacobian group law, a = 0 (ark-ec 0.5 `short_weierstrass::Projective`:
 *     add-2007-bl, madd-2007-bl, dbl-2009-l)
 *   - VariableBaseMSM::msm_unchecked = signed-digit windowed Pippenger with
 *     c = ceil(log2 n) * 69 / 100 + 2 (ark-ec 0.5 `msm_bigint_wnaf`, `make_digits`)
 *   - Keccak-f[1600] / SHA3-256 (FIPS 202)
 *   - ark-serialize 0.5 compressed encodings (transcript bytes: PARITY UNPINNED --
 *     the reference holds no known-answer vector for any rho_0!/rho_1! output)
 *
 * Pinned by tests/test_oracle_golden.py: all 16,386 points of code/src/consts.rs
 * (digest + samples in tests/golden/urs_kat.json), the Python big-int model
 * oracle/pallas_model.py, and the identities of the reference's own unit tests.
 */
#define _POSIX_C_SOURCE 199309L
#include "halo_cpu.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef unsigned __int128 u128;
typedef uint64_t u64;

typedef struct { u64 l[4]; } fe;
typedef struct {
    u64 p[4];
    u64 inv; /* -p^-1 mod 2^64 */
    fe one;  /* R mod p */
    fe r2;   /* R^2 mod p */
    u64 pm2[4];
} field_t;

/* Both moduli and their Montgomery constants as compile-time constants: every field function below is `static inline` and is
 * always called with &FQ or &FR, so the compiler folds the limbs in (p[2] = 0 and p[3] = 2^62 for both Pasta primes: two of
 * the four reduction products of a row disappear).  ensure_init() recomputes every constant from p alone and refuses to run
 * if one of them differs (field_selfcheck). */
static const field_t FQ = {
    {0x992d30ed00000001ULL, 0x224698fc094cf91bULL, 0x0000000000000000ULL, 0x4000000000000000ULL},
    0x992d30ecffffffffULL,
    {{0x34786d38fffffffdULL, 0x992c350be41914adULL, 0xffffffffffffffffULL, 0x3fffffffffffffffULL}},
    {{0x8c78ecb30000000fULL, 0xd7d30dbd8b0de0e7ULL, 0x7797a99bc3c95d18ULL, 0x096d41af7b9cb714ULL}},
    {0x992d30ecffffffffULL, 0x224698fc094cf91bULL, 0x0000000000000000ULL, 0x4000000000000000ULL}};
static const field_t FR = {
    {0x8c46eb2100000001ULL, 0x224698fc0994a8ddULL, 0x0000000000000000ULL, 0x4000000000000000ULL},
    0x8c46eb20ffffffffULL,
    {{0x5b2b3e9cfffffffdULL, 0x992c350be3420567ULL, 0xffffffffffffffffULL, 0x3fffffffffffffffULL}},
    {{0xfc9678ff0000000fULL, 0x67bb433d891a16e3ULL, 0x7fae231004ccf590ULL, 0x096d41af7ccfdaa9ULL}},
    {0x8c46eb20ffffffffULL, 0x224698fc0994a8ddULL, 0x0000000000000000ULL, 0x4000000000000000ULL}};
static int g_init = 0;
static char g_err[256] = "";

const char *orc_last_error(void) { return g_err; }
static int fail(const char *msg) { snprintf(g_err, sizeof g_err, "%s", msg); return -1; }

/* ------------------------------------------------------------------ field */
#define INL static inline __attribute__((always_inline))
static int ge4(const u64 a[4], const u64 b[4]) {
    for (int i = 3; i >= 0; i--) { if (a[i] != b[i]) return a[i] > b[i]; }
    return 1;
}
INL u64 sub4(u64 r[4], const u64 a[4], const u64 b[4]) {
    u64 br = 0;
    for (int i = 0; i < 4; i++) { u128 d = (u128)a[i] - b[i] - br; r[i] = (u64)d; br = (u64)(d >> 64) & 1; }
    return br;
}
INL u64 add4(u64 r[4], const u64 a[4], const u64 b[4]) {
    u64 c = 0;
    for (int i = 0; i < 4; i++) { u128 s = (u128)a[i] + b[i] + c; r[i] = (u64)s; c = (u64)(s >> 64); }
    return c;
}
/* r = keep ? a : b, limb by limb, without a branch (a data-dependent branch here is mispredicted every other time: the sums
 * of a point addition are uniform below 2 p) */
INL void sel4(u64 r[4], u64 keep, const u64 a[4], const u64 b[4]) {
    const u64 m = (u64)0 - keep;
    for (int i = 0; i < 4; i++) r[i] = (a[i] & m) | (b[i] & ~m);
}
/* ark-ff 0.5 Fp::add_assign: a + b, minus p if that is not below p (both moduli < 2^255: the sum has no fifth word) */
INL void fe_add(fe *r, const fe *a, const fe *b, const field_t *F) {
    u64 t[4], s[4];
    add4(t, a->l, b->l);
    const u64 below = sub4(s, t, F->p);
    sel4(r->l, below, t, s);
}
/* Fp::sub_assign: a - b, plus p if that borrowed */
INL void fe_sub(fe *r, const fe *a, const fe *b, const field_t *F) {
    u64 t[4], s[4];
    const u64 borrowed = sub4(t, a->l, b->l);
    add4(s, t, F->p);
    sel4(r->l, borrowed, s, t);
}
INL int fe_is_zero(const fe *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
INL int fe_eq(const fe *a, const fe *b) { return ((a->l[0] ^ b->l[0]) | (a->l[1] ^ b->l[1]) | (a->l[2] ^ b->l[2]) | (a->l[3] ^ b->l[3])) == 0; }
INL void fe_neg(fe *r, const fe *a, const field_t *F) {
    if (fe_is_zero(a)) { *r = *a; return; }
    sub4(r->l, F->p, a->l);
}
INL void fe_dbl(fe *r, const fe *a, const field_t *F) { fe_add(r, a, a, F); }

/* CIOS Montgomery multiplication, the four rows written out.  Both Pasta moduli are below 2^255, so the running value stays
 * below 2 p < 2^256 and needs no fifth word: ark-ff's "no-carry" variant for moduli with a spare top bit
 * (ark-ff 0.5 montgomery_backend.rs, mul_assign: `can_use_no_carry_mul_optimization`; pinned by code/Cargo.lock). */
#define MUL_ROW(bi) do { \
        u128 c = (u128)a0 * (bi) + t0; u64 lo = (u64)c; c >>= 64; \
        u64 m = lo * inv; \
        u128 k = (u128)m * p0 + lo; k >>= 64; \
        c += (u128)a1 * (bi) + t1; k += (u128)m * p1 + (u64)c; t0 = (u64)k; c >>= 64; k >>= 64; \
        c += (u128)a2 * (bi) + t2; k += (u128)m * p2 + (u64)c; t1 = (u64)k; c >>= 64; k >>= 64; \
        c += (u128)a3 * (bi) + t3; k += (u128)m * p3 + (u64)c; t2 = (u64)k; c >>= 64; k >>= 64; \
        t3 = (u64)c + (u64)k; \
    } while (0)
INL void fe_mul(fe *r, const fe *a, const fe *b, const field_t *F) {
    const u64 a0 = a->l[0], a1 = a->l[1], a2 = a->l[2], a3 = a->l[3];
    const u64 b0 = b->l[0], b1 = b->l[1], b2 = b->l[2], b3 = b->l[3];
    const u64 p0 = F->p[0], p1 = F->p[1], p2 = F->p[2], p3 = F->p[3], inv = F->inv;
    u64 t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    MUL_ROW(b0); MUL_ROW(b1); MUL_ROW(b2); MUL_ROW(b3);
    const u64 t[4] = {t0, t1, t2, t3};
    u64 s[4];
    const u64 below = sub4(s, t, F->p);
    sel4(r->l, below, t, s);
}
INL void fe_sqr(fe *r, const fe *a, const field_t *F) { fe_mul(r, a, a, F); }
static void fe_to_mont(fe *r, const fe *canon, const field_t *F) { fe_mul(r, canon, &F->r2, F); }
static void fe_from_mont(fe *r, const fe *m, const field_t *F) { fe one = {{1, 0, 0, 0}}; fe_mul(r, m, &one, F); }
static void fe_pow(fe *r, const fe *a, const u64 e[4], const field_t *F) {
    fe acc = F->one;
    for (int i = 255; i >= 0; i--) {
        fe_sqr(&acc, &acc, F);
        if ((e[i / 64] >> (i % 64)) & 1) fe_mul(&acc, &acc, a, F);
    }
    *r = acc;
}
/* ark-ff `inverse()`: returns None for zero */
static int fe_inv(fe *r, const fe *a, const field_t *F) {
    if (fe_is_zero(a)) return 0;
    fe_pow(r, a, F->pm2, F);
    return 1;
}

/* every constant of a field_t recomputed from p alone (Newton for -p^-1, 512 modular doublings for R and R^2) */
static int field_selfcheck(const field_t *F) {
    u64 x = 1;
    for (int i = 0; i < 6; i++) x *= 2 - F->p[0] * x;
    if (F->inv != (u64)0 - x) return 0;
    fe acc = {{1, 0, 0, 0}};
    for (int i = 0; i < 512; i++) {
        fe_add(&acc, &acc, &acc, F);
        if (i == 255 && !fe_eq(&acc, &F->one)) return 0;
    }
    if (!fe_eq(&acc, &F->r2)) return 0;
    u64 two[4] = {2, 0, 0, 0}, pm2[4];
    sub4(pm2, F->p, two);
    return memcmp(pm2, F->pm2, 32) == 0;
}

static void ensure_init(void) {
    if (g_init) return;
    if (!field_selfcheck(&FQ) || !field_selfcheck(&FR)) { fprintf(stderr, "oracle: field constants do not match their moduli\n"); abort(); }
    g_init = 1;
}

/* ns per Fq Montgomery product on this host: four independent chains (the point formulas offer at least that much parallelism),
 * so the figure is the product's throughput, which is what an MSM's cost is made of.  bench.py puts it into cpu_baseline. */
double orc_bench_fq_mul(size_t iters, uint64_t out[4]) {
    ensure_init();
    fe x[4], b = FQ.r2;
    for (int k = 0; k < 4; k++) { x[k] = FQ.one; x[k].l[0] ^= (u64)k; }
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (size_t i = 0; i < iters; i++)
        for (int k = 0; k < 4; k++) fe_mul(&x[k], &x[k], &b, &FQ);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    fe acc = x[0];
    for (int k = 1; k < 4; k++) fe_add(&acc, &acc, &x[k], &FQ);
    memcpy(out, acc.l, 32);
    return ((double)(t1.tv_sec - t0.tv_sec) * 1e9 + (double)(t1.tv_nsec - t0.tv_nsec)) / ((double)iters * 4.0);
}

/* ------------------------------------------------------------------ curve */
typedef struct { fe X, Y, Z; } jac;
typedef struct { fe x, y; } aff; /* (0,0) = infinity */

static int jac_is_inf(const jac *p) { return fe_is_zero(&p->Z); }
static int aff_is_inf(const aff *p) { return fe_is_zero(&p->x) && fe_is_zero(&p->y); }
static void jac_set_inf(jac *p) { p->X = FQ.one; p->Y = FQ.one; memset(&p->Z, 0, sizeof(fe)); }
static void jac_from_aff(jac *r, const aff *a) {
    if (aff_is_inf(a)) { jac_set_inf(r); return; }
    r->X = a->x; r->Y = a->y; r->Z = FQ.one;
}
/* dbl-2009-l (a = 0) */
static void jac_dbl(jac *r, const jac *p) {
    if (jac_is_inf(p)) { *r = *p; return; }
    fe A, B, C, D, E, Fv, t, X3, Y3, Z3;
    fe_sqr(&A, &p->X, &FQ);
    fe_sqr(&B, &p->Y, &FQ);
    fe_sqr(&C, &B, &FQ);
    fe_add(&t, &p->X, &B, &FQ); fe_sqr(&t, &t, &FQ); fe_sub(&t, &t, &A, &FQ); fe_sub(&t, &t, &C, &FQ);
    fe_dbl(&D, &t, &FQ);
    fe_dbl(&E, &A, &FQ); fe_add(&E, &E, &A, &FQ);
    fe_sqr(&Fv, &E, &FQ);
    fe_mul(&Z3, &p->Y, &p->Z, &FQ); fe_dbl(&Z3, &Z3, &FQ);
    fe_sub(&X3, &Fv, &D, &FQ); fe_sub(&X3, &X3, &D, &FQ);
    fe_sub(&t, &D, &X3, &FQ); fe_mul(&Y3, &E, &t, &FQ);
    fe_dbl(&t, &C, &FQ); fe_dbl(&t, &t, &FQ); fe_dbl(&t, &t, &FQ);
    fe_sub(&Y3, &Y3, &t, &FQ);
    r->X = X3; r->Y = Y3; r->Z = Z3;
}
/* add-2007-bl */
static void jac_add(jac *r, const jac *p, const jac *q) {
    if (jac_is_inf(p)) { *r = *q; return; }
    if (jac_is_inf(q)) { *r = *p; return; }
    fe Z1Z1, Z2Z2, U1, U2, S1, S2, H, I, J, rr, V, t, X3, Y3, Z3;
    fe_sqr(&Z1Z1, &p->Z, &FQ);
    fe_sqr(&Z2Z2, &q->Z, &FQ);
    fe_mul(&U1, &p->X, &Z2Z2, &FQ);
    fe_mul(&U2, &q->X, &Z1Z1, &FQ);
    fe_mul(&S1, &p->Y, &q->Z, &FQ); fe_mul(&S1, &S1, &Z2Z2, &FQ);
    fe_mul(&S2, &q->Y, &p->Z, &FQ); fe_mul(&S2, &S2, &Z1Z1, &FQ);
    if (fe_eq(&U1, &U2)) {
        if (fe_eq(&S1, &S2)) { jac_dbl(r, p); return; }
        jac_set_inf(r); return;
    }
    fe_sub(&H, &U2, &U1, &FQ);
    fe_dbl(&I, &H, &FQ); fe_sqr(&I, &I, &FQ);
    fe_mul(&J, &H, &I, &FQ);
    fe_sub(&rr, &S2, &S1, &FQ); fe_dbl(&rr, &rr, &FQ);
    fe_mul(&V, &U1, &I, &FQ);
    fe_sqr(&X3, &rr, &FQ); fe_sub(&X3, &X3, &J, &FQ); fe_sub(&X3, &X3, &V, &FQ); fe_sub(&X3, &X3, &V, &FQ);
    fe_sub(&t, &V, &X3, &FQ); fe_mul(&Y3, &rr, &t, &FQ);
    fe_mul(&t, &S1, &J, &FQ); fe_dbl(&t, &t, &FQ); fe_sub(&Y3, &Y3, &t, &FQ);
    fe_add(&Z3, &p->Z, &q->Z, &FQ); fe_sqr(&Z3, &Z3, &FQ); fe_sub(&Z3, &Z3, &Z1Z1, &FQ); fe_sub(&Z3, &Z3, &Z2Z2, &FQ);
    fe_mul(&Z3, &Z3, &H, &FQ);
    r->X = X3; r->Y = Y3; r->Z = Z3;
}
/* madd-2007-bl */
static void jac_add_aff(jac *r, const jac *p, const aff *q) {
    if (aff_is_inf(q)) { *r = *p; return; }
    if (jac_is_inf(p)) { jac_from_aff(r, q); return; }
    fe Z1Z1, U2, S2, H, HH, I, J, rr, V, t, X3, Y3, Z3;
    fe_sqr(&Z1Z1, &p->Z, &FQ);
    fe_mul(&U2, &q->x, &Z1Z1, &FQ);
    fe_mul(&S2, &q->y, &p->Z, &FQ); fe_mul(&S2, &S2, &Z1Z1, &FQ);
    if (fe_eq(&p->X, &U2)) {
        if (fe_eq(&p->Y, &S2)) { jac_dbl(r, p); return; }
        jac_set_inf(r); return;
    }
    fe_sub(&H, &U2, &p->X, &FQ);
    fe_sqr(&HH, &H, &FQ);
    fe_dbl(&I, &HH, &FQ); fe_dbl(&I, &I, &FQ);
    fe_mul(&J, &H, &I, &FQ);
    fe_sub(&rr, &S2, &p->Y, &FQ); fe_dbl(&rr, &rr, &FQ);
    fe_mul(&V, &p->X, &I, &FQ);
    fe_sqr(&X3, &rr, &FQ); fe_sub(&X3, &X3, &J, &FQ); fe_sub(&X3, &X3, &V, &FQ); fe_sub(&X3, &X3, &V, &FQ);
    fe_sub(&t, &V, &X3, &FQ); fe_mul(&Y3, &rr, &t, &FQ);
    fe_mul(&t, &p->Y, &J, &FQ); fe_dbl(&t, &t, &FQ); fe_sub(&Y3, &Y3, &t, &FQ);
    fe_add(&Z3, &p->Z, &H, &FQ); fe_sqr(&Z3, &Z3, &FQ); fe_sub(&Z3, &Z3, &Z1Z1, &FQ); fe_sub(&Z3, &Z3, &HH, &FQ);
    r->X = X3; r->Y = Y3; r->Z = Z3;
}
static void jac_neg(jac *r, const jac *p) { r->X = p->X; r->Z = p->Z; fe_neg(&r->Y, &p->Y, &FQ); }
static void aff_neg(aff *r, const aff *p) { r->x = p->x; fe_neg(&r->y, &p->y, &FQ); }
/* CurveGroup::into_affine: one field inversion per point (group.rs:19) */
static void jac_to_aff(aff *r, const jac *p) {
    if (jac_is_inf(p)) { memset(r, 0, sizeof *r); return; }
    fe zi = {{0, 0, 0, 0}}, zi2, zi3;
    fe_inv(&zi, &p->Z, &FQ);
    fe_sqr(&zi2, &zi, &FQ);
    fe_mul(&zi3, &zi2, &zi, &FQ);
    fe_mul(&r->x, &p->X, &zi2, &FQ);
    fe_mul(&r->y, &p->Y, &zi3, &FQ);
}
static void jac_normalize(jac *p) {
    if (jac_is_inf(p)) { jac_set_inf(p); return; }
    aff a; jac_to_aff(&a, p); jac_from_aff(p, &a);
}
static int jac_eq(const jac *a, const jac *b) { /* projective equality as ark-ec PartialEq */
    if (jac_is_inf(a) || jac_is_inf(b)) return jac_is_inf(a) && jac_is_inf(b);
    fe z1z1, z2z2, l, r;
    fe_sqr(&z1z1, &a->Z, &FQ); fe_sqr(&z2z2, &b->Z, &FQ);
    fe_mul(&l, &a->X, &z2z2, &FQ); fe_mul(&r, &b->X, &z1z1, &FQ);
    if (!fe_eq(&l, &r)) return 0;
    fe_mul(&z1z1, &z1z1, &a->Z, &FQ); fe_mul(&z2z2, &z2z2, &b->Z, &FQ);
    fe_mul(&l, &a->Y, &z2z2, &FQ); fe_mul(&r, &b->Y, &z1z1, &FQ);
    return fe_eq(&l, &r);
}
/* `Projective * Fr`: MSB-first double-and-add over the canonical scalar (ark-ec mul_bigint) */
static void jac_mul_canon(jac *r, const jac *p, const u64 k[4]) {
    jac acc; jac_set_inf(&acc);
    int started = 0;
    for (int i = 255; i >= 0; i--) {
        int bit = (k[i / 64] >> (i % 64)) & 1;
        if (started) jac_dbl(&acc, &acc);
        if (bit) { jac_add(&acc, &acc, p); started = 1; }
    }
    *r = acc;
}
static void jac_mul(jac *r, const jac *p, const fe *k_mont) {
    fe k; fe_from_mont(&k, k_mont, &FR);
    jac_mul_canon(r, p, k.l);
}

static void load_jac(jac *p, const u64 w[12]) { memcpy(p, w, 96); }
static void store_jac(u64 w[12], const jac *p) { jac t = *p; jac_normalize(&t); memcpy(w, &t, 96); }
static void load_aff(aff *p, const u64 w[8]) { memcpy(p, w, 64); }
static void load_fe(fe *p, const u64 w[4]) { memcpy(p, w, 32); }
static void store_fe(u64 w[4], const fe *p) { memcpy(w, p, 32); }

/* ------------------------------------------------------------- public helpers */
void orc_fr_to_mont(const u64 c[4], u64 o[4]) { ensure_init(); fe a, r; load_fe(&a, c); fe_to_mont(&r, &a, &FR); store_fe(o, &r); }
void orc_fr_from_mont(const u64 m[4], u64 o[4]) { ensure_init(); fe a, r; load_fe(&a, m); fe_from_mont(&r, &a, &FR); store_fe(o, &r); }
void orc_fq_to_mont(const u64 c[4], u64 o[4]) { ensure_init(); fe a, r; load_fe(&a, c); fe_to_mont(&r, &a, &FQ); store_fe(o, &r); }
void orc_fq_from_mont(const u64 m[4], u64 o[4]) { ensure_init(); fe a, r; load_fe(&a, m); fe_from_mont(&r, &a, &FQ); store_fe(o, &r); }
void orc_fr_mul(const u64 a[4], const u64 b[4], u64 o[4]) { ensure_init(); fe x, y, r; load_fe(&x, a); load_fe(&y, b); fe_mul(&r, &x, &y, &FR); store_fe(o, &r); }
void orc_fr_add(const u64 a[4], const u64 b[4], u64 o[4]) { ensure_init(); fe x, y, r; load_fe(&x, a); load_fe(&y, b); fe_add(&r, &x, &y, &FR); store_fe(o, &r); }
int orc_fr_inv(const u64 a[4], u64 o[4]) { ensure_init(); fe x, r; load_fe(&x, a); if (!fe_inv(&r, &x, &FR)) return -1; store_fe(o, &r); return 0; }
void orc_fq_mul(const u64 a[4], const u64 b[4], u64 o[4]) { ensure_init(); fe x, y, r; load_fe(&x, a); load_fe(&y, b); fe_mul(&r, &x, &y, &FQ); store_fe(o, &r); }

int orc_point_canonical(const u64 w[12], uint8_t out[64]) {
    ensure_init();
    jac p; load_jac(&p, w);
    memset(out, 0, 64);
    if (jac_is_inf(&p)) return 1;
    aff a; jac_to_aff(&a, &p);
    fe x, y; fe_from_mont(&x, &a.x, &FQ); fe_from_mont(&y, &a.y, &FQ);
    memcpy(out, x.l, 32); memcpy(out + 32, y.l, 32); /* little-endian host */
    return 0;
}
void orc_point_add(const u64 a[12], const u64 b[12], u64 o[12]) { ensure_init(); jac p, q, r; load_jac(&p, a); load_jac(&q, b); jac_add(&r, &p, &q); store_jac(o, &r); }
void orc_point_mul(const u64 a[12], const u64 k[4], u64 o[12]) { ensure_init(); jac p, r; fe s; load_jac(&p, a); load_fe(&s, k); jac_mul(&r, &p, &s); store_jac(o, &r); }
void orc_affine_to_jac(const u64 a[8], u64 o[12]) { ensure_init(); aff p; jac r; load_aff(&p, a); jac_from_aff(&r, &p); memcpy(o, &r, 96); }

/* ------------------------------------------------------------------- rng */
u64 orc_rng_u64(u64 *s) {
    *s += 0x9E3779B97F4A7C15ULL;
    u64 z = *s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
void orc_rng_scalar(u64 *s, u64 out[4]) {
    ensure_init();
    fe v;
    for (int i = 0; i < 4; i++) v.l[i] = orc_rng_u64(s);
    while (ge4(v.l, FR.p)) sub4(v.l, v.l, FR.p); /* 2^256 < 4r: at most 3 subtractions */
    fe m; fe_to_mont(&m, &v, &FR); store_fe(out, &m);
}
void orc_rng_scalars(u64 *s, size_t n, u64 *out) { for (size_t i = 0; i < n; i++) orc_rng_scalar(s, out + 4 * i); }

/* ------------------------------------------------------------------ sha3 */
static const u64 KRC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
    0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
    0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
    0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
static const int KROT[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
static const int KPIL[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
static u64 rotl64(u64 x, int n) { return (x << n) | (x >> (64 - n)); }
static void keccakf(u64 st[25]) {
    for (int round = 0; round < 24; round++) {
        u64 bc[5], t;
        for (int i = 0; i < 5; i++) bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
        for (int i = 0; i < 5; i++) { t = bc[(i + 4) % 5] ^ rotl64(bc[(i + 1) % 5], 1); for (int j = 0; j < 25; j += 5) st[j + i] ^= t; }
        t = st[1];
        for (int i = 0; i < 24; i++) { int j = KPIL[i]; u64 b = st[j]; st[j] = rotl64(t, KROT[i]); t = b; }
        for (int j = 0; j < 25; j += 5) {
            for (int i = 0; i < 5; i++) bc[i] = st[j + i];
            for (int i = 0; i < 5; i++) st[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
        }
        st[0] ^= KRC[round];
    }
}
void orc_sha3_256(const uint8_t *data, size_t len, uint8_t out[32]) {
    u64 st[25]; memset(st, 0, sizeof st);
    const size_t rate = 136;
    uint8_t block[136];
    while (len >= rate) {
        for (size_t i = 0; i < rate / 8; i++) { u64 w; memcpy(&w, data + 8 * i, 8); st[i] ^= w; }
        keccakf(st); data += rate; len -= rate;
    }
    memset(block, 0, rate); memcpy(block, data, len);
    block[len] ^= 0x06; block[rate - 1] ^= 0x80;
    for (size_t i = 0; i < rate / 8; i++) { u64 w; memcpy(&w, block + 8 * i, 8); st[i] ^= w; }
    keccakf(st);
    memcpy(out, st, 32);
}

/* ark-ff from_le_bytes_mod_order on a 32-byte digest */
static void fe_from_le_bytes_mod_order(fe *out_mont, const uint8_t b[32], const field_t *F) {
    fe v; memcpy(v.l, b, 32);
    while (ge4(v.l, F->p)) sub4(v.l, v.l, F->p);
    fe_to_mont(out_mont, &v, F);
}

/* ------------------------------------------------- transcript (group.rs:41-89) */
typedef struct { uint8_t *buf; size_t len, cap; } bytes_t;
static void bput(bytes_t *b, const void *src, size_t n) {
    if (b->len + n > b->cap) { b->cap = (b->len + n) * 2 + 64; b->buf = (uint8_t *)realloc(b->buf, b->cap); }
    memcpy(b->buf + b->len, src, n); b->len += n;
}
static void bput_u64(bytes_t *b, u64 v) { bput(b, &v, 8); }
static void bput_u8(bytes_t *b, uint8_t v) { bput(b, &v, 1); }
static void bput_scalar(bytes_t *b, const fe *s_mont) { fe c; fe_from_mont(&c, s_mont, &FR); bput(b, c.l, 32); }
/* ark-serialize compressed SW point: x LE (32 B) then flag byte: bit7 = y > -y, bit6 = infinity */
static void bput_point(bytes_t *b, const jac *p) {
    uint8_t out[33]; memset(out, 0, 33);
    if (jac_is_inf(p)) { out[32] = 0x40; bput(b, out, 33); return; }
    aff a; jac_to_aff(&a, p);
    fe x, y, ny, nyc; fe_from_mont(&x, &a.x, &FQ); fe_from_mont(&y, &a.y, &FQ);
    fe_neg(&ny, &a.y, &FQ); fe_from_mont(&nyc, &ny, &FQ);
    memcpy(out, x.l, 32);
    int y_gt = !ge4(nyc.l, y.l); /* y > -y */
    if (y_gt) out[32] |= 0x80;
    bput(b, out, 33);
}
static void rho_finish(fe *out, bytes_t *b, uint32_t tag) {
    bput(b, &tag, 4);
    uint8_t dig[32]; orc_sha3_256(b->buf, b->len, dig);
    fe_from_le_bytes_mod_order(out, dig, &FR);
    free(b->buf); b->buf = NULL; b->len = b->cap = 0;
}
void orc_rho(int tag, const int *kinds, const u64 *const *items, size_t n, u64 out[4]) {
    ensure_init();
    bytes_t b = {0, 0, 0};
    for (size_t i = 0; i < n; i++) {
        if (kinds[i] == 0) { fe s; load_fe(&s, items[i]); bput_scalar(&b, &s); }
        else { jac p; load_jac(&p, items[i]); bput_point(&b, &p); }
    }
    fe r; rho_finish(&r, &b, (uint32_t)tag); store_fe(out, &r);
}

/* -------------------------------------------------------- main.rs:18-45 (URS) */
static const char GENESIS[] = "To understand recursion, one must first understand recursion";
static void urs_scalar(fe *out, u64 index) {
    uint8_t buf[sizeof(GENESIS) - 1 + 8];
    memcpy(buf, GENESIS, sizeof(GENESIS) - 1);
    memcpy(buf + sizeof(GENESIS) - 1, &index, 8);
    uint8_t dig[32]; orc_sha3_256(buf, sizeof buf, dig);
    fe_from_le_bytes_mod_order(out, dig, &FR);
}
static void generator(jac *g) {
    fe one = FQ.one, two;
    fe_neg(&g->X, &one, &FQ);
    fe_add(&two, &one, &one, &FQ);
    g->Y = two; g->Z = one;
}
void orc_urs_scalar(u64 index, u64 out[4]) { ensure_init(); fe s; urs_scalar(&s, index); store_fe(out, &s); }
void orc_urs_point(u64 index, u64 out[12]) {
    ensure_init();
    fe s; urs_scalar(&s, index);
    jac g, r; generator(&g); jac_mul(&r, &g, &s); store_jac(out, &r);
}
void orc_urs_affine(u64 first, size_t count, u64 *out) {
    ensure_init();
    jac g; generator(&g);
    for (size_t i = 0; i < count; i++) {
        fe s; urs_scalar(&s, first + i);
        jac r; jac_mul(&r, &g, &s);
        aff a; jac_to_aff(&a, &r); memcpy(out + 8 * i, &a, 64);
    }
}
void orc_pp_init(orc_pp *pp, const u64 *gs, size_t n) {
    ensure_init();
    orc_urs_point(0, pp->S); orc_urs_point(1, pp->H);
    pp->GS = gs; pp->N = n;
}

/* ---------------------------------------------------------------- group.rs */
/* group.rs:13-15 */
void orc_scalar_dot(const u64 *xs, const u64 *ys, size_t m, u64 out[4]) {
    ensure_init();
    fe acc; memset(&acc, 0, sizeof acc);
    for (size_t i = 0; i < m; i++) { fe a, b, t; load_fe(&a, xs + 4 * i); load_fe(&b, ys + 4 * i); fe_mul(&t, &a, &b, &FR); fe_add(&acc, &acc, &t, &FR); }
    store_fe(out, &acc);
}
/* group.rs:29-37 */
void orc_powers(const u64 z[4], size_t n, u64 *out) {
    ensure_init();
    fe zz, cur = FR.one; load_fe(&zz, z);
    for (size_t i = 0; i < n; i++) { store_fe(out + 4 * i, &cur); fe_mul(&cur, &cur, &zz, &FR); }
}

static size_t log2_ceil(size_t x) { /* ark_std::log2 */
    if (x <= 1) return 0;
    size_t l = 0, v = x - 1; while (v) { l++; v >>= 1; } return l;
}
/* ark-ec 0.5 make_digits: signed radix-2^w digits, last digit not recentred */
static void make_digits(const u64 s[4], size_t w, size_t num_bits, int64_t *digits, size_t digits_count) {
    (void)num_bits;
    u64 radix = 1ULL << w, mask = radix - 1, carry = 0;
    for (size_t i = 0; i < digits_count; i++) {
        size_t bit_offset = i * w, idx = bit_offset / 64, bit = bit_offset % 64;
        u64 buf;
        if (bit < 64 - w || idx == 3) buf = s[idx] >> bit;
        else buf = (s[idx] >> bit) | (s[idx + 1] << (64 - bit));
        u64 coef = carry + (buf & mask);
        carry = (coef + radix / 2) >> w;
        int64_t d = (int64_t)coef - (int64_t)(carry << w);
        if (i == digits_count - 1) d += (int64_t)(carry << w);
        digits[i] = d;
    }
}
/* ark-ec 0.5 msm_bigint_wnaf over affine bases and canonical scalars */
static void msm_wnaf(jac *out, const aff *bases, const fe *canon, size_t size) {
    if (size == 0) { jac_set_inf(out); return; }
    size_t c = size < 32 ? 3 : (log2_ceil(size) * 69 / 100) + 2;
    size_t num_bits = 255, digits_count = (num_bits + c - 1) / c;
    int64_t *digits = (int64_t *)malloc(sizeof(int64_t) * digits_count * size);
    for (size_t i = 0; i < size; i++) make_digits(canon[i].l, c, num_bits, digits + i * digits_count, digits_count);
    size_t nb = (size_t)1 << c;
    jac *buckets = (jac *)malloc(sizeof(jac) * nb);
    jac *window_sums = (jac *)malloc(sizeof(jac) * digits_count);
    for (size_t w = 0; w < digits_count; w++) {
        for (size_t b = 0; b < nb; b++) jac_set_inf(&buckets[b]);
        for (size_t i = 0; i < size; i++) {
            int64_t d = digits[i * digits_count + w];
            if (d > 0) jac_add_aff(&buckets[d - 1], &buckets[d - 1], &bases[i]);
            else if (d < 0) { aff n; aff_neg(&n, &bases[i]); jac_add_aff(&buckets[-d - 1], &buckets[-d - 1], &n); }
        }
        jac running, res; jac_set_inf(&running); jac_set_inf(&res);
        for (size_t b = nb; b-- > 0;) { jac_add(&running, &running, &buckets[b]); jac_add(&res, &res, &running); }
        window_sums[w] = res;
    }
    jac total; jac_set_inf(&total);
    for (size_t w = digits_count; w-- > 1;) {
        jac_add(&total, &total, &window_sums[w]);
        for (size_t k = 0; k < c; k++) jac_dbl(&total, &total);
    }
    jac_add(out, &window_sums[0], &total);
    free(digits); free(buckets); free(window_sums);
}
static void msm_affine_mont(jac *out, const u64 *bases, const u64 *scalars, size_t n) {
    fe *canon = (fe *)malloc(sizeof(fe) * (n ? n : 1));
    for (size_t i = 0; i < n; i++) { fe s; load_fe(&s, scalars + 4 * i); fe_from_mont(&canon[i], &s, &FR); } /* into_bigint */
    msm_wnaf(out, (const aff *)bases, canon, n);
    free(canon);
}
/* group.rs:24-26 point_dot_affine */
void orc_msm_affine(const u64 *bases, const u64 *scalars, size_t n, u64 out[12]) {
    ensure_init();
    jac r; msm_affine_mont(&r, bases, scalars, n); store_jac(out, &r);
}
/* group.rs:18-21 point_dot: per-point into_affine, then MSM */
static void point_dot_jac(jac *out, const fe *xs_mont, const jac *pts, size_t m) {
    aff *a = (aff *)malloc(sizeof(aff) * (m ? m : 1));
    for (size_t i = 0; i < m; i++) jac_to_aff(&a[i], &pts[i]);
    msm_affine_mont(out, (const u64 *)a, (const u64 *)xs_mont, m);
    free(a);
}
void orc_msm_jac(const u64 *pts, const u64 *scalars, size_t m, u64 out[12]) {
    ensure_init();
    jac r; point_dot_jac(&r, (const fe *)scalars, (const jac *)pts, m); store_jac(out, &r);
}
void orc_msm_naive(const u64 *bases, const u64 *scalars, size_t n, u64 out[12]) {
    ensure_init();
    jac acc; jac_set_inf(&acc);
    for (size_t i = 0; i < n; i++) {
        aff a; load_aff(&a, bases + 8 * i); jac p, t; jac_from_aff(&p, &a);
        fe s; load_fe(&s, scalars + 4 * i); jac_mul(&t, &p, &s); jac_add(&acc, &acc, &t);
    }
    store_jac(out, &acc);
}

/* ------------------------------------------------------------ pedersen.rs:6-20 */
static int pedersen_commit(jac *out, const orc_pp *pp, const fe *w, const u64 *bases, size_t nb, const u64 *ms, size_t nm) {
    if (nb != nm) return fail("Length did not match for pedersen commitment");
    jac acc; msm_affine_mont(&acc, bases, ms, nb);
    if (w) { jac S, t; load_jac(&S, pp->S); jac_mul(&t, &S, w); jac_add(&acc, &t, &acc); }
    *out = acc; return 0;
}
int orc_pedersen_commit(const orc_pp *pp, const u64 *w, const u64 *bases, size_t nb, const u64 *ms, size_t nm, u64 out[12]) {
    ensure_init();
    fe wf; if (w) load_fe(&wf, w);
    jac r; int rc = pedersen_commit(&r, pp, w ? &wf : NULL, bases, nb, ms, nm);
    if (rc) return rc;
    store_jac(out, &r); return 0;
}

/* ------------------------------------------------------------------ pcdl.rs */
static size_t poly_degree(const u64 *coeffs, size_t len) { /* DensePolynomial::degree after trimming */
    size_t d = 0;
    for (size_t i = 0; i < len; i++) if (coeffs[4 * i] | coeffs[4 * i + 1] | coeffs[4 * i + 2] | coeffs[4 * i + 3]) d = i;
    return d;
}
static int is_pow2(size_t n) { return n && !(n & (n - 1)); }
static size_t ilog2(size_t n) { size_t l = 0; while (n > 1) { n >>= 1; l++; } return l; }

/* pcdl.rs:56-77 : coeff[k] = prod_{bit i of k set} xi_{lg_n - i} (pcdl.rs:496-505) */
void orc_h_coeffs(const u64 *xis, size_t lg_n, u64 *out) {
    ensure_init();
    store_fe(out, &FR.one);
    size_t len = 1;
    for (size_t i = 0; i < lg_n; i++) {
        fe x; load_fe(&x, xis + 4 * (lg_n - i));
        for (size_t k = 0; k < len; k++) { fe c, t; load_fe(&c, out + 4 * k); fe_mul(&t, &c, &x, &FR); store_fe(out + 4 * (len + k), &t); }
        len *= 2;
    }
}
/* pcdl.rs:79-91 */
static void h_eval(fe *out, const fe *xis, size_t lg_n, const fe *z) {
    fe v, zi = *z, t;
    fe_mul(&t, &xis[lg_n], z, &FR); fe_add(&v, &FR.one, &t, &FR);
    for (size_t i = 1; i < lg_n; i++) {
        fe_sqr(&zi, &zi, &FR);
        fe_mul(&t, &xis[lg_n - i], &zi, &FR); fe_add(&t, &FR.one, &t, &FR);
        fe_mul(&v, &v, &t, &FR);
    }
    *out = v;
}
void orc_h_eval(const u64 *xis, size_t lg_n, const u64 z[4], u64 out[4]) {
    ensure_init();
    fe zz, r; load_fe(&zz, z); h_eval(&r, (const fe *)xis, lg_n, &zz); store_fe(out, &r);
}
/* DensePolynomial::evaluate (Horner) -- pcdl.rs:135 */
static void poly_eval(fe *out, const u64 *coeffs, size_t len, const fe *z) {
    fe acc; memset(&acc, 0, sizeof acc);
    for (size_t i = len; i-- > 0;) { fe c; load_fe(&c, coeffs + 4 * i); fe_mul(&acc, &acc, z, &FR); fe_add(&acc, &acc, &c, &FR); }
    *out = acc;
}
void orc_poly_eval(const u64 *coeffs, size_t len, const u64 z[4], u64 out[4]) {
    ensure_init();
    fe zz, r; load_fe(&zz, z); poly_eval(&r, coeffs, len, &zz); store_fe(out, &r);
}

/* pcdl.rs:99-110 */
static int pcdl_commit(jac *out, const orc_pp *pp, const u64 *coeffs, size_t len, size_t d, const fe *w) {
    size_t n = d + 1;
    if (!is_pow2(n)) return fail("commit: d+1 is not a power of two");
    if (poly_degree(coeffs, len) > d) return fail("commit: p.degree() > d");
    if (d > pp->N - 1) return fail("commit: d > D");
    u64 *cs = (u64 *)calloc(n, 32);
    memcpy(cs, coeffs, 32 * (len < n ? len : n));
    int rc = pedersen_commit(out, pp, w, pp->GS, n, cs, n);
    free(cs); return rc;
}
int orc_pcdl_commit(const orc_pp *pp, const u64 *coeffs, size_t len, size_t d, const u64 *w, u64 out[12]) {
    ensure_init();
    fe wf; if (w) load_fe(&wf, w);
    jac r; int rc = pcdl_commit(&r, pp, coeffs, len, d, w ? &wf : NULL);
    if (rc) return rc;
    store_jac(out, &r); return 0;
}

/* proof blob accessors */
static u64 *pf_L(u64 *pf, size_t i) { return pf + 2 + 12 * i; }
static u64 *pf_R(u64 *pf, size_t lg, size_t i) { return pf + 2 + 12 * lg + 12 * i; }
static u64 *pf_U(u64 *pf, size_t lg) { return pf + 2 + 24 * lg; }
static u64 *pf_c(u64 *pf, size_t lg) { return pf + 2 + 24 * lg + 12; }
static u64 *pf_Cbar(u64 *pf, size_t lg) { return pf + 2 + 24 * lg + 16; }
static u64 *pf_wp(u64 *pf, size_t lg) { return pf + 2 + 24 * lg + 28; }

static void rho0_xi_L_R(fe *out, const fe *xi, const jac *L, const jac *R) {
    bytes_t b = {0, 0, 0}; bput_scalar(&b, xi); bput_point(&b, L); bput_point(&b, R); rho_finish(out, &b, 0);
}
static void rho0_C_z_v(fe *out, const jac *C, const fe *z, const fe *v) {
    bytes_t b = {0, 0, 0}; bput_point(&b, C); bput_scalar(&b, z); bput_scalar(&b, v); rho_finish(out, &b, 0);
}
static void rho0_C_z_v_Cbar(fe *out, const jac *C, const fe *z, const fe *v, const jac *Cb) {
    bytes_t b = {0, 0, 0}; bput_point(&b, C); bput_scalar(&b, z); bput_scalar(&b, v); bput_point(&b, Cb); rho_finish(out, &b, 0);
}

/* pcdl.rs:195-227, one round, on explicit Jacobian state */
static void ipa_round_lr(jac *L, jac *R, const jac *gs, const fe *cs, const fe *zs, size_t m, const jac *Hp) {
    fe dot_l, dot_r; jac t;
    orc_scalar_dot((const u64 *)(cs + m), (const u64 *)zs, m, dot_l.l);
    point_dot_jac(L, cs + m, gs, m);              /* point_dot(c_r, g_l) : m into_affine + MSM */
    jac_mul(&t, Hp, &dot_l); jac_add(L, L, &t);
    orc_scalar_dot((const u64 *)cs, (const u64 *)(zs + m), m, dot_r.l);
    point_dot_jac(R, cs, gs + m, m);              /* point_dot(c_l, g_r) */
    jac_mul(&t, Hp, &dot_r); jac_add(R, R, &t);
}
static void ipa_round_fold(jac *gs, fe *cs, fe *zs, size_t m, const fe *xi, const fe *xi_inv) {
    for (size_t j = 0; j < m; j++) {
        jac t; jac_mul(&t, &gs[j + m], xi); jac_add(&gs[j], &gs[j], &t); /* pcdl.rs:218 */
        fe u; fe_mul(&u, &cs[j + m], xi_inv, &FR); fe_add(&cs[j], &cs[j], &u, &FR); /* :222 */
        fe_mul(&u, &zs[j + m], xi, &FR); fe_add(&zs[j], &zs[j], &u, &FR); /* :223 */
    }
}
void orc_ipa_round_lr(const u64 *gs, const u64 *cs, const u64 *zs, size_t m, const u64 Hp[12], u64 L[12], u64 R[12]) {
    ensure_init();
    jac l, r, h; load_jac(&h, Hp);
    ipa_round_lr(&l, &r, (const jac *)gs, (const fe *)cs, (const fe *)zs, m, &h);
    store_jac(L, &l); store_jac(R, &r);
}
void orc_ipa_round_fold(u64 *gs, u64 *cs, u64 *zs, size_t m, const u64 xi[4], const u64 xi_inv[4]) {
    ensure_init();
    ipa_round_fold((jac *)gs, (fe *)cs, (fe *)zs, m, (const fe *)xi, (const fe *)xi_inv);
}

/* pcdl.rs:120-242 */
int orc_pcdl_open(const orc_pp *pp, u64 *rng, const u64 *coeffs, size_t len, const u64 Cw[12], size_t d,
                  const u64 zw[4], const u64 *w, u64 *proof) {
    ensure_init();
    size_t n = d + 1;
    if (!is_pow2(n)) return fail("open: d+1 is not a power of two");
    size_t lg_n = ilog2(n), deg = poly_degree(coeffs, len);
    if (deg > d) return fail("open: p.degree() > d");
    if (d > pp->N - 1) return fail("open: d > D");
    fe z, v; load_fe(&z, zw);
    poly_eval(&v, coeffs, len, &z);                                  /* :135 */
    jac C, C_prime, S, Hh; load_jac(&C, Cw); load_jac(&S, pp->S); load_jac(&Hh, pp->H);
    fe *cs = (fe *)calloc(n, sizeof(fe));
    memcpy(cs, coeffs, 32 * (len < n ? len : n));
    memset(proof, 0, 8 * ORC_PROOF_WORDS(lg_n));
    proof[1] = lg_n;
    if (w) {
        if (deg == 0) { free(cs); return fail("open: hiding needs p.degree() >= 1"); } /* usize underflow in :141 */
        fe wf; load_fe(&wf, w);
        /* :140-142  p_bar = q * (X - z), q uniform of degree deg-1 */
        fe *q = (fe *)malloc(sizeof(fe) * deg), *pbar = (fe *)calloc(deg + 1, sizeof(fe));
        for (size_t i = 0; i < deg; i++) orc_rng_scalar(rng, q[i].l);
        for (size_t i = 0; i < deg; i++) {
            fe t; fe_mul(&t, &z, &q[i], &FR);
            fe_sub(&pbar[i], &pbar[i], &t, &FR);
            fe_add(&pbar[i + 1], &pbar[i + 1], &q[i], &FR);
        }
        fe w_bar; orc_rng_scalar(rng, w_bar.l);                      /* :147 */
        jac C_bar;
        int rc = pcdl_commit(&C_bar, pp, (const u64 *)pbar, deg + 1, d, &w_bar); /* :150 */
        if (rc) { free(q); free(pbar); free(cs); return rc; }
        fe a; rho0_C_z_v_Cbar(&a, &C, &z, &v, &C_bar);               /* :153 */
        for (size_t i = 0; i <= deg; i++) { fe t; fe_mul(&t, &pbar[i], &a, &FR); fe_add(&cs[i], &cs[i], &t, &FR); } /* :156 */
        fe w_prime; fe_mul(&w_prime, &w_bar, &a, &FR); fe_add(&w_prime, &w_prime, &wf, &FR); /* :159 */
        jac t1, t2; jac_mul(&t1, &C_bar, &a); jac_mul(&t2, &S, &w_prime); jac_neg(&t2, &t2);
        jac_add(&C_prime, &C, &t1); jac_add(&C_prime, &C_prime, &t2); /* :162 */
        proof[0] = 1;
        store_jac(pf_Cbar(proof, lg_n), &C_bar);
        store_fe(pf_wp(proof, lg_n), &w_prime);
        free(q); free(pbar);
    } else {
        C_prime = C;
        jac inf; jac_set_inf(&inf); memcpy(pf_Cbar(proof, lg_n), &inf, 96);
    }
    fe xi; rho0_C_z_v(&xi, &C_prime, &z, &v);                        /* :180 */
    jac Hp; jac_mul(&Hp, &Hh, &xi);                                  /* :181 */
    jac *gs = (jac *)malloc(sizeof(jac) * n);
    for (size_t i = 0; i < n; i++) { aff a; load_aff(&a, pp->GS + 8 * i); jac_from_aff(&gs[i], &a); } /* :185 */
    fe *zs = (fe *)malloc(sizeof(fe) * n);
    orc_powers(z.l, n, (u64 *)zs);                                   /* :186 */
    size_t m = n / 2;
    for (size_t round = 0; round < lg_n; round++) {
        jac L, R;
        ipa_round_lr(&L, &R, gs, cs, zs, m, &Hp);
        store_jac(pf_L(proof, round), &L); store_jac(pf_R(proof, lg_n, round), &R);
        fe xi_next, xi_inv; rho0_xi_L_R(&xi_next, &xi, &L, &R);       /* :212 */
        if (!fe_inv(&xi_inv, &xi_next, &FR)) { free(gs); free(zs); free(cs); return fail("open: xi = 0"); }
        xi = xi_next;
        ipa_round_fold(gs, cs, zs, m, &xi, &xi_inv);
        m /= 2;
    }
    store_jac(pf_U(proof, lg_n), &gs[0]);
    store_fe(pf_c(proof, lg_n), &cs[0]);
    free(gs); free(zs); free(cs);
    return 0;
}

/* pcdl.rs:252-314 */
static int succinct_check(const orc_pp *pp, const jac *C, size_t d, const fe *z, const fe *v, const u64 *proof_c,
                          fe *xis, jac *U_out) {
    u64 *proof = (u64 *)proof_c;
    size_t n = d + 1;
    if (!is_pow2(n)) return fail("d+1 is not a power of 2!");
    if (d > pp->N - 1) return fail("d was larger than D!");
    size_t lg_n = ilog2(n);
    if (proof[1] != lg_n) return fail("proof length does not match d");
    jac S, Hh, C_prime; load_jac(&S, pp->S); load_jac(&Hh, pp->H);
    if (proof[0]) {
        jac C_bar; load_jac(&C_bar, pf_Cbar(proof, lg_n));
        fe wp; load_fe(&wp, pf_wp(proof, lg_n));
        fe a; rho0_C_z_v_Cbar(&a, C, z, v, &C_bar);
        jac t1, t2; jac_mul(&t1, &C_bar, &a); jac_mul(&t2, &S, &wp); jac_neg(&t2, &t2);
        jac_add(&C_prime, C, &t1); jac_add(&C_prime, &C_prime, &t2);
    } else C_prime = *C;
    rho0_C_z_v(&xis[0], &C_prime, z, v);
    jac Hp, Ci, t; jac_mul(&Hp, &Hh, &xis[0]);
    jac_mul(&t, &Hp, v); jac_add(&Ci, &C_prime, &t);
    for (size_t i = 0; i < lg_n; i++) {
        jac L, R; load_jac(&L, pf_L(proof, i)); load_jac(&R, pf_R(proof, lg_n, i));
        rho0_xi_L_R(&xis[i + 1], &xis[i], &L, &R);
        fe inv; if (!fe_inv(&inv, &xis[i + 1], &FR)) return fail("xi = 0");
        jac a, b; jac_mul(&a, &L, &inv); jac_mul(&b, &R, &xis[i + 1]); jac_add(&a, &a, &b); jac_add(&Ci, &Ci, &a);
    }
    fe c, hz, v_prime; load_fe(&c, pf_c(proof, lg_n));
    h_eval(&hz, xis, lg_n, z); fe_mul(&v_prime, &c, &hz, &FR);
    jac U; load_jac(&U, pf_U(proof, lg_n));
    jac rhs, t2; jac_mul(&rhs, &U, &c); jac_mul(&t2, &Hp, &v_prime); jac_add(&rhs, &rhs, &t2);
    if (!jac_eq(&Ci, &rhs)) return fail("C_(log_n) != CM.Commit_Sigma(c || v')");
    *U_out = U; return 0;
}
int orc_pcdl_succinct_check(const orc_pp *pp, const u64 C[12], size_t d, const u64 z[4], const u64 v[4], const u64 *proof,
                            u64 *xis_out, u64 U_out[12]) {
    ensure_init();
    jac Cj, U; fe zz, vv; load_jac(&Cj, C); load_fe(&zz, z); load_fe(&vv, v);
    size_t lg = is_pow2(d + 1) ? ilog2(d + 1) : 0;
    fe *xis = (fe *)malloc(sizeof(fe) * (lg + 1));
    int rc = succinct_check(pp, &Cj, d, &zz, &vv, proof, xis, &U);
    if (!rc) { memcpy(xis_out, xis, 32 * (lg + 1)); store_jac(U_out, &U); }
    free(xis); return rc;
}
/* pcdl.rs:323-342 */
static int pcdl_check(const orc_pp *pp, const jac *C, size_t d, const fe *z, const fe *v, const u64 *proof) {
    size_t lg = is_pow2(d + 1) ? ilog2(d + 1) : 0;
    fe *xis = (fe *)malloc(sizeof(fe) * (lg + 1));
    jac U;
    int rc = succinct_check(pp, C, d, z, v, proof, xis, &U);
    if (rc) { free(xis); return rc; }
    size_t n = d + 1;
    u64 *h = (u64 *)malloc(32 * n);
    orc_h_coeffs((const u64 *)xis, lg, h);
    jac comm; rc = pedersen_commit(&comm, pp, NULL, pp->GS, n, h, n);
    free(h); free(xis);
    if (rc) return rc;
    if (!jac_eq(&U, &comm)) return fail("U != CM.Commit(ck, h_vec)");
    return 0;
}
int orc_pcdl_check(const orc_pp *pp, const u64 C[12], size_t d, const u64 z[4], const u64 v[4], const u64 *proof) {
    ensure_init();
    jac Cj; fe zz, vv; load_jac(&Cj, C); load_fe(&zz, z); load_fe(&vv, v);
    return pcdl_check(pp, &Cj, d, &zz, &vv, proof);
}

/* ------------------------------------------------------------------- acc.rs */
/* instance blob: C 12 | d 1 | z 4 | v 4 | proof ; accumulator: instance | h0 8 | U0 12 | w 4 */
static const u64 *in_C(const u64 *q) { return q; }
static size_t in_d(const u64 *q) { return (size_t)q[12]; }
static const u64 *in_z(const u64 *q) { return q + 13; }
static const u64 *in_v(const u64 *q) { return q + 17; }
static const u64 *in_pi(const u64 *q) { return q + 21; }

typedef struct {
    fe h0[2]; size_t m; fe **xis; size_t lg_n; fe alpha; fe *alphas; /* alphas[0..m] */
} acc_hpolys;

/* acc.rs:97-106 */
static void acc_h_eval(fe *out, const acc_hpolys *h, const fe *z) {
    fe v; poly_eval(&v, (const u64 *)h->h0, 2, z);
    for (size_t i = 0; i < h->m; i++) { fe e; h_eval(&e, h->xis[i], h->lg_n, z); fe_mul(&e, &e, &h->alphas[i + 1], &FR); fe_add(&v, &v, &e, &FR); }
    *out = v;
}
/* acc.rs:85-94 */
static void acc_h_get_poly(u64 *out, const acc_hpolys *h, size_t n) {
    memset(out, 0, 32 * n);
    memcpy(out, h->h0, 64 <= 32 * n ? 64 : 32 * n);
    u64 *tmp = (u64 *)malloc(32 * n);
    for (size_t i = 0; i < h->m; i++) {
        orc_h_coeffs((const u64 *)h->xis[i], h->lg_n, tmp);
        for (size_t k = 0; k < n; k++) {
            fe c, o; load_fe(&c, tmp + 4 * k); fe_mul(&c, &c, &h->alphas[i + 1], &FR);
            load_fe(&o, out + 4 * k); fe_add(&o, &o, &c, &FR); store_fe(out + 4 * k, &o);
        }
    }
    free(tmp);
}
static void acc_hpolys_free(acc_hpolys *h) {
    for (size_t i = 0; i < h->m; i++) free(h->xis[i]);
    free(h->xis); free(h->alphas);
}
/* acc.rs:135-188 */
static int common_subroutine(const orc_pp *pp, size_t d, const u64 *qs, size_t m, const fe h0[2], const jac *U0, const fe *w,
                             jac *C_bar_out, fe *z_out, acc_hpolys *hs) {
    size_t lg = is_pow2(d + 1) ? ilog2(d + 1) : 0;
    size_t iw = ORC_INSTANCE_WORDS(lg);
    memset(hs, 0, sizeof *hs);
    hs->h0[0] = h0[0]; hs->h0[1] = h0[1]; hs->lg_n = lg;
    hs->xis = (fe **)calloc(m ? m : 1, sizeof(fe *));
    jac *Us = (jac *)malloc(sizeof(jac) * (m + 1));
    Us[0] = *U0;
    jac chk; int rc = pcdl_commit(&chk, pp, (const u64 *)h0, 2, d, NULL); /* :152-155 */
    if (rc) { free(Us); return rc; }
    if (!jac_eq(U0, &chk)) { free(Us); return fail("U_0 != PCDL.Commit(h_0)"); }
    for (size_t i = 0; i < m; i++) {
        const u64 *q = qs + i * iw;
        jac C; fe z, v; load_jac(&C, in_C(q)); load_fe(&z, in_z(q)); load_fe(&v, in_v(q));
        hs->xis[i] = (fe *)malloc(sizeof(fe) * (lg + 1)); hs->m = i + 1;
        rc = succinct_check(pp, &C, in_d(q), &z, &v, in_pi(q), hs->xis[i], &Us[i + 1]); /* :164 */
        if (rc) { free(Us); return rc; }
        if (in_d(q) != d) { free(Us); return fail("d_i != d"); }
    }
    /* :173 alpha = rho_1(hs): h_0: Option<Poly>, hs: Vec<HPoly>, alpha: None, alphas: empty Vec */
    bytes_t b = {0, 0, 0};
    size_t h0len = fe_is_zero(&h0[1]) ? (fe_is_zero(&h0[0]) ? 0 : 1) : 2;
    bput_u8(&b, 1); bput_u64(&b, h0len);
    for (size_t k = 0; k < h0len; k++) bput_scalar(&b, &h0[k]);
    bput_u64(&b, m);
    for (size_t i = 0; i < m; i++) { bput_u64(&b, lg + 1); for (size_t k = 0; k <= lg; k++) bput_scalar(&b, &hs->xis[i][k]); }
    bput_u8(&b, 0); bput_u64(&b, 0);
    rho_finish(&hs->alpha, &b, 1);
    hs->alphas = (fe *)malloc(sizeof(fe) * (m + 1));
    orc_powers(hs->alpha.l, m + 1, (u64 *)hs->alphas);
    jac C; point_dot_jac(&C, hs->alphas, Us, m + 1);                 /* :178 */
    bytes_t b2 = {0, 0, 0}; bput_point(&b2, &C); bput_scalar(&b2, &hs->alpha); rho_finish(z_out, &b2, 1); /* :181 */
    jac S, t; load_jac(&S, pp->S); jac_mul(&t, &S, w); jac_add(C_bar_out, &C, &t); /* :184 */
    free(Us); return 0;
}
/* acc.rs:190-220 */
int orc_acc_prover(const orc_pp *pp, u64 *rng, size_t d, const u64 *qs, size_t m, u64 *acc) {
    ensure_init();
    if (!is_pow2(d + 1)) return fail("prover: d+1 is not a power of two");
    size_t lg = ilog2(d + 1), n = d + 1;
    fe h0[2], w; orc_rng_scalar(rng, h0[0].l); orc_rng_scalar(rng, h0[1].l);   /* :192 */
    jac U0; int rc = pcdl_commit(&U0, pp, (const u64 *)h0, 2, d, NULL);           /* :195 */
    if (rc) return rc;
    orc_rng_scalar(rng, w.l);                                                    /* :198 */
    jac C_bar; fe z; acc_hpolys hs;
    rc = common_subroutine(pp, d, qs, m, h0, &U0, &w, &C_bar, &z, &hs);
    if (rc) { acc_hpolys_free(&hs); return rc; }
    fe v; acc_h_eval(&v, &hs, &z);                                               /* :205 */
    u64 *poly = (u64 *)malloc(32 * n); acc_h_get_poly(poly, &hs, n);
    memset(acc, 0, 8 * ORC_ACC_WORDS(lg));
    store_jac(acc, &C_bar); acc[12] = d; store_fe(acc + 13, &z); store_fe(acc + 17, &v);
    u64 Cw[12]; memcpy(Cw, acc, 96);
    rc = orc_pcdl_open(pp, rng, poly, n, Cw, d, z.l, w.l, acc + 21);             /* :209 */
    u64 *piV = acc + ORC_INSTANCE_WORDS(lg);
    memcpy(piV, h0, 64); store_jac(piV + 8, &U0); store_fe(piV + 20, &w);
    free(poly); acc_hpolys_free(&hs);
    return rc;
}
/* acc.rs:223-243 */
int orc_acc_verifier(const orc_pp *pp, size_t d, const u64 *qs, size_t m, const u64 *acc) {
    ensure_init();
    if (!is_pow2(d + 1)) return fail("verifier: d+1 is not a power of two");
    size_t lg = ilog2(d + 1);
    const u64 *piV = acc + ORC_INSTANCE_WORDS(lg);
    fe h0[2], w; memcpy(h0, piV, 64); load_fe(&w, piV + 20);
    jac U0; load_jac(&U0, piV + 8);
    jac C_bar_p, C_bar; fe z_p, z, v; acc_hpolys hs;
    int rc = common_subroutine(pp, d, qs, m, h0, &U0, &w, &C_bar_p, &z_p, &hs);
    if (rc) { acc_hpolys_free(&hs); return rc; }
    load_jac(&C_bar, acc); load_fe(&z, acc + 13); load_fe(&v, acc + 17);
    fe hz; acc_h_eval(&hz, &hs, &z);
    acc_hpolys_free(&hs);
    if (!jac_eq(&C_bar_p, &C_bar)) return fail("C_bar' != C_bar");
    if (!fe_eq(&z_p, &z)) return fail("z' != z");
    if (acc[12] != d) return fail("d' != d");
    if (!fe_eq(&hz, &v)) return fail("h(z) != v");
    return 0;
}
/* acc.rs:245-255 */
int orc_acc_decider(const orc_pp *pp, const u64 *acc) {
    ensure_init();
    jac C; fe z, v; load_jac(&C, acc); load_fe(&z, acc + 13); load_fe(&v, acc + 17);
    return pcdl_check(pp, &C, (size_t)acc[12], &z, &v, acc + 21);
}
/* benches/acc.rs:15-29 */
int orc_random_instance(const orc_pp *pp, u64 *rng, size_t d, u64 *inst) {
    ensure_init();
    if (!is_pow2(d + 1) || d < 2) return fail("random_instance: bad d");
    size_t lg = ilog2(d + 1);
    size_t lo = d / 2, d_prime = lo + (size_t)(orc_rng_u64(rng) % (u64)(d - lo));
    if (d_prime == 0) d_prime = 1;
    fe w, z, v; orc_rng_scalar(rng, w.l);
    u64 *p = (u64 *)malloc(32 * (d_prime + 1));
    orc_rng_scalars(rng, d_prime + 1, p);
    jac C; int rc = pcdl_commit(&C, pp, p, d_prime + 1, d, &w);
    if (rc) { free(p); return rc; }
    orc_rng_scalar(rng, z.l);
    poly_eval(&v, p, d_prime + 1, &z);
    memset(inst, 0, 8 * ORC_INSTANCE_WORDS(lg));
    store_jac(inst, &C); inst[12] = d; store_fe(inst + 13, &z); store_fe(inst + 17, &v);
    u64 Cw[12]; memcpy(Cw, inst, 96);
    rc = orc_pcdl_open(pp, rng, p, d_prime + 1, Cw, d, z.l, w.l, inst + 21);
    free(p); return rc;
}
