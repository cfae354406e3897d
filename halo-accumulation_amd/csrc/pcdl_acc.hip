// Host side above the kernels: the reference's `pedersen`, `pcdl` and `acc` modules with the
// same names, argument meaning and error behaviour (pedersen.rs:6-20, pcdl.rs:99-342,
// acc.rs:135-255), every linear-time step running in HIP.  The reference is Rust and there is
// no Rust toolchain in this image, so this layer is C++ behind a C ABI; INTEGRATION.md shows
// the Rust shim that would sit on the group.rs-level entry points instead.
//
// What stays on the host, exactly as in the reference: Fiat-Shamir hashing (rho_0!/rho_1!),
// challenge inversion, the O(lg n) succinct check and the struct packing.
#include <atomic>
#include <functional>
#include <memory>
#include <thread>

#include "internal.hpp"

namespace halo {

using host::Fr;
using host::Point;
using host::Transcript;

// ---- flat layouts (include/halo_accumulation.h, "pcdl / acc level")
static size_t proof_words(size_t lg) { return 2 + 24 * lg + 32; }
static size_t instance_words(size_t lg) { return 21 + proof_words(lg); }
static size_t acc_words(size_t lg) { return instance_words(lg) + 24; }
static uint64_t *pf_L(uint64_t *pf, size_t i) { return pf + 2 + 12 * i; }
static uint64_t *pf_R(uint64_t *pf, size_t lg, size_t i) { return pf + 2 + 12 * lg + 12 * i; }
static uint64_t *pf_U(uint64_t *pf, size_t lg) { return pf + 2 + 24 * lg; }
static uint64_t *pf_c(uint64_t *pf, size_t lg) { return pf + 2 + 24 * lg + 12; }
static uint64_t *pf_Cbar(uint64_t *pf, size_t lg) { return pf + 2 + 24 * lg + 16; }
static uint64_t *pf_wp(uint64_t *pf, size_t lg) { return pf + 2 + 24 * lg + 28; }

static bool is_pow2(size_t n) { return n && !(n & (n - 1)); }
static size_t ilog2(size_t n) { size_t l = 0; while (n > 1) { n >>= 1; ++l; } return l; }
static int fail_assert(const char *m) { set_error(m); return HALO_E_ASSERT; }
static int fail_reject(const char *m) { set_error(m); return HALO_E_REJECT; }

struct PublicPoints { Point S, H; };
static const PublicPoints &public_points() {  // main.rs:35-45 / consts.rs:26-65
    static PublicPoints pp = [] {
        Point g = Point::generator();
        return PublicPoints{g.mul(host::urs_scalar(0)).normalized(), g.mul(host::urs_scalar(1)).normalized()};
    }();
    return pp;
}

static Fr rho0_C_z_v(const Point &C, const Fr &z, const Fr &v) {
    Transcript t; t.point(C); t.scalar(z); t.scalar(v); return t.finish(0);
}
static Fr rho0_C_z_v_Cbar(const Point &C, const Fr &z, const Fr &v, const Point &Cb) {
    Transcript t; t.point(C); t.scalar(z); t.scalar(v); t.point(Cb); return t.finish(0);
}
static Fr rho0_xi_L_R(const Fr &xi, const Point &L, const Point &R) {
    Transcript t; t.scalar(xi); t.point(L); t.point(R); return t.finish(0);
}

static int ensure_poly_buffers(halo_ctx *ctx) {
    size_t n = ctx->n < 64 ? 64 : ctx->n;
    if (!ctx->d_poly || !ctx->d_poly2) alloc_epoch_bump(ctx);
    if (!ctx->d_poly) HALO_HIP(hipMalloc(&ctx->d_poly, n * 32));
    if (!ctx->d_poly2) HALO_HIP(hipMalloc(&ctx->d_poly2, n * 32));
    return HALO_OK;
}

// DensePolynomial::degree: the index of the last non-zero coefficient (0 for the zero polynomial).  Scanned from the END: a
// dense polynomial answers at its first look (the forward scan this replaces read all 32 MiB of a 2^20-coefficient polynomial
// on the host, ~1 ms of every halo_pcdl_open / halo_pcdl_commit with host coefficients).
static size_t host_poly_degree(const uint64_t *coeffs, size_t len) {
    for (size_t i = len; i-- > 0;)
        if (coeffs[4 * i] | coeffs[4 * i + 1] | coeffs[4 * i + 2] | coeffs[4 * i + 3]) return i;
    return 0;
}

// pedersen::commit over GS[0..n) for device-resident, zero-padded scalars (pedersen.rs:6-20)
static int pedersen_commit_dev(halo_ctx *ctx, const Fr *w, const uint64_t *d_ms, size_t n, Point *out) {
    Point acc;
    int rc = msm_run(ctx, ctx->d_bases, d_ms, true, n, &acc);
    if (rc) return rc;
    if (w) acc = public_s_table().mul(*w) + acc;
    *out = acc;
    return HALO_OK;
}

// pcdl::commit for a short host polynomial (acc.rs:153,195: two coefficients): same point, no launch
static int commit_short_host(halo_ctx *ctx, const uint64_t *coeffs, size_t len, const Fr *w, Point *out) {
    std::vector<uint64_t> bases(8 * len);
    int rc = halo_ctx_read_bases(ctx, 0, len, bases.data());
    if (rc) return rc;
    std::vector<Point> pts(len);
    std::vector<Fr> ks(len);
    for (size_t i = 0; i < len; ++i) { pts[i] = Point::load_affine(&bases[8 * i]); ks[i] = Fr::load(coeffs + 4 * i); }
    Point acc = host::small_msm(pts, ks);
    if (w) acc = public_s_table().mul(*w) + acc;
    *out = acc;
    return HALO_OK;
}

// pcdl.rs:99-110
static int pcdl_commit_host(halo_ctx *ctx, const uint64_t *coeffs, size_t len, size_t d, const Fr *w, Point *out) {
    size_t n = d + 1;
    if (!is_pow2(n)) return fail_assert("commit: d + 1 is not a power of two");
    size_t deg = host_poly_degree(coeffs, len);
    if (deg > d) return fail_assert("commit: p.degree() > d");
    if (d + 1 > ctx->n) return fail_assert("commit: d > D");
    size_t used = deg + 1 < len ? deg + 1 : len;
    if (used <= 16) return commit_short_host(ctx, coeffs, used, w, out);
    // pedersen::commit over GS[0..n) (pedersen.rs:6-20) with the coefficients still in host memory: halo_msm's path (abi.hip
    // msm_host_run: the copy in stretches under the kernels where that pays), zero-padded beyond the polynomial
    Point acc;
    int rc = msm_host_run(ctx, 0, n, coeffs, used, 1, &acc);
    if (rc) return rc;
    if (w) acc = public_s_table().mul(*w) + acc;
    *out = acc;
    return HALO_OK;
}

// pcdl.rs:120-242 on a device-resident polynomial (d_poly: n coefficients, zero padded, clobbered)
static int pcdl_open_dev(halo_ctx *ctx, host::Rng *rng, size_t deg, const Point &C, size_t d, const Fr &z, const Fr *w,
                         uint64_t *proof) {
    size_t n = d + 1, lg_n = ilog2(n);
    std::memset(proof, 0, 8 * proof_words(lg_n));
    proof[1] = lg_n;
    Fr v;
    int rc = fr_poly_eval(ctx, ctx->d_poly, deg + 1, z, &v);  // :135
    if (rc) return rc;
    Point C_prime = C;
    if (w) {
        if (deg == 0) return fail_assert("open: hiding needs p.degree() >= 1");  // usize underflow at :141
        // :140-142  q uniform of degree deg-1, p_bar = q (X - z)
        rc = rng_scalars_dev(ctx, rng->state, deg, ctx->d_tmp_a);
        if (rc) return rc;
        rng->state += 4 * (uint64_t)deg * 0x9E3779B97F4A7C15ULL;
        HALO_HIP(hipMemsetAsync(ctx->d_poly2, 0, n * 32, ctx->stream));
        rc = pbar_dev(ctx, ctx->d_tmp_a, deg, z, ctx->d_poly2);
        if (rc) return rc;
        Fr w_bar = rng->scalar();  // :147
        Point C_bar;
        rc = pedersen_commit_dev(ctx, &w_bar, ctx->d_poly2, n, &C_bar);  // :150
        if (rc) return rc;
        Fr a = rho0_C_z_v_Cbar(C, z, v, C_bar);                            // :153
        rc = axpy_dev(ctx, ctx->d_poly, ctx->d_poly2, deg + 1, a);         // :156
        if (rc) return rc;
        Fr w_prime = w_bar * a + *w;                                       // :159
        C_prime = C + C_bar.mul(a) - public_s_table().mul(w_prime);                    // :162
        proof[0] = 1;
        C_bar.store_normalized(pf_Cbar(proof, lg_n));
        w_prime.store(pf_wp(proof, lg_n));
    } else {
        Point::infinity().store(pf_Cbar(proof, lg_n));
    }
    Fr xi = rho0_C_z_v(C_prime, z, v);  // :180
    Point Hp = public_h_table().mul(xi).normalized();  // :181
    uint64_t Hp_w[12];
    Hp.store(Hp_w);
    halo_ipa *st = nullptr;
    rc = ipa_begin_dev(ctx, n, ctx->d_poly, z, &st);  // :183-186
    if (rc) return rc;
    std::unique_ptr<halo_ipa, void (*)(halo_ipa *)> guard(st, halo_ipa_destroy);
    ipa_set_hprime_scalar(st, xi);  // H' = xi_0 H: the rounds take k H' = (k xi_0) H from the process-wide table of H
    for (size_t round = 0; round < lg_n; ++round) {
        uint64_t *Lw = pf_L(proof, round), *Rw = pf_R(proof, lg_n, round);
        rc = halo_ipa_round_lr(st, Hp_w, Lw, Rw);  // :203-208
        if (rc) return rc;
        Fr xi_next = rho0_xi_L_R(xi, Point::load(Lw), Point::load(Rw));  // :212
        if (xi_next.is_zero()) return fail_assert("open: challenge is zero (inverse().unwrap())");
        Fr xi_inv = xi_next.inv();  // :213
        xi = xi_next;
        uint64_t xw[4], xiw[4];
        xi.store(xw);
        xi_inv.store(xiw);
        rc = halo_ipa_round_fold(st, xw, xiw);  // :216-224
        if (rc) return rc;
    }
    return halo_ipa_finish(st, pf_U(proof, lg_n), pf_c(proof, lg_n));  // :230-231
}

static bool scalar_ok(const Fr &s) { return !Fr::geq(s.l, host::FrP::M); }
// every point of the blob on the curve, every scalar canonical, flag word 0 or 1
static bool proof_wellformed(uint64_t *proof, size_t lg_n) {
    if (proof[0] > 1) return false;
    for (size_t i = 0; i < lg_n; ++i)
        if (!Point::load(pf_L(proof, i)).on_curve() || !Point::load(pf_R(proof, lg_n, i)).on_curve()) return false;
    if (!Point::load(pf_U(proof, lg_n)).on_curve() || !scalar_ok(Fr::load(pf_c(proof, lg_n)))) return false;
    if (proof[0] && (!Point::load(pf_Cbar(proof, lg_n)).on_curve() || !scalar_ok(Fr::load(pf_wp(proof, lg_n))))) return false;
    return true;
}

// pcdl.rs:252-314 in two halves.  challenges(): everything the transcript decides -- C', xi_0 .. xi_lg (pcdl.rs:272-296);
// this is all pcdl::check's linear-time half (h.get_poly + the MSM, pcdl.rs:338) needs, so that half is launched
// before relation() runs on the host.  relation(): the 2 lg n + O(1) scalar multiplications as one interleaved
// host MSM and the final comparison (pcdl.rs:288-310).
struct SuccinctState {
    size_t lg_n = 0;
    Point C_prime, Hp, U;
    std::vector<Fr> xis;
};
// key_n: the size of the key the check is against (ctx->n; a rank's cyclic shard stands for stride * ctx->n points)
static int succinct_challenges(halo_ctx *ctx, const Point &C, size_t d, const Fr &z, const Fr &v, const uint64_t *proof_c, SuccinctState *st,
                               bool need_hp = true, size_t key_n = 0) {
    uint64_t *proof = const_cast<uint64_t *>(proof_c);
    size_t n = d + 1;
    if (!is_pow2(n)) return fail_reject("d+1 is not a power of 2!");
    if (d + 1 > (key_n ? key_n : ctx->n)) return fail_reject("d was larger than D!");
    size_t lg_n = ilog2(n);
    if (proof[1] != lg_n) return fail_reject("proof length does not match d");
    // Everything below reads exactly proof_words(lg_n) words.  The blob is verifier input: the reference's typed
    // `PallasPoint`/`PallasScalar` values are on the curve / below the modulus by construction, here that is checked.
    if (!C.on_curve() || !scalar_ok(z) || !scalar_ok(v)) return fail_reject("instance holds an invalid point or scalar");
    if (!proof_wellformed(proof, lg_n)) return fail_reject("proof holds an invalid point or scalar");
    st->lg_n = lg_n;
    st->C_prime = C;
    if (proof[0]) {
        Point C_bar = Point::load(pf_Cbar(proof, lg_n));
        Fr wp = Fr::load(pf_wp(proof, lg_n));
        Fr a = rho0_C_z_v_Cbar(C, z, v, C_bar);
        st->C_prime = C + C_bar.mul(a) - public_s_table().mul(wp);
    }
    st->xis.assign(lg_n + 1, Fr::zero());
    st->xis[0] = rho0_C_z_v(st->C_prime, z, v);
    if (need_hp) st->Hp = public_h_table().mul(st->xis[0]);  // the batched relation multiplies H by (v - v') xi_0 instead
    for (size_t i = 0; i < lg_n; ++i) {
        st->xis[i + 1] = rho0_xi_L_R(st->xis[i], Point::load(pf_L(proof, i)), Point::load(pf_R(proof, lg_n, i)));
        if (st->xis[i + 1].is_zero()) return fail_reject("challenge is zero");
    }
    st->U = Point::load(pf_U(proof, lg_n));
    return HALO_OK;
}
static int succinct_relation(const SuccinctState &st, const Fr &z, const Fr &v, const uint64_t *proof_c) {
    uint64_t *proof = const_cast<uint64_t *>(proof_c);
    size_t lg_n = st.lg_n;
    const std::vector<Fr> &xis = st.xis;
    std::vector<Point> pts;
    std::vector<Fr> ks;
    pts.reserve(2 * lg_n + 1);
    ks.reserve(2 * lg_n + 1);
    for (size_t i = 0; i < lg_n; ++i) {
        pts.push_back(Point::load(pf_L(proof, i))); ks.push_back(xis[i + 1]);  // scalar replaced by its inverse below
        pts.push_back(Point::load(pf_R(proof, lg_n, i))); ks.push_back(xis[i + 1]);
    }
    // one inversion for all challenges (Montgomery's trick)
    {
        std::vector<Fr> pref(lg_n + 1, Fr::one());
        for (size_t i = 0; i < lg_n; ++i) pref[i + 1] = pref[i] * xis[i + 1];
        Fr inv = lg_n ? pref[lg_n].inv() : Fr::one();
        for (size_t i = lg_n; i-- > 0;) {
            ks[2 * i] = inv * pref[i];  // xi_{i+1}^-1
            inv = inv * xis[i + 1];
        }
    }
    pts.push_back(st.Hp); ks.push_back(v);
    // :288-298.  2 lg n + 1 scalar multiples (~1.1 ms on one thread at lg n = 20): four interleaved-window sums on the host
    // pool, added in order (the verifier's two instances run side by side: eight threads for ~0.35 ms)
    Point C_i = st.C_prime;
    {
        const size_t ways = pts.size() >= 16 ? 4 : 1, per = (pts.size() + ways - 1) / ways;
        std::vector<Point> part(ways, Point::infinity());
        pool_run(ways, [&](size_t k) {
            size_t lo = k * per, hi = lo + per < pts.size() ? lo + per : pts.size();
            if (lo >= hi) return;
            part[k] = host::small_msm(std::vector<Point>(pts.begin() + lo, pts.begin() + hi), std::vector<Fr>(ks.begin() + lo, ks.begin() + hi));
        });
        for (size_t k = 0; k < ways; ++k) C_i = C_i + part[k];
    }
    // :301-304  v' = c * h(z)
    Fr c = Fr::load(pf_c(proof, lg_n));
    Fr hz = Fr::one() + xis[lg_n] * z, zi = z;
    for (size_t i = 1; i < lg_n; ++i) { zi = zi.sqr(); hz = hz * (Fr::one() + xis[lg_n - i] * zi); }
    Fr v_prime = c * hz;
    std::vector<Point> p2{st.U, st.Hp};
    std::vector<Fr> k2{c, v_prime};
    if (C_i != host::small_msm(p2, k2)) return fail_reject("C_(log_n) != CM.Commit_Sigma(c || v')");  // :307-310
    return HALO_OK;
}
static int succinct_check_host(halo_ctx *ctx, const Point &C, size_t d, const Fr &z, const Fr &v, const uint64_t *proof_c,
                               std::vector<Fr> *xis_out, Point *U_out) {
    SuccinctState st;
    int rc = succinct_challenges(ctx, C, d, z, v, proof_c, &st);
    if (!rc) rc = succinct_relation(st, z, v, proof_c);
    if (rc) return rc;
    *xis_out = std::move(st.xis);
    *U_out = st.U;
    return HALO_OK;
}

// ------------------------------------------------------------------ batched succinct checks (acc.rs:158-170)
// m instances at once: the transcripts (hashes, C') are computed by a bounded pool of host threads, then ONE launch evaluates
// the m polynomials h_i at their own z_i (k_h_eval_z) and ONE launch computes the m relations (k_batch_small_msm):
//     C'_i + sum_j (xi_j^-1 L_j + xi_j R_j) + (v_i - c_i h_i(z_i)) xi_0 H - c_i U_i  ==  0        (pcdl.rs:288-310)
// as 2 lg n + 2 scalar multiples per instance, compared with -C'_i on the host.  Per-instance outcome equals the host path's.
constexpr size_t kBatchVerifyMin = 64;  // below this the host pool is faster than a 256-step device ladder (~2 ms)
static int verify_staging(halo_ctx *ctx, size_t words) {
    if (words <= ctx->verify_words) return HALO_OK;
    alloc_epoch_bump(ctx);
    (void)hipFree(ctx->d_verify);
    ctx->d_verify = nullptr;
    ctx->verify_words = 0;
    HALO_HIP(hipMalloc(&ctx->d_verify, words * 8));
    ctx->verify_words = words;
    return HALO_OK;
}
struct BatchCheck { int rc = HALO_OK; std::string err; SuccinctState st; };
// instances: m blobs at stride instance_words(lg(d+1)); res[i].rc / .err / .st filled; returns a device / argument error only
static int succinct_check_batch(halo_ctx *ctx, size_t d, const uint64_t *qs, size_t m, std::vector<BatchCheck> &res) {
    size_t lg = ilog2(d + 1), iw = instance_words(lg), K = 2 * lg + 2;
    res.assign(m, BatchCheck());
    if (K > 64) { set_error("batched succinct check: lg n too large"); return HALO_E_ARG; }
    pool_run(m, [&](size_t i) {
        const uint64_t *q = qs + i * iw;
        res[i].rc = succinct_challenges(ctx, Point::load(q), (size_t)q[12], Fr::load(q + 13), Fr::load(q + 17), q + 21, &res[i].st, false);
        if (res[i].rc) res[i].err = halo_last_error();
    });
    // staging layout (words): xis m (lg+1) 4 | zs m 4 | hz m 4 | points m K 8 | scalars m K 4 | out m 12
    size_t o_xis = 0, o_zs = o_xis + m * (lg + 1) * 4, o_hz = o_zs + m * 4, o_pts = o_hz + m * 4, o_sc = o_pts + m * K * 8, o_out = o_sc + m * K * 4,
           total = o_out + m * 12;
    int rc = verify_staging(ctx, total);
    if (rc) return rc;
    std::vector<uint64_t> host(total, 0);
    for (size_t i = 0; i < m; ++i) {
        if (res[i].rc) continue;  // a rejected transcript: its (zero) rows are computed and ignored
        for (size_t k = 0; k <= lg; ++k) res[i].st.xis[k].store(&host[o_xis + (i * (lg + 1) + k) * 4]);
        std::memcpy(&host[o_zs + 4 * i], qs + i * iw + 13, 32);
    }
    HALO_HIP(hipMemcpyAsync(ctx->d_verify, host.data(), (o_hz) * 8, hipMemcpyHostToDevice, ctx->stream));
    rc = h_eval_each(ctx, ctx->d_verify + o_xis, ctx->d_verify + o_zs, m, lg, ctx->d_verify + o_hz);
    if (rc) return rc;
    HALO_HIP(hipMemcpyAsync(&host[o_hz], ctx->d_verify + o_hz, m * 32, hipMemcpyDeviceToHost, ctx->stream));
    HALO_HIP(hipStreamSynchronize(ctx->stream));
    const PublicPoints &pp = public_points();
    host::Affine Ha = pp.H.to_affine();
    pool_run(m, [&](size_t i) {
        if (res[i].rc) return;
        const uint64_t *q = qs + i * iw;
        uint64_t *proof = const_cast<uint64_t *>(q + 21);
        const SuccinctState &st = res[i].st;
        uint64_t *pts = &host[o_pts + i * K * 8], *sc = &host[o_sc + i * K * 4];
        auto put_point = [&](size_t slot, const Point &p) {
            host::Affine a = p.to_affine();
            if (!a.inf) { a.x.store(pts + 8 * slot); a.y.store(pts + 8 * slot + 4); }
        };
        // challenge inverses with one inversion (Montgomery's trick)
        std::vector<Fr> pref(lg + 1, Fr::one()), inv(lg + 1);
        for (size_t j = 0; j < lg; ++j) pref[j + 1] = pref[j] * st.xis[j + 1];
        Fr run = lg ? pref[lg].inv() : Fr::one();
        for (size_t j = lg; j-- > 0;) { inv[j + 1] = run * pref[j]; run = run * st.xis[j + 1]; }
        for (size_t j = 0; j < lg; ++j) {
            put_point(j, Point::load(pf_L(proof, j)));
            inv[j + 1].from_mont().store(sc + 4 * j);
            put_point(lg + j, Point::load(pf_R(proof, lg, j)));
            st.xis[j + 1].from_mont().store(sc + 4 * (lg + j));
        }
        Fr c = Fr::load(pf_c(proof, lg)), v = Fr::load(q + 17), hz = Fr::load(&host[o_hz + 4 * i]);
        Ha.x.store(pts + 8 * (2 * lg)); Ha.y.store(pts + 8 * (2 * lg) + 4);
        ((v - c * hz) * st.xis[0]).from_mont().store(sc + 4 * (2 * lg));      // (v - v') xi_0 on H: the two H' terms of :288 and :307
        put_point(2 * lg + 1, st.U);
        (-c).from_mont().store(sc + 4 * (2 * lg + 1));
    });
    HALO_HIP(hipMemcpyAsync(ctx->d_verify + o_pts, &host[o_pts], (o_out - o_pts) * 8, hipMemcpyHostToDevice, ctx->stream));
    rc = batch_small_msm(ctx, ctx->d_verify + o_pts, ctx->d_verify + o_sc, m, K, ctx->d_verify + o_out);
    if (rc) return rc;
    HALO_HIP(hipMemcpyAsync(&host[o_out], ctx->d_verify + o_out, m * 96, hipMemcpyDeviceToHost, ctx->stream));
    HALO_HIP(hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < m; ++i) {
        if (res[i].rc) continue;
        if (Point::load(&host[o_out + 12 * i]) != -res[i].st.C_prime) {
            res[i].rc = HALO_E_REJECT;
            res[i].err = "C_(log_n) != CM.Commit_Sigma(c || v')";  // :307-310
        }
    }
    return HALO_OK;
}

// pcdl.rs:323-342.  The device half (h coefficients + the n-point MSM, :338) runs while the host evaluates the
// succinct relation; errors are reported in the reference's order (succinct check first).
static int pcdl_check_host(halo_ctx *ctx, const Point &C, size_t d, const Fr &z, const Fr &v, const uint64_t *proof) {
    SuccinctState st;
    int rc = succinct_challenges(ctx, C, d, z, v, proof, &st, false);  // (H' = xi_0 H: only the relation needs it -- below, under the MSM)
    if (rc) return rc;
    size_t lg_n = st.lg_n, n = d + 1;
    rc = h_coeffs_dev(ctx, st.xis.data(), lg_n, Fr::one(), false, ctx->d_tmp_a);  // h.get_poly().coeffs
    if (rc) return rc;
    {
        BorrowScope scope(ctx);  // (nothing else of the caller's is in flight during a check: a large MSM may use slot 1 as well)
        rc = msm_enqueue(ctx, 0, ctx->d_bases, ctx->d_tmp_a, true, n);  // :338, asynchronous
    }
    if (rc) return rc;
    st.Hp = public_h_table().mul(st.xis[0]);
    int rc_rel = succinct_relation(st, z, v, proof);
    std::string rel_err = rc_rel ? halo_last_error() : "";
    Point comm;
    rc = msm_finish(ctx, 0, &comm);
    if (rc_rel) { set_error(rel_err); return rc_rel; }
    if (rc) return rc;
    if (st.U != comm) return fail_reject("U != CM.Commit(ck, h_vec)");  // :339
    return HALO_OK;
}

// One rank's half of pcdl::check when the key is sharded cyclically (point i on rank i mod P): the succinct check (host
// arithmetic, the same on every rank) and this rank's share of CM.Commit(ck, h) (:338).  The coefficient of X^(r + j P) is
//     prod_{i < p, bit i of r} xi_(lg n - i)  *  prod_{bit i' of j} xi_(lg n - p - i')            (P = 2^p, pcdl.rs:56-77)
// i.e. a constant of the rank times the j-th coefficient of the h polynomial of the first lg n - p challenges: the shard's
// scalars are h_coeffs_dev over lg n - p variables scaled by that constant.  The caller adds the P shares and compares
// with U (:339).
static int pcdl_check_partial_host(halo_ctx *ctx, const Point &C, size_t d, const Fr &z, const Fr &v, const uint64_t *proof, uint64_t stride,
                                   uint64_t offset, Point *U_out, Point *part_out) {
    if (stride == 0 || !is_pow2(stride) || offset >= stride) { set_error("check_partial: stride must be a power of two, offset below it"); return HALO_E_ARG; }
    if (ctx->n == 0 || !is_pow2(ctx->n)) { set_error("check_partial: the shard's key must hold a power of two of points"); return HALO_E_ARG; }
    SuccinctState st;
    int rc = succinct_challenges(ctx, C, d, z, v, proof, &st, true, ctx->n * stride);
    if (rc) return rc;
    size_t lg_n = st.lg_n, n = d + 1, p = ilog2((size_t)stride);
    if (n < stride) { set_error("check_partial: fewer coefficients than ranks"); return HALO_E_ARG; }
    size_t n_local = n / stride;  // <= ctx->n
    Fr scale = Fr::one();
    for (size_t i = 0; i < p; ++i)
        if ((offset >> i) & 1) scale = scale * st.xis[lg_n - i];
    rc = h_coeffs_dev(ctx, st.xis.data(), lg_n - p, scale, false, ctx->d_tmp_a);
    if (rc) return rc;
    rc = msm_enqueue(ctx, 0, ctx->d_bases, ctx->d_tmp_a, true, n_local);  // asynchronous: the relation runs on the host meanwhile
    if (rc) return rc;
    int rc_rel = succinct_relation(st, z, v, proof);
    std::string rel_err = rc_rel ? halo_last_error() : "";
    Point part;
    rc = msm_finish(ctx, 0, &part);
    if (rc_rel) { set_error(rel_err); return rc_rel; }
    if (rc) return rc;
    *U_out = st.U;
    *part_out = part;
    return HALO_OK;
}

// ------------------------------------------------------------------ acc.rs
struct AccHPolys {  // acc.rs:61-66
    Fr h0[2];
    std::vector<std::vector<Fr>> xis;
    Fr alpha;
    std::vector<Fr> alphas;  // alpha^0 .. alpha^m
    size_t lg_n = 0;

    Fr eval(const Fr &z) const {  // acc.rs:97-106
        Fr v = h0[0] + h0[1] * z;
        for (size_t i = 0; i < xis.size(); ++i) {
            const std::vector<Fr> &x = xis[i];
            Fr hz = Fr::one() + x[lg_n] * z, zi = z;
            for (size_t k = 1; k < lg_n; ++k) { zi = zi.sqr(); hz = hz * (Fr::one() + x[lg_n - k] * zi); }
            v = v + hz * alphas[i + 1];
        }
        return v;
    }
};

// acc.rs:135-188
static int common_subroutine(halo_ctx *ctx, size_t d, const uint64_t *qs, size_t m, const Fr h0[2], const Point &U0, const Fr &w,
                             Point *C_bar_out, Fr *z_out, AccHPolys *hs) {
    if (!is_pow2(d + 1)) return fail_reject("d+1 is not a power of 2!");
    size_t lg = ilog2(d + 1), iw = instance_words(lg);
    hs->h0[0] = h0[0];
    hs->h0[1] = h0[1];
    hs->lg_n = lg;
    std::vector<Point> Us{U0};
    uint64_t h0w[8];
    h0[0].store(h0w);
    h0[1].store(h0w + 4);
    Point chk;
    int rc = pcdl_commit_host(ctx, h0w, 2, d, nullptr, &chk);  // :152-155
    if (rc) return rc;
    if (U0 != chk) return fail_reject("U_0 != PCDL.Commit(h_0)");
    // :158-170  the m succinct checks are independent host work (hashing + a 2 lg n + 1 point MSM
    // each): one thread per instance; errors are reported in instance order like the serial loop
    struct CheckResult { int rc = HALO_OK; std::string err; std::vector<Fr> xis; Point U; };
    std::vector<CheckResult> res(m);
    // The instances are laid out with the stride of degree d.  An instance that claims another degree (or whose
    // proof claims another length) would be read past its end: it is rejected here, before anything is parsed --
    // the reference fails on it too (acc.rs:169, after its typed, bounds-safe succinct check).
    for (size_t i = 0; i < m; ++i) {
        const uint64_t *q = qs + i * iw;
        if ((size_t)q[12] != d || q[22] != lg) return fail_reject("d_i != d");  // :169
    }
    auto run_one = [&](size_t i) {
        const uint64_t *q = qs + i * iw;
        res[i].rc = succinct_check_host(ctx, Point::load(q), (size_t)q[12], Fr::load(q + 13), Fr::load(q + 17), q + 21, &res[i].xis,
                                        &res[i].U);  // :164
        if (res[i].rc) res[i].err = halo_last_error();
    };
    if (m >= kBatchVerifyMin && ctx->batch_verify) {  // the relations of all instances in two launches
        std::vector<BatchCheck> bres;
        rc = succinct_check_batch(ctx, d, qs, m, bres);
        if (rc) return rc;
        for (size_t i = 0; i < m; ++i) {
            res[i].rc = bres[i].rc;
            res[i].err = std::move(bres[i].err);
            res[i].xis = std::move(bres[i].st.xis);
            res[i].U = bres[i].st.U;
        }
    } else if (m <= 1) {
        for (size_t i = 0; i < m; ++i) run_one(i);
    } else {  // a bounded pool: at most 16 host threads pull instances off a shared counter
        pool_run(m, run_one);
    }
    for (size_t i = 0; i < m; ++i) {
        if (res[i].rc) { set_error(res[i].err); return res[i].rc; }
        hs->xis.push_back(std::move(res[i].xis));
        Us.push_back(res[i].U);
    }
    // :173  alpha = rho_1(hs): h_0 Some(poly), hs Vec<HPoly>, alpha None, alphas empty
    Transcript t;
    size_t h0len = h0[1].is_zero() ? (h0[0].is_zero() ? 0 : 1) : 2;
    t.byte(1); t.u64le(h0len);
    for (size_t k = 0; k < h0len; ++k) t.scalar(h0[k]);
    t.u64le(m);
    for (size_t i = 0; i < m; ++i) { t.u64le(lg + 1); for (size_t k = 0; k <= lg; ++k) t.scalar(hs->xis[i][k]); }
    t.byte(0); t.u64le(0);
    hs->alpha = t.finish(1);
    hs->alphas.assign(m + 1, Fr::one());
    for (size_t i = 1; i <= m; ++i) hs->alphas[i] = hs->alphas[i - 1] * hs->alpha;
    Point C = host::small_msm(Us, hs->alphas);  // :178  (m + 1 points)
    Transcript t2;
    t2.point(C); t2.scalar(hs->alpha);
    *z_out = t2.finish(1);                       // :181
    *C_bar_out = C + public_s_table().mul(w);   // :184
    return HALO_OK;
}

}  // namespace halo

using namespace halo;

#define HALO_CTX2(ctx)                                                   \
    do {                                                                 \
        if (!(ctx)) { halo::set_error("null context"); return HALO_E_ARG; } \
        hipError_t _e = hipSetDevice((ctx)->device);                     \
        if (_e != hipSuccess) return halo::hip_fail(_e, "hipSetDevice"); \
    } while (0)

extern "C" {

size_t halo_proof_words(size_t lg_n) { return proof_words(lg_n); }
size_t halo_instance_words(size_t lg_n) { return instance_words(lg_n); }
size_t halo_accumulator_words(size_t lg_n) { return acc_words(lg_n); }

int halo_pedersen_commit(halo_ctx *ctx, const uint64_t *w, size_t n_bases, const uint64_t *ms, size_t n_ms, uint64_t out[12]) {
    HALO_CTX2(ctx);
    if (n_bases != n_ms) return fail_assert("Length did not match for pedersen commitment");  // pedersen.rs:7-12
    if (n_bases > ctx->n) return fail_assert("pedersen commit: more bases than the key holds");
    if (!out || (n_ms && !ms)) { set_error("pedersen commit: null pointer"); return HALO_E_ARG; }
    // pedersen.rs:14-17 with the scalars in host memory: the host-scalar MSM path of halo_msm (abi.hip msm_host_run)
    Point r;
    int rc = msm_host_run(ctx, 0, n_ms, ms, n_ms, 1, &r);
    if (rc) return rc;
    if (w) r = public_s_table().mul(Fr::load(w)) + r;
    r.store_normalized(out);
    return HALO_OK;
}

int halo_pedersen_commit_affine(halo_ctx *ctx, const uint64_t *w, const uint64_t *bases_affine, size_t n_bases, const uint64_t *ms,
                                size_t n_ms, uint64_t out[12]) {
    HALO_CTX2(ctx);
    if (n_bases != n_ms) return fail_assert("Length did not match for pedersen commitment");  // pedersen.rs:7-12
    uint64_t acc_w[12];
    int rc = halo_msm_affine(ctx, bases_affine, ms, n_ms, 1, acc_w);  // pedersen.rs:14 over the caller's generators
    if (rc) return rc;
    Point acc = Point::load(acc_w);
    if (w) acc = public_s_table().mul(Fr::load(w)) + acc;  // pedersen.rs:15-17
    acc.store_normalized(out);
    return HALO_OK;
}

int halo_pcdl_commit(halo_ctx *ctx, const uint64_t *coeffs, size_t len, size_t d, const uint64_t *w, uint64_t out[12]) {
    HALO_CTX2(ctx);
    Fr wf = w ? Fr::load(w) : Fr::zero();
    Point r;
    int rc = pcdl_commit_host(ctx, coeffs, len, d, w ? &wf : nullptr, &r);
    if (rc) return rc;
    r.store_normalized(out);
    return HALO_OK;
}

int halo_pcdl_open(halo_ctx *ctx, uint64_t *rng_state, const uint64_t *coeffs, size_t len, const uint64_t C[12], size_t d,
                   const uint64_t z[4], const uint64_t *w, uint64_t *proof_out) {
    HALO_CTX2(ctx);
    size_t n = d + 1;
    if (!is_pow2(n)) return fail_assert("open: d + 1 is not a power of two");  // pcdl.rs:130
    size_t deg = host_poly_degree(coeffs, len);
    if (deg > d) return fail_assert("open: p.degree() > d");                   // pcdl.rs:131
    if (n > ctx->n) return fail_assert("open: d > D");                         // pcdl.rs:132
    int rc = ensure_poly_buffers(ctx);
    if (rc) return rc;
    HALO_HIP(hipMemsetAsync(ctx->d_poly, 0, n * 32, ctx->stream));
    rc = upload_words(ctx, ctx->d_poly, coeffs, (deg + 1) * 4);
    if (rc) return rc;
    host::Rng rng{rng_state ? *rng_state : 0};
    Fr wf = w ? Fr::load(w) : Fr::zero();
    rc = pcdl_open_dev(ctx, &rng, deg, Point::load(C), d, Fr::load(z), w ? &wf : nullptr, proof_out);
    if (rng_state) *rng_state = rng.state;
    return rc;
}

// pcdl::open for a polynomial that already lives in device memory (the coefficients are copied, not clobbered)
int halo_pcdl_open_dev(halo_ctx *ctx, uint64_t *rng_state, const void *d_coeffs, size_t len, const uint64_t C[12], size_t d,
                       const uint64_t z[4], const uint64_t *w, uint64_t *proof_out) {
    HALO_CTX2(ctx);
    size_t n = d + 1;
    if (!is_pow2(n)) return fail_assert("open: d + 1 is not a power of two");  // pcdl.rs:130
    if (len == 0 || !d_coeffs || !C || !z || !proof_out) { set_error("open_dev: null pointer or empty polynomial"); return HALO_E_ARG; }
    if (len - 1 > d) return fail_assert("open: p.degree() > d");                // pcdl.rs:131
    if (n > ctx->n) return fail_assert("open: d > D");                          // pcdl.rs:132
    int rc = ensure_poly_buffers(ctx);
    if (rc) return rc;
    HALO_HIP(hipMemcpyAsync(ctx->d_poly, d_coeffs, len * 32, hipMemcpyDeviceToDevice, ctx->stream));
    if (len < n) HALO_HIP(hipMemsetAsync(ctx->d_poly + 4 * len, 0, (n - len) * 32, ctx->stream));
    host::Rng rng{rng_state ? *rng_state : 0};
    Fr wf = w ? Fr::load(w) : Fr::zero();
    rc = pcdl_open_dev(ctx, &rng, len - 1, Point::load(C), d, Fr::load(z), w ? &wf : nullptr, proof_out);
    if (rng_state) *rng_state = rng.state;
    return rc;
}
// pcdl::commit for device-resident coefficients (len <= d + 1)
int halo_pcdl_commit_dev(halo_ctx *ctx, const void *d_coeffs, size_t len, size_t d, const uint64_t *w, uint64_t out[12]) {
    HALO_CTX2(ctx);
    size_t n = d + 1;
    if (!is_pow2(n)) return fail_assert("commit: d + 1 is not a power of two");
    if (!out || (len && !d_coeffs)) { set_error("commit_dev: null pointer"); return HALO_E_ARG; }
    if (len > n) return fail_assert("commit: p.degree() > d");
    if (n > ctx->n) return fail_assert("commit: d > D");
    int rc = ensure_poly_buffers(ctx);
    if (rc) return rc;
    if (len) HALO_HIP(hipMemcpyAsync(ctx->d_poly2, d_coeffs, len * 32, hipMemcpyDeviceToDevice, ctx->stream));
    if (len < n) HALO_HIP(hipMemsetAsync(ctx->d_poly2 + 4 * len, 0, (n - len) * 32, ctx->stream));
    Fr wf = w ? Fr::load(w) : Fr::zero();
    Point r;
    rc = pedersen_commit_dev(ctx, w ? &wf : nullptr, ctx->d_poly2, n, &r);
    if (rc) return rc;
    r.store_normalized(out);
    return HALO_OK;
}

int halo_pcdl_succinct_check(halo_ctx *ctx, const uint64_t C[12], size_t d, const uint64_t z[4], const uint64_t v[4],
                             const uint64_t *proof, uint64_t *xis_out, uint64_t U_out[12]) {
    HALO_CTX2(ctx);
    std::vector<Fr> xis;
    Point U;
    int rc = succinct_check_host(ctx, Point::load(C), d, Fr::load(z), Fr::load(v), proof, &xis, &U);
    if (rc) return rc;
    for (size_t i = 0; i < xis.size(); ++i) xis[i].store(xis_out + 4 * i);
    U.store_normalized(U_out);
    return HALO_OK;
}

// m succinct checks at once (instances at stride halo_instance_words(lg(d+1)), all of degree bound d): on the device from
// 64 instances on, on a pool of host threads below.  status[i] = 0 or HALO_E_REJECT; returns HALO_E_REJECT (message names the
// first rejected instance) if any was rejected.  xis_out: m x (lg+1) x 4, U_out: m x 12 (both nullable).
int halo_pcdl_succinct_check_batch(halo_ctx *ctx, size_t d, const uint64_t *instances, size_t m, uint64_t *xis_out, uint64_t *U_out,
                                   int *status) {
    HALO_CTX2(ctx);
    if (m && !instances) { set_error("succinct_check_batch: null pointer"); return HALO_E_ARG; }
    if (!is_pow2(d + 1)) return fail_reject("d+1 is not a power of 2!");
    size_t lg = ilog2(d + 1), iw = instance_words(lg);
    for (size_t i = 0; i < m; ++i)
        if ((size_t)(instances + i * iw)[12] != d || (instances + i * iw)[22] != lg) return fail_reject("d_i != d");
    std::vector<BatchCheck> res;
    if (m >= kBatchVerifyMin && ctx->batch_verify) {
        int rc = succinct_check_batch(ctx, d, instances, m, res);
        if (rc) return rc;
    } else {
        res.assign(m, BatchCheck());
        pool_run(m, [&](size_t i) {
            const uint64_t *q = instances + i * iw;
            res[i].rc = succinct_challenges(ctx, Point::load(q), d, Fr::load(q + 13), Fr::load(q + 17), q + 21, &res[i].st);
            if (!res[i].rc) res[i].rc = succinct_relation(res[i].st, Fr::load(q + 13), Fr::load(q + 17), q + 21);
            if (res[i].rc) res[i].err = halo_last_error();
        });
    }
    int first = -1;
    for (size_t i = 0; i < m; ++i) {
        if (status) status[i] = res[i].rc;
        if (res[i].rc) { if (first < 0) first = (int)i; continue; }
        if (xis_out) for (size_t k = 0; k <= lg; ++k) res[i].st.xis[k].store(xis_out + ((lg + 1) * i + k) * 4);
        if (U_out) res[i].st.U.store_normalized(U_out + 12 * i);
    }
    if (first >= 0) { set_error("instance " + std::to_string(first) + ": " + res[first].err); return res[first].rc; }
    return HALO_OK;
}

int halo_pcdl_check(halo_ctx *ctx, const uint64_t C[12], size_t d, const uint64_t z[4], const uint64_t v[4], const uint64_t *proof) {
    HALO_CTX2(ctx);
    return pcdl_check_host(ctx, Point::load(C), d, Fr::load(z), Fr::load(v), proof);
}

int halo_pcdl_check_partial(halo_ctx *ctx, const uint64_t C[12], size_t d, const uint64_t z[4], const uint64_t v[4], const uint64_t *proof,
                            uint64_t stride, uint64_t offset, uint64_t U_out[12], uint64_t part_out[12]) {
    HALO_CTX2(ctx);
    if (!C || !z || !v || !proof || !U_out || !part_out) { set_error("check_partial: null pointer"); return HALO_E_ARG; }
    Point U, part;
    int rc = pcdl_check_partial_host(ctx, Point::load(C), d, Fr::load(z), Fr::load(v), proof, stride, offset, &U, &part);
    if (rc) return rc;
    U.store_normalized(U_out);
    part.store_normalized(part_out);
    return HALO_OK;
}

// ------------------------------------------------------------------ sharded open / check in one call each
// The round loop of sharded.ShardedOpen (Python) in the library: G, c and the z-powers are placed cyclically (element i on
// rank i mod P; the context is that rank's halo_ctx_create_urs_strided shard) and every collective of the open is one call
// of the caller's all-gather: P x words in rank order.  What a rank exchanges: its share of p(z) (4 words; with the hiding
// branch 16: the share of C_bar as well), per round L | R | dot_l | dot_r (32 words), at the end its last element of G, c
// and z (20 words) -- and ONE STATUS WORD behind every record.  Every rank computes the same challenges and returns the
// same proof: the bytes of halo_pcdl_open.
//
// Failure safety.  The number and order of collectives of a call depend only on the arguments every rank passes alike
// (P, d, hiding or not).  Whatever fails on ONE rank between two collectives -- a device allocation, a launch, a copy, a
// per-rank argument such as too many local coefficients -- does not make that rank return: it enters the next collective
// with its error code in the status word (its record zeroed), every rank reads the P status words of that collective and
// all of them return the first non-zero one in rank order, at the same collective.  No rank is left waiting in an
// all-gather its peers never enter.  (A collective that itself fails -- the callback returns non-zero -- is the caller's
// fabric failing: the call returns HALO_E_ARG on the ranks that see it and the caller must abort its process group.)
namespace {
// The development library's hooks "shard_fail_rank" / "shard_fail_at" (tuning.hpp DevHooks; never the environment): the rank with
// that offset fails locally (HALO_E_DEVICE) before collective number `at` of a sharded open (0 = the share of p(z), 1.. = the
// rounds, then the tail), or, with at = -2, in a sharded check -- what tests/test_sharded_gloo.py and
// tests/test_gpu_pcdl_acc.py use to drive the failure path above
int shard_test_failure(uint64_t offset, long step, bool in_check) {
    const DevHooks &h = dev_hooks();
    if (h.shard_fail_rank < 0 || (uint64_t)h.shard_fail_rank != offset) return HALO_OK;
    const bool hit = in_check ? h.shard_fail_at == -2 : (h.shard_fail_at >= 0 && (long)h.shard_fail_at == step);
    if (!hit) return HALO_OK;
    set_error("sharded call: local failure injected by the development library's shard_fail hook");
    return HALO_E_DEVICE;
}
struct StatusGather {
    size_t P;
    uint64_t offset;
    halo_allgather_fn fn;
    void *user;
    const char *who;
    std::vector<uint64_t> sbuf, rbuf;
    // all-gather of `words` record words + this rank's status; recv = P x words.  Returns the first non-zero status in rank
    // order (the same value on every rank), HALO_E_ARG if the collective itself failed, else 0.
    int run(const uint64_t *rec, size_t words, int local_rc, std::vector<uint64_t> &recv) {
        recv.assign(P * words, 0);
        if (!fn) {  // one rank, no callback: nothing to agree on
            if (local_rc) return local_rc;
            std::memcpy(recv.data(), rec, words * 8);
            return HALO_OK;
        }
        sbuf.assign(words + 1, 0);
        if (!local_rc) std::memcpy(sbuf.data(), rec, words * 8);
        sbuf[words] = (uint64_t)(int64_t)local_rc;
        rbuf.assign(P * (words + 1), 0);
        std::string own = local_rc ? halo_last_error() : "";
        if (fn(user, sbuf.data(), words + 1, rbuf.data())) {
            set_error(std::string(who) + ": the caller's all-gather failed (abort the process group: the ranks are no longer in step)");
            return HALO_E_ARG;
        }
        for (size_t r = 0; r < P; ++r) {
            int st = (int)(int64_t)rbuf[r * (words + 1) + words];
            if (!st) continue;
            if (r == offset || local_rc == st) set_error(own);  // (a rejection every rank found by itself keeps its own wording)
            else set_error(std::string(who) + ": rank " + std::to_string(r) + " failed locally (code " + std::to_string(st) + "); every rank returns its code");
            return st;
        }
        for (size_t r = 0; r < P; ++r) std::memcpy(&recv[r * words], &rbuf[r * (words + 1)], words * 8);
        return HALO_OK;
    }
};
}  // namespace

int halo_pcdl_open_sharded(halo_ctx *ctx, uint64_t stride, uint64_t offset, uint64_t *rng_state, const uint64_t *coeffs_local, size_t len_local,
                           size_t deg, const uint64_t C_w[12], size_t d, const uint64_t z_w[4], const uint64_t *w_w, halo_allgather_fn allgather,
                           void *user, uint64_t *proof, uint64_t v_out[4]) {
    HALO_CTX2(ctx);
    // (arguments every rank passes alike: a mistake here is the same mistake everywhere, returned before any collective)
    if (!C_w || !z_w || !proof || !v_out || (w_w && !rng_state)) { set_error("open_sharded: null pointer"); return HALO_E_ARG; }
    const size_t P = (size_t)stride;
    if (P == 0 || !is_pow2(P) || offset >= stride || (P > 1 && !allgather)) { set_error("open_sharded: stride must be a power of two, offset below it, and an all-gather given"); return HALO_E_ARG; }
    if (P > 64) { set_error("open_sharded: at most 64 ranks"); return HALO_E_ARG; }
    size_t n = d + 1;
    if (!is_pow2(n)) return fail_assert("open: d+1 is not a power of 2");  // pcdl.rs:130-132
    if (n < P) return fail_assert("open: d > D");
    const size_t nl = n / P, lg_n = ilog2(n), lg_l = ilog2(nl);
    std::memset(proof, 0, 8 * proof_words(lg_n));
    proof[1] = lg_n;
    StatusGather sg{P, offset, allgather, user, "open_sharded", {}, {}};
    long step = 0;
    // lrc: this rank's own failure since the last collective; it rides into the next one (see above)
    int lrc = HALO_OK, rc;
    if (len_local && !coeffs_local) { set_error("open_sharded: null pointer"); lrc = HALO_E_ARG; }
    else if (nl > ctx->n) lrc = fail_assert("open: d > D");                    // (per-rank: the shard contexts may differ)
    else if (len_local > nl) lrc = fail_assert("open: p.degree() > d");
    halo_ipa *st = nullptr;
    if (!lrc) lrc = halo_ipa_begin_strided(ctx, nl, coeffs_local, len_local, z_w, stride, offset, &st);
    std::unique_ptr<halo_ipa, void (*)(halo_ipa *)> guard(st, halo_ipa_destroy);
    std::vector<uint64_t> recv;
    uint64_t send[32] = {};
    if (!lrc) lrc = halo_ipa_dot_cz(st, send);  // this shard's share of p(z)   (:135)
    uint64_t Cm[12];
    std::memcpy(Cm, C_w, sizeof Cm);
    if (!lrc) lrc = shard_test_failure(offset, step, false);
    if (w_w) {  // :137-164
        if (!lrc) lrc = halo_ipa_hiding_partial(st, *rng_state, deg, z_w, stride, offset, send + 4);
        rc = sg.run(send, 16, lrc, recv);
        if (rc) return rc;
        std::vector<uint64_t> v_parts(4 * P), cb_parts(12 * P);
        for (size_t r = 0; r < P; ++r) { std::memcpy(&v_parts[4 * r], &recv[16 * r], 32); std::memcpy(&cb_parts[12 * r], &recv[16 * r + 4], 96); }
        uint64_t alpha[4], Cprime[12];
        // (host arithmetic on gathered data: the same outcome on every rank)
        rc = halo_open_hiding_combine(C_w, z_w, v_parts.data(), cb_parts.data(), P, w_w, rng_state, deg, pf_Cbar(proof, lg_n), alpha, pf_wp(proof, lg_n), Cprime);
        if (rc) return rc;
        lrc = halo_ipa_apply_hiding(st, alpha);  // p' = p + alpha p_bar   (:156)
        std::memcpy(Cm, Cprime, sizeof Cm);
        proof[0] = 1;
        recv.swap(v_parts);
    } else {
        Point::infinity().store(pf_Cbar(proof, lg_n));
        rc = sg.run(send, 4, lrc, recv);
        if (rc) return rc;
    }
    uint64_t xi[4], Hp[12];
    rc = halo_open_start(Cm, z_w, recv.data(), P, v_out, xi, Hp);  // v, xi_0, H'   (:135, :180-181)
    if (rc) return rc;
    for (size_t round = 0; round < lg_l; ++round) {
        ++step;
        if (!lrc) lrc = shard_test_failure(offset, step, false);
        if (!lrc) lrc = halo_ipa_round_lr_partial(st, send, send + 12, send + 24);
        rc = sg.run(send, 32, lrc, recv);
        if (rc) return rc;
        uint64_t xn[4], xinv[4];
        rc = halo_open_combine(recv.data(), P, Hp, xi, pf_L(proof, round), pf_R(proof, lg_n, round), xn, xinv);  // :203-213
        if (rc) return rc;
        std::memcpy(xi, xn, sizeof xi);
        lrc = halo_ipa_round_fold(st, xn, xinv);  // :216-224
    }
    uint64_t last[20] = {};
    ++step;
    if (!lrc) lrc = shard_test_failure(offset, step, false);
    if (!lrc) lrc = halo_ipa_finish_z(st, last, last + 12, last + 16);
    if (P == 1) {
        if (lrc) return lrc;
        std::memcpy(pf_U(proof, lg_n), last, 96);
        std::memcpy(pf_c(proof, lg_n), last + 12, 32);
        return HALO_OK;
    }
    rc = sg.run(last, 20, lrc, recv);  // the P remaining elements, in index order
    if (rc) return rc;
    size_t lgP = ilog2(P);
    std::vector<uint64_t> Ls(12 * lgP), Rs(12 * lgP);
    rc = halo_open_tail(recv.data(), P, Hp, xi, Ls.data(), Rs.data(), pf_U(proof, lg_n), pf_c(proof, lg_n));  // the last lg P rounds
    if (rc) return rc;
    for (size_t k = 0; k < lgP; ++k) {
        std::memcpy(pf_L(proof, lg_l + k), &Ls[12 * k], 96);
        std::memcpy(pf_R(proof, lg_n, lg_l + k), &Rs[12 * k], 96);
    }
    return HALO_OK;
}

// pcdl::check over the same shards: halo_pcdl_check_partial, one all-gather of 12 words + the status word, the shares added
// in rank order, U compared (pcdl.rs:338-339).  HALO_E_REJECT on every rank alike; a rank whose own half failed (device error)
// still enters the collective and every rank returns its code (see halo_pcdl_open_sharded).
int halo_pcdl_check_sharded(halo_ctx *ctx, uint64_t stride, uint64_t offset, const uint64_t C[12], size_t d, const uint64_t z[4], const uint64_t v[4],
                            const uint64_t *proof, halo_allgather_fn allgather, void *user) {
    HALO_CTX2(ctx);
    if (!C || !z || !v || !proof || (stride > 1 && !allgather)) { set_error("check_sharded: null pointer"); return HALO_E_ARG; }
    if (stride == 0 || stride > 64 || offset >= stride) { set_error("check_sharded: stride in 1..64, offset below it"); return HALO_E_ARG; }
    Point U, part;
    // The succinct check is host arithmetic on the same proof on every rank: its HALO_E_REJECT is the same everywhere, but it
    // goes through the status word like any other outcome, so that the collective count stays fixed (one).
    int lrc = shard_test_failure(offset, 0, true);
    if (!lrc) lrc = pcdl_check_partial_host(ctx, Point::load(C), d, Fr::load(z), Fr::load(v), proof, stride, offset, &U, &part);
    uint64_t send[12] = {};
    if (!lrc) part.store_normalized(send);
    StatusGather sg{(size_t)stride, offset, allgather, user, "check_sharded", {}, {}};
    std::vector<uint64_t> recv;
    int rc = sg.run(send, 12, lrc, recv);
    if (rc) return rc;
    Point comm = Point::infinity();
    for (uint64_t r = 0; r < stride; ++r) comm = comm + Point::load(&recv[12 * r]);
    if (U != comm) return fail_reject("U != CM.Commit(ck, h_vec)");  // :339
    return HALO_OK;
}

// acc.rs:190-220
int halo_acc_prover(halo_ctx *ctx, uint64_t *rng_state, size_t d, const uint64_t *qs, size_t m, uint64_t *acc) {
    HALO_CTX2(ctx);
    if (!is_pow2(d + 1)) return fail_assert("prover: d + 1 is not a power of two");
    if (d + 1 > ctx->n) return fail_assert("prover: d > D");
    size_t lg = ilog2(d + 1), n = d + 1;
    host::Rng rng{rng_state ? *rng_state : 0};
    Fr h0[2] = {rng.scalar(), rng.scalar()};  // :192
    uint64_t h0w[8];
    h0[0].store(h0w);
    h0[1].store(h0w + 4);
    Point U0;
    int rc = pcdl_commit_host(ctx, h0w, 2, d, nullptr, &U0);  // :195
    if (rc) return rc;
    Fr w = rng.scalar();  // :198
    Point C_bar;
    Fr z;
    AccHPolys hs;
    rc = common_subroutine(ctx, d, qs, m, h0, U0, w, &C_bar, &z, &hs);  // :202
    if (rc) return rc;
    Fr v = hs.eval(z);  // :205
    // h.get_poly(): h_0 + sum alpha^(i+1) h_i(X), expanded on the device (acc.rs:85-94)
    rc = ensure_poly_buffers(ctx);
    if (rc) return rc;
    HALO_HIP(hipMemsetAsync(ctx->d_poly, 0, n * 32, ctx->stream));
    rc = upload_words(ctx, ctx->d_poly, h0w, (n < 2 ? n : 2) * 4);
    if (rc) return rc;
    for (size_t i = 0; i < m; ++i) {
        rc = h_coeffs_dev(ctx, hs.xis[i].data(), lg, hs.alphas[i + 1], true, ctx->d_poly);
        if (rc) return rc;
    }
    std::memset(acc, 0, 8 * acc_words(lg));
    C_bar = C_bar.normalized();
    C_bar.store(acc);
    acc[12] = d;
    z.store(acc + 13);
    v.store(acc + 17);
    size_t deg = m ? d : (h0[1].is_zero() ? 0 : 1);
    rc = pcdl_open_dev(ctx, &rng, deg, C_bar, d, z, &w, acc + 21);  // :209
    uint64_t *piV = acc + instance_words(lg);
    std::memcpy(piV, h0w, 64);
    U0.store_normalized(piV + 8);
    w.store(piV + 20);
    if (rng_state) *rng_state = rng.state;
    return rc;
}

// acc.rs:223-243
int halo_acc_verifier(halo_ctx *ctx, size_t d, const uint64_t *qs, size_t m, const uint64_t *acc) {
    HALO_CTX2(ctx);
    if (!is_pow2(d + 1)) return fail_reject("d+1 is not a power of 2!");
    size_t lg = ilog2(d + 1);
    const uint64_t *piV = acc + instance_words(lg);
    Fr h0[2] = {Fr::load(piV), Fr::load(piV + 4)};
    if (!Point::load(piV + 8).on_curve() || !Point::load(acc).on_curve() || !scalar_ok(h0[0]) || !scalar_ok(h0[1]) ||
        !scalar_ok(Fr::load(piV + 20)) || !scalar_ok(Fr::load(acc + 13)) || !scalar_ok(Fr::load(acc + 17)))
        return fail_reject("accumulator holds an invalid point or scalar");
    Point C_bar_p;
    Fr z_p;
    AccHPolys hs;
    int rc = common_subroutine(ctx, d, qs, m, h0, Point::load(piV + 8), Fr::load(piV + 20), &C_bar_p, &z_p, &hs);
    if (rc) return rc;
    Fr z = Fr::load(acc + 13), v = Fr::load(acc + 17);
    if (C_bar_p != Point::load(acc)) return fail_reject("C_bar' != C_bar");
    if (z_p != z) return fail_reject("z' != z");
    if ((size_t)acc[12] != d) return fail_reject("d' != d");
    if (hs.eval(z) != v) return fail_reject("h(z) != v");
    return HALO_OK;
}

// acc.rs:245-255
int halo_acc_decider(halo_ctx *ctx, const uint64_t *acc) {
    HALO_CTX2(ctx);
    return pcdl_check_host(ctx, Point::load(acc), (size_t)acc[12], Fr::load(acc + 13), Fr::load(acc + 17), acc + 21);
}

// benches/acc.rs:15-29 random_instance (workload generator for BASELINE config 4)
int halo_random_instance(halo_ctx *ctx, uint64_t *rng_state, size_t d, uint64_t *inst) {
    HALO_CTX2(ctx);
    if (!is_pow2(d + 1) || d < 2) return fail_assert("random_instance: bad d");
    if (d + 1 > ctx->n) return fail_assert("random_instance: d > D");
    size_t lg = ilog2(d + 1), n = d + 1;
    host::Rng rng{rng_state ? *rng_state : 0};
    size_t lo = d / 2, d_prime = lo + (size_t)(rng.next() % (uint64_t)(d - lo));
    if (d_prime == 0) d_prime = 1;
    Fr w = rng.scalar();
    int rc = ensure_poly_buffers(ctx);
    if (rc) return rc;
    // p = PallasPoly::rand(d_prime): d_prime + 1 scalars of the stream, generated on the device
    HALO_HIP(hipMemsetAsync(ctx->d_poly, 0, n * 32, ctx->stream));
    rc = rng_scalars_dev(ctx, rng.state, d_prime + 1, ctx->d_poly);
    if (rc) return rc;
    rng.state += 4 * (uint64_t)(d_prime + 1) * 0x9E3779B97F4A7C15ULL;
    Point C;
    rc = pedersen_commit_dev(ctx, &w, ctx->d_poly, n, &C);
    if (rc) return rc;
    C = C.normalized();
    Fr z = rng.scalar(), v;
    rc = fr_poly_eval(ctx, ctx->d_poly, d_prime + 1, z, &v);
    if (rc) return rc;
    std::memset(inst, 0, 8 * instance_words(lg));
    C.store(inst);
    inst[12] = d;
    z.store(inst + 13);
    v.store(inst + 17);
    // the leading coefficient is non-zero with overwhelming probability: degree = d_prime
    rc = pcdl_open_dev(ctx, &rng, d_prime, C, d, z, &w, inst + 21);
    if (rng_state) *rng_state = rng.state;
    return rc;
}

}  // extern "C"
