// K3-K9: the kernels behind pcdl::open's halving loop (pcdl.rs:195-227), construct_powers
// (group.rs:29-37), scalar_dot (group.rs:13-15), p(z) (pcdl.rs:135) and h(X) (pcdl.rs:56-91).
//
// Fr work is one 32-byte element per lane, loaded as two dwordx4 from contiguous arrays
// (coalesced 2 KiB per wave-instruction pair); these kernels are the HBM-bound part of the path.
// The point fold uses ONE scalar for the whole launch, so every lane runs the same
// double-and-add schedule with no divergence.
#include "curve_quad.hpp"
#include "fr29.hpp"
#include "internal.hpp"

namespace halo {

// An Fr multiplier in N-form (fr29.hpp: k 2^261 mod r = the Montgomery limbs of 32 k), 9 x 29-bit limbs by value
struct FsArg { uint32_t v[9]; };
static FsArg to_nform(const host::Fr &k) {
    host::Fr t = k * host::Fr::from_u64(32);
    FsArg a;
    for (int i = 0; i < 9; ++i) {
        int bit = 29 * i, w = bit >> 6, sh = bit & 63;
        uint64_t lo = t.l[w] >> sh;
        if (sh > 35 && w + 1 < 4) lo |= t.l[w + 1] << (64 - sh);
        a.v[i] = (uint32_t)(i < 8 ? (lo & ((1u << 29) - 1)) : lo);
    }
    return a;
}
HALO_DEV Fs<1> from_nform(const FsArg &a) {
    Fs<1> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = a.v[i];
    return r;
}

struct FeArg { uint32_t v[8]; };  // by-value kernel argument (SGPRs)
static FeArg to_arg(const host::Fr &f) {
    FeArg a;
    for (int i = 0; i < 4; ++i) { a.v[2 * i] = (uint32_t)f.l[i]; a.v[2 * i + 1] = (uint32_t)(f.l[i] >> 32); }
    return a;
}
HALO_DEV Fe from_arg(const FeArg &a) {
    Fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = a.v[i];
    return r;
}

// ------------------------------------------------------------------ K3: G'[j] = G[j] + xi * G[j+m]
// xi is one scalar for the whole launch, expanded on the host as xi = sum_i d_i 2^i with digits from the six
// Eisenstein units {+-1, +-lambda, +-lambda^2} (host_math.hpp glv_digits): a ~127-step double-and-add with
// ~71 additions, each of a "free" point d_i * P = (beta^e x, +-y) -- two multiplications per input point.
// Every branch below depends only on kernel arguments, so the 64 lanes of a wave never diverge.
struct GlvArg {
    uint32_t dig[14];  // ten 3-bit digit codes per word, least significant digit first (host_math.hpp glv_digits)
    int ndigits;
};
HALO_DEV Fq<2> fq_const(const uint32_t (&c)[9]) {
    Fq<2> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = c[i];
    return r;
}
HALO_DEV Fq<2> pick3(int e, const Fq<2> &a, const Fq<2> &b, const Fq<2> &c) {
    Fq<2> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = e == 0 ? a.v[i] : (e == 1 ? b.v[i] : c.v[i]);
    return r;
}
// G[j] + xi * G[j + m] as a Jacobian point
HALO_DEV JacN fold_one(const uint32_t *__restrict__ G, uint32_t j, uint32_t m, const GlvArg &a) {
    AffN hi = aff_load(G + AFF_STRIDE * (size_t)(j + m));
    AffN lo = aff_load(G + AFF_STRIDE * (size_t)j);
    if (aff_is_inf(hi)) return jac_from_aff(lo);  // G[j] + xi * infinity = G[j]
    constexpr uint32_t BETA[9] = {0x1342a796, 0x3fdac51, 0x54dab11, 0x5b221a6, 0xccd27ac, 0x15cc87a4, 0x1b1533b6, 0x169e85e1, 0x3b0093};
    constexpr uint32_t BETA2[9] = {0xcbd58eb, 0x1a2f8f16, 0xd140efa, 0x7bdfb9, 0x1333ecad, 0xa33785b, 0x4eacc49, 0x9617a1e, 0x4ff6c};
    Fq<2> x0 = hi.x, x1 = fq_mul(hi.x, fq_const(BETA)), x2 = fq_mul(hi.x, fq_const(BETA2));
    Fq<2> yp = hi.y, yn = fq_neg<2>(hi.y);
    JacN acc = jac_inf();
    int top = a.ndigits - 1;
#pragma unroll 1
    for (int word = top / 10; word >= 0; word--) {
        uint32_t w = 0;
#pragma unroll
        for (int q = 0; q < 14; q++) w = (q == word) ? a.dig[q] : w;
#pragma unroll 1
        for (int k = (word == top / 10) ? (top % 10) : 9; k >= 0; k--) {
            acc = jac_dbl(acc);
            uint32_t code = (w >> (3 * k)) & 7u;
            if (code) {  // wave-uniform: +-w^e * hi = (beta^e x, +-y)
                AffN t;
                t.x = pick3((int)((code - 1) % 3), x0, x1, x2);
                t.y = code > 3 ? yn : yp;
                acc = jac_madd(acc, t);
            }
        }
    }
    return jac_madd(acc, lo);
}
// Each lane folds two points (j and j + half) and brings both back to affine with ONE Fermat inversion
// (of Z_a * Z_b): the inversion is ~15 % of a single fold.
__global__ __launch_bounds__(256) void k_fold_points(uint32_t *__restrict__ G, uint32_t m, uint32_t half, GlvArg a) {
    uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= half) return;
    bool two = j + half < m;
    JacN ra = fold_one(G, j, m, a);
    JacN rb = jac_inf();
    if (two) rb = fold_one(G, j + half, m, a);
    bool ia = jac_is_inf(ra), ib = jac_is_inf(rb);
    Fq<4> za = ia ? fq_widen<4>(fq_one()) : ra.z, zb = ib ? fq_widen<4>(fq_one()) : rb.z;
    Fq<2> zi = fq_inv(fq_mul(za, zb));
    Fq<2> zia = fq_mul(zi, zb), zib = fq_mul(zi, za);
    AffN oa = aff_inf(), ob = aff_inf();
    if (!ia) {
        Fq<2> z2 = fq_sqr(zia);
        oa.x = fq_mul(ra.x, z2);
        oa.y = fq_mul(ra.y, fq_mul(z2, zia));
    }
    if (!ib) {
        Fq<2> z2 = fq_sqr(zib);
        ob.x = fq_mul(rb.x, z2);
        ob.y = fq_mul(rb.y, fq_mul(z2, zib));
    }
    aff_store(G + AFF_STRIDE * (size_t)j, oa);
    if (two) aff_store(G + AFF_STRIDE * (size_t)(j + half), ob);
}

// ------------------------------------------------------------------ K3': two halving rounds of G in one pass
// After two rounds without touching G the folded key is G''[j] = G[j] + s1 G[j+m] + s2 G[j+2m] + s3 G[j+3m] with
// (s1, s2, s3) = (xi_2, xi_1, xi_1 xi_2), m = a quarter of the key (pcdl.rs:218 applied twice).  The three scalar
// multiplications share ONE doubling chain (Straus): ~128 doublings + 3 x ~71 additions per output instead of
// 2 x (128 + 71) for each of the 1.5 outputs the two separate folds produce -- ~38 % fewer field products for the
// same two rounds.  The rounds in between take L and R from MSMs over the unfolded key (the "no-fold" form below).
struct GlvArg3 {
    uint32_t dig[3][14];  // as GlvArg, one digit string per scalar
    int ndigits;          // longest of the three
};
HALO_DEV JacN fold_one4(const uint32_t *G, uint32_t j, uint32_t m, const GlvArg3 &a) {
    constexpr uint32_t BETA[9] = {0x1342a796, 0x3fdac51, 0x54dab11, 0x5b221a6, 0xccd27ac, 0x15cc87a4, 0x1b1533b6, 0x169e85e1, 0x3b0093};
    constexpr uint32_t BETA2[9] = {0xcbd58eb, 0x1a2f8f16, 0xd140efa, 0x7bdfb9, 0x1333ecad, 0xa33785b, 0x4eacc49, 0x9617a1e, 0x4ff6c};
    AffN p1 = aff_load(G + AFF_STRIDE * (size_t)(j + m)), p2 = aff_load(G + AFF_STRIDE * (size_t)(j + 2 * m)),
         p3 = aff_load(G + AFF_STRIDE * (size_t)(j + 3 * m));
    bool live1 = !aff_is_inf(p1), live2 = !aff_is_inf(p2), live3 = !aff_is_inf(p3);
    // acc += unit(code) * p: code is wave-uniform; lambda^e * (x, y) = (beta^e x, y)
    auto step = [&](JacN &acc, const AffN &p, bool live, uint32_t code) {
        if (!code) return;
        int e = (int)((code - 1) % 3);
        AffN q;
        q.x = p.x;
        if (e) q.x = fq_mul(p.x, fq_const(e == 1 ? BETA : BETA2));
        q.y = code > 3 ? fq_neg<2>(p.y) : p.y;
        if (live) acc = jac_madd(acc, q);
    };
    JacN acc = jac_inf();
    int top = a.ndigits - 1;
#pragma unroll 1
    for (int word = top / 10; word >= 0; word--) {
        uint32_t w1 = 0, w2 = 0, w3 = 0;
#pragma unroll
        for (int q = 0; q < 14; q++) {
            w1 = (q == word) ? a.dig[0][q] : w1;
            w2 = (q == word) ? a.dig[1][q] : w2;
            w3 = (q == word) ? a.dig[2][q] : w3;
        }
#pragma unroll 1
        for (int k = (word == top / 10) ? (top % 10) : 9; k >= 0; k--) {
            acc = jac_dbl(acc);
            step(acc, p1, live1, (w1 >> (3 * k)) & 7u);
            step(acc, p2, live2, (w2 >> (3 * k)) & 7u);
            step(acc, p3, live3, (w3 >> (3 * k)) & 7u);
        }
    }
    return jac_madd(acc, aff_load(G + AFF_STRIDE * (size_t)j));
}
// G (read) and out (written) may be the same array: a lane reads the indices j + t m and writes index j only
__global__ __launch_bounds__(256, 2) void k_fold_points4(const uint32_t *G, uint32_t *out, uint32_t m, uint32_t half, GlvArg3 a) {
    // the first result waits in LDS while the second ladder runs (three bases + the accumulator + a mixed addition's
    // temporaries fill the 256 registers that two waves per SIMD allow); word k of thread t at park[256 k + t]
    __shared__ uint32_t park[27 * 256];
    uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= half) return;
    bool two = j + half < m;
    JacN rb = jac_inf();
    {
        JacN ra0 = fold_one4(G, j, m, a);
        uint32_t *mine = park + threadIdx.x;
#pragma unroll
        for (int k = 0; k < 9; k++) { mine[256 * k] = ra0.x.v[k]; mine[256 * (9 + k)] = ra0.y.v[k]; mine[256 * (18 + k)] = ra0.z.v[k]; }
    }
    if (two) rb = fold_one4(G, j + half, m, a);
    JacN ra;
    {
        const uint32_t *mine = park + threadIdx.x;
#pragma unroll
        for (int k = 0; k < 9; k++) { ra.x.v[k] = mine[256 * k]; ra.y.v[k] = mine[256 * (9 + k)]; ra.z.v[k] = mine[256 * (18 + k)]; }
    }
    bool ia = jac_is_inf(ra), ib = jac_is_inf(rb);
    Fq<4> za = ia ? fq_widen<4>(fq_one()) : ra.z, zb = ib ? fq_widen<4>(fq_one()) : rb.z;
    Fq<2> zi = fq_inv(fq_mul(za, zb));
    Fq<2> zia = fq_mul(zi, zb), zib = fq_mul(zi, za);
    AffN oa = aff_inf(), ob = aff_inf();
    if (!ia) {
        Fq<2> z2 = fq_sqr(zia);
        oa.x = fq_mul(ra.x, z2);
        oa.y = fq_mul(ra.y, fq_mul(z2, zia));
    }
    if (!ib) {
        Fq<2> z2 = fq_sqr(zib);
        ob.x = fq_mul(rb.x, z2);
        ob.y = fq_mul(rb.y, fq_mul(z2, zib));
    }
    aff_store(out + AFF_STRIDE * (size_t)j, oa);
    if (two) aff_store(out + AFF_STRIDE * (size_t)(j + half), ob);
}

// The same two-level fold for SMALL keys (fewer than 2^16 outputs): there the pass is one latency chain (~128 doublings,
// ~210 additions, one inversion: 1.5 ms whatever m is), so the 4 lanes of a quad share every group operation
// (curve_quad.hpp: 3 product levels per doubling, 5 per mixed addition instead of 7 and 11 products) -- one output per quad.
__global__ __launch_bounds__(256, 2) void k_fold_points4_quad(const uint32_t *G, uint32_t *out, uint32_t m, GlvArg3 a) {
    constexpr uint32_t BETA[9] = {0x1342a796, 0x3fdac51, 0x54dab11, 0x5b221a6, 0xccd27ac, 0x15cc87a4, 0x1b1533b6, 0x169e85e1, 0x3b0093};
    constexpr uint32_t BETA2[9] = {0xcbd58eb, 0x1a2f8f16, 0xd140efa, 0x7bdfb9, 0x1333ecad, 0xa33785b, 0x4eacc49, 0x9617a1e, 0x4ff6c};
    uint32_t t = blockIdx.x * 256 + threadIdx.x;
    uint32_t j = t >> 2;
    int ql = (int)(t & 3);
    bool in = j < m;
    if (!in) j = m - 1;  // whole quads and waves stay busy (DPP); the surplus quads recompute the last output and do not store
    AffN p1 = aff_load(G + AFF_STRIDE * (size_t)(j + m)), p2 = aff_load(G + AFF_STRIDE * (size_t)(j + 2 * m)),
         p3 = aff_load(G + AFF_STRIDE * (size_t)(j + 3 * m));
    bool live1 = !aff_is_inf(p1), live2 = !aff_is_inf(p2), live3 = !aff_is_inf(p3);
    Fq<2> one = fq_widen<2>(fq_one()), beta = fq_const(BETA), beta2 = fq_const(BETA2);
    auto step = [&](JacN &acc, const AffN &p, bool live, uint32_t code) {
        if (!code) return;  // wave-uniform
        int e = (int)((code - 1) % 3);
        Fq<2> bc = e == 0 ? one : (e == 1 ? beta : beta2);
        Fq<2> y = code > 3 ? fq_neg<2>(p.y) : p.y;
        acc = jac_madd_quad(acc, p.x, y, bc, live, ql);
    };
    JacN acc = jac_inf();
    int top = a.ndigits - 1;
#pragma unroll 1
    for (int word = top / 10; word >= 0; word--) {
        uint32_t w1 = 0, w2 = 0, w3 = 0;
#pragma unroll
        for (int q = 0; q < 14; q++) {
            w1 = (q == word) ? a.dig[0][q] : w1;
            w2 = (q == word) ? a.dig[1][q] : w2;
            w3 = (q == word) ? a.dig[2][q] : w3;
        }
#pragma unroll 1
        for (int k = (word == top / 10) ? (top % 10) : 9; k >= 0; k--) {
            acc = jac_dbl_quad(acc, ql);
            step(acc, p1, live1, (w1 >> (3 * k)) & 7u);
            step(acc, p2, live2, (w2 >> (3 * k)) & 7u);
            step(acc, p3, live3, (w3 >> (3 * k)) & 7u);
        }
    }
    AffN lo = aff_load(G + AFF_STRIDE * (size_t)j);
    acc = jac_madd_quad(acc, lo.x, lo.y, one, !aff_is_inf(lo), ql);
    if (in && ql == 0) aff_store(out + AFF_STRIDE * (size_t)j, jac_to_aff(acc));  // the inversion: one lane of the quad
}

// ------------------------------------------------------------------ K4: c' = c_l + xi^-1 c_r ; z' = z_l + xi z_r
// 192 bytes moved and two products per element: HBM-bound with the 29-bit product (fr29.hpp)
__global__ __launch_bounds__(256) void k_fold_scalars(uint64_t *__restrict__ c, uint64_t *__restrict__ z, uint32_t m, FsArg xi,
                                                      FsArg xi_inv) {
    uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    Fe cl = fe_load(c + 4 * (size_t)j), cr = fe_load(c + 4 * (size_t)(j + m));
    Fe zl = fe_load(z + 4 * (size_t)j), zr = fe_load(z + 4 * (size_t)(j + m));
    fs_store(c + 4 * (size_t)j, fs_add(fs_from_fe(cl), fs_mul(fs_from_fe(cr), from_nform(xi_inv))));
    fs_store(z + 4 * (size_t)j, fs_add(fs_from_fe(zl), fs_mul(fs_from_fe(zr), from_nform(xi))));
}

// ------------------------------------------------------------------ block-wide Fr sum
HALO_DEV Fe block_sum_fr(Fe v, Fe *lds /* >= 4 entries */) {
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll 1
    for (int off = 32; off >= 1; off >>= 1) {
        Fe o = fe_shfl(v, (lane + off) & 63);
        if (lane < off) v = fe_add<FrCfg>(v, o);
    }
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); w++) v = fe_add<FrCfg>(v, lds[w]);
    }
    return v;  // valid in thread 0
}

// ------------------------------------------------------------------ K5: two dot products per launch
// A lane takes two elements of each pair of vectors per trip -- x_i y_i + x_j y_j share one reduction -- and the next
// trip's operands are requested before the current products run.  The products of two A-form values carry 1/32
// (fr29.hpp): the partial sums stay in that skewed form and k_sum_partials multiplies the total by 32 once.
// The grid is kept small (two waves per SIMD): with one trip per lane the block-wide sum cost as many instructions as the
// products themselves.
struct DotOps { Fe x, y, x2, y2; };
HALO_DEV DotOps dot_load(const uint64_t *__restrict__ xs, const uint64_t *__restrict__ ys, uint32_t i, uint32_t i2, uint32_t m) {
    DotOps o;
    uint32_t j = i2 < m ? i2 : i;  // (past the end: the slot re-reads element i and is zeroed)
    o.x = fe_load(xs + 4 * (size_t)i); o.y = fe_load(ys + 4 * (size_t)i);
    o.x2 = fe_load(xs + 4 * (size_t)j); o.y2 = fe_load(ys + 4 * (size_t)j);
    if (i2 >= m) o.x2 = fe_zero();
    return o;
}
HALO_DEV Fs<2> dot_step(const Fs<2> &acc, const DotOps &o) {
    return fs_tighten(fs_add(acc, fs_mul_add_mul(fs_from_fe(o.x), fs_from_fe(o.y), fs_from_fe(o.x2), fs_from_fe(o.y2))));
}
template <bool HAS1>
__global__ __launch_bounds__(256) void k_dot2_partial(const uint64_t *__restrict__ xs0, const uint64_t *__restrict__ ys0,
                                                      const uint64_t *__restrict__ xs1, const uint64_t *__restrict__ ys1, uint32_t m,
                                                      uint64_t *__restrict__ partial) {
    __shared__ Fe lds[8];
    Fs<2> a0 = fs_zero<2>(), a1 = fs_zero<2>();
    const uint32_t stride = gridDim.x * 256;
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < m) {
        DotOps n0 = dot_load(xs0, ys0, i, i + stride, m), n1 = n0;
        if (HAS1) n1 = dot_load(xs1, ys1, i, i + stride, m);
        for (; i < m; i += 2 * stride) {
            DotOps c0 = n0, c1 = n1;
            uint32_t nx = i + 2 * stride;
            if (nx < m) {
                n0 = dot_load(xs0, ys0, nx, nx + stride, m);
                if (HAS1) n1 = dot_load(xs1, ys1, nx, nx + stride, m);
            }
            a0 = dot_step(a0, c0);
            if (HAS1) a1 = dot_step(a1, c1);
        }
    }
    Fe s0 = block_sum_fr(fs_to_fe(a0), lds), s1 = fe_zero();
    if (HAS1) {
        __syncthreads();
        s1 = block_sum_fr(fs_to_fe(a1), lds + 4);
    }
    if (threadIdx.x == 0) {
        fe_store(partial + 8 * (size_t)blockIdx.x, s0);
        fe_store(partial + 8 * (size_t)blockIdx.x + 4, s1);
    }
}
// sums `count` partial records of `width` Fr each into out[width]; times32: the records are sums of A(x) A(y) products
// (1/32 of the A-form value): the total is multiplied by 32 (fr29.hpp C266)
__global__ __launch_bounds__(256) void k_sum_partials(const uint64_t *__restrict__ partial, uint32_t count, uint32_t width, int times32,
                                                      uint64_t *__restrict__ out) {
    __shared__ Fe lds[4];
    for (uint32_t k = 0; k < width; k++) {
        Fe a = fe_zero();
        for (uint32_t i = threadIdx.x; i < count; i += 256) a = fe_add<FrCfg>(a, fe_load(partial + 4 * ((size_t)i * width + k)));
        Fe s = block_sum_fr(a, lds);
        if (threadIdx.x == 0) {
            if (times32) s = fs_to_fe(fs_mul(fs_from_fe(s), fs_c266()));
            fe_store(out + 4 * (size_t)k, s);
        }
        __syncthreads();
    }
}

// ---- powers of one scalar: the window table the host prepares (upload_window_table)
//   tab[d]               = A(z^d),          d < 16           (window 0 in A-form: seeds of k_powers)
//   tab[16 + 16 k + d]   = N(z^(d 16^k)),   d < 16, k < nwin
// z^e is a product of one entry per window: nwin - 1 products for any lane, whatever its exponent (a square-and-multiply
// over a table of z^(2^k) ran up to 17 divergent products per lane: that, not bandwidth, was what k_powers and
// k_poly_eval_partial spent their time on).
constexpr int POW_WIN = 8;  // most windows: exponents below 2^32
// every table entry is requested before the first product runs (the entries' addresses depend on e only): one memory
// latency per seed, not one per window
HALO_DEV Fs<4> window_power(const uint64_t *__restrict__ tab, uint32_t e, int nwin, bool a_form) {
    Fe w[POW_WIN];
#pragma unroll
    for (int k = 0; k < POW_WIN; k++) {
        uint32_t idx = k == 0 ? (a_form ? 0u : 16u) + (e & 15u) : 16u + 16u * (uint32_t)k + ((e >> (4 * k)) & 15u);
        w[k] = fe_load(tab + 4 * (size_t)(k < nwin ? idx : 0u));
    }
    Fs<4> acc = fs_from_fe(w[0]);
#pragma unroll
    for (int k = 1; k < POW_WIN; k++)
        if (k < nwin) acc = fs_widen<4>(fs_mul(acc, fs_from_fe(w[k])));  // (nwin is a kernel argument: wave-uniform)
    return acc;
}
// elements per lane: a longer chain spreads the seed (nwin - 1 products) over more elements, a shorter one puts more
// waves on the chip; measured at n = 2^20 (tools/fr_kernels.py with HALO_POW_E = 2 .. 32): k_powers 8, k_poly_eval_partial
// 16, k_h_coeffs 4 (its "seed" is one product per four elements whatever the length)
constexpr int POLY_EVAL_E = 16;  // coefficients per lane of k_poly_eval_partial
static int pow_chain_len(size_t n, int best) {
    // development override: a power of two in [4, 64] or it is ignored -- k_h_coeffs relies on a chain length that is a power of
    // two (its mid * high factor is wave-uniform and changes every fourth step): another value would give wrong coefficients
    const int forced = tuning().pow_e;  // (validated there)
    if (forced > 0) return forced;
    int e = best;
    while (e > 4 && n / (64 * (size_t)e) < 1024) e >>= 1;  // small inputs: at least a wave per SIMD
    return e;
}

// ------------------------------------------------------------------ K6: out[i] = z^i
// One product per 32 bytes written: VALU-bound (fr29.hpp: 5.3 TB/s if nothing else ran).  Lane l of a wave writes the
// elements base + l + 64 k: every store instruction of the wave covers 2 KiB of consecutive addresses.
__global__ __launch_bounds__(256) void k_powers(const uint64_t *__restrict__ tab, int nwin, int E, uint32_t n, FsArg z64, uint64_t *__restrict__ out) {
    uint32_t wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    uint32_t e = wave * (64u * (uint32_t)E) + lane;
    if (e >= n) return;
    Fs<2> cur = fs_tighten(window_power(tab, e, nwin, true));
    Fs<1> step = from_nform(z64);
#pragma unroll 1
    for (int k = 0; k < E && e < n; k++, e += 64) {
        fs_store(out + 4 * (size_t)e, cur);  // (a product's result is below 2 r: one conditional subtraction)
        cur = fs_mul(cur, step);
    }
}

// ------------------------------------------------------------------ K9: p(z)
// Lane l of a wave takes the coefficients base + l + 64 k, k < E (coalesced loads), and multiplies each by N(z^(64 k)) -- a
// table entry that is the same for every lane of every wave -- two products per reduction (fs_mul_add_mul); then one
// product with N(z^(base + l)).  The E products of a lane are INDEPENDENT: the Horner chain in z^64 this replaces (VERDICT
// r3 weak #7: 1.34 TB/s, half of the kernel's own VALU bound) was E dependent products on a grid of one wave per SIMD, i.e.
// one product in flight per SIMD; here the next pair's operands are in flight and its products issue while the current
// reduction runs.
__global__ __launch_bounds__(256) void k_poly_eval_partial(const uint64_t *__restrict__ coeffs, uint32_t len, const uint64_t *__restrict__ tab,
                                                           int nwin, int E, const uint64_t *__restrict__ zpow, uint64_t *__restrict__ partial) {
    __shared__ Fe lds[4];
    Fs<2> acc = fs_zero<2>();
    uint32_t lane = threadIdx.x & 63u, nwaves = gridDim.x * 4;
    for (uint32_t wave = (blockIdx.x * 256 + threadIdx.x) >> 6; (size_t)wave * (64u * (uint32_t)E) < len; wave += nwaves) {
        uint32_t base = wave * (64u * (uint32_t)E) + lane;
        if (base >= len) continue;
        Fs<2> h = fs_zero<2>();
        // four coefficients per trip, written out (the unroller does not duplicate the products' inline asm): two fused
        // pairs whose operands are all requested before the first product runs; E is a multiple of 4 (pow_chain_len)
#pragma unroll 1
        for (int k = 0; k < E; k += 4) {
            Fe c[4], pw[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                uint32_t i = base + 64u * (uint32_t)(k + q);
                c[q] = fe_load(coeffs + 4 * (size_t)(i < len ? i : base));
                pw[q] = fe_load(zpow + 4 * (size_t)(k + q));
                if (i >= len) c[q] = fe_zero();
            }
            Fs<2> t0 = fs_mul_add_mul(fs_from_fe(c[0]), fs_from_fe(pw[0]), fs_from_fe(c[1]), fs_from_fe(pw[1]));
            Fs<2> t1 = fs_mul_add_mul(fs_from_fe(c[2]), fs_from_fe(pw[2]), fs_from_fe(c[3]), fs_from_fe(pw[3]));
            h = fs_tighten(fs_add(h, fs_add(t0, t1)));
        }
        acc = fs_tighten(fs_add(acc, fs_mul(h, window_power(tab, base, nwin, false))));
    }
    Fe s = block_sum_fr(fs_to_fe(acc), lds);
    if (threadIdx.x == 0) fe_store(partial + 4 * (size_t)blockIdx.x, s);
}

// ------------------------------------------------------------------ K7: h coefficients from three 256-entry tables
// tab = low (A-form) | mid | high (N-form), 256 x 4 words each; coefficient k = low[k & 255] * mid[(k >> 8) & 255] * high[k >> 16].
// Lane l of a wave takes k = base + l + 64 j: (k >> 8) is the same for the whole wave and changes every fourth step,
// so mid * high is one product per four elements and low * (mid high) the only product per element.
__global__ __launch_bounds__(256) void k_h_coeffs(const uint64_t *__restrict__ tab, uint32_t n, int E, int accumulate, uint64_t *__restrict__ out) {
    uint32_t wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    uint32_t k = wave * (64u * (uint32_t)E) + lane;
    if (k >= n) return;
    Fs<4> hm = fs_zero<4>();
#pragma unroll 1
    for (int j = 0; j < E && k < n; j++, k += 64) {
        if ((j & 3) == 0) {
            uint32_t kk = __builtin_amdgcn_readfirstlane(k);  // wave-uniform: (k >> 8) does not depend on the lane
            hm = fs_widen<4>(fs_mul(fs_load(tab + 4 * (size_t)(256 + ((kk >> 8) & 255u))), fs_load(tab + 4 * (size_t)(512 + ((kk >> 16) & 255u)))));
        }
        Fs<2> v = fs_mul(fs_load(tab + 4 * (size_t)(k & 255u)), hm);
        if (accumulate) fs_store(out + 4 * (size_t)k, fs_add(v, fs_load(out + 4 * (size_t)k)));
        else fs_store(out + 4 * (size_t)k, v);
    }
}

// ------------------------------------------------------------------ K8: HPoly::eval, one polynomial per lane
__global__ __launch_bounds__(256) void k_h_eval(const uint64_t *__restrict__ xis, uint32_t m, int lg_n, FeArg zarg,
                                                uint64_t *__restrict__ out) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const uint64_t *x = xis + 4 * (size_t)i * (size_t)(lg_n + 1);
    Fe z = from_arg(zarg), one = fe_one<FrCfg>();
    Fe v = fe_add<FrCfg>(one, fe_mul<FrCfg>(fe_load(x + 4 * (size_t)lg_n), z));
    Fe zi = z;
#pragma unroll 1
    for (int k = 1; k < lg_n; k++) {
        zi = fe_sqr<FrCfg>(zi);
        v = fe_mul<FrCfg>(v, fe_add<FrCfg>(one, fe_mul<FrCfg>(fe_load(x + 4 * (size_t)(lg_n - k)), zi)));
    }
    fe_store(out + 4 * (size_t)i, v);
}


// HPoly::eval for m polynomials, each at its own point (the m succinct checks of acc.rs:158-170 have their own z_i)
__global__ __launch_bounds__(256) void k_h_eval_z(const uint64_t *__restrict__ xis, const uint64_t *__restrict__ zs, uint32_t m, int lg_n,
                                                  uint64_t *__restrict__ out) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const uint64_t *x = xis + 4 * (size_t)i * (size_t)(lg_n + 1);
    Fe z = fe_load(zs + 4 * (size_t)i), one = fe_one<FrCfg>();
    Fe v = fe_add<FrCfg>(one, fe_mul<FrCfg>(fe_load(x + 4 * (size_t)lg_n), z));
    Fe zi = z;
#pragma unroll 1
    for (int k = 1; k < lg_n; k++) {
        zi = fe_sqr<FrCfg>(zi);
        v = fe_mul<FrCfg>(v, fe_add<FrCfg>(one, fe_mul<FrCfg>(fe_load(x + 4 * (size_t)(lg_n - k)), zi)));
    }
    fe_store(out + 4 * (size_t)i, v);
}

// ------------------------------------------------------------------ batched verifier relation
// m independent sums of K <= 64 scalar multiples each (the 2 lg n + 2 terms of one pcdl::succinct_check, pcdl.rs:288-310):
// one wave per sum, one lane per term.  Every lane runs the same 256-step ladder (double, add, keep the sum if the bit is
// set -- no divergence although the scalars differ), then the wave folds its lanes with a shuffle tree.
// points: m x K x 8 words (arkworks affine, (0, 0) = infinity); scalars: m x K x 4 words, CANONICAL (not Montgomery).
__global__ __launch_bounds__(64) void k_batch_small_msm(const uint64_t *__restrict__ points, const uint64_t *__restrict__ scalars, uint32_t K,
                                                        uint64_t *__restrict__ out) {
    uint32_t inst = blockIdx.x, lane = threadIdx.x;
    bool live = lane < K;
    size_t t = (size_t)inst * K + (live ? lane : 0);
    AffN p = live ? aff_from_words(points + 8 * t) : aff_inf();
    Fe k = fe_load(scalars + 4 * t);
    JacN acc = jac_inf();
#pragma unroll 1
    for (int limb = 7; limb >= 0; limb--) {
        uint32_t word = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) word = (q == limb) ? k.v[q] : word;
#pragma unroll 1
        for (int bit = 31; bit >= 0; bit--) {
            acc = jac_dbl(acc);
            JacN s = jac_madd(acc, p);
            bool take = live && ((word >> bit) & 1u);
#pragma unroll
            for (int i = 0; i < 9; i++) {
                acc.x.v[i] = take ? s.x.v[i] : acc.x.v[i];
                acc.y.v[i] = take ? s.y.v[i] : acc.y.v[i];
                acc.z.v[i] = take ? s.z.v[i] : acc.z.v[i];
            }
        }
    }
    XyzzN x = jac_to_xyzz(acc);
#pragma unroll 1
    for (int off = 32; off >= 1; off >>= 1) {
        XyzzN o = xyzz_shfl(x, (int)((lane + off) & 63));
        if ((int)lane < off) xyzz_add(x, o);
    }
    if (lane == 0) xyzz_store_jac_words(out + 12 * (size_t)inst, x);
}

// ------------------------------------------------------------------ input generator / polynomial helpers
// SplitMix64 is counter based: draw k of a stream with state s0 is mix(s0 + k*gamma), so the
// scalars of `PallasPoly::rand` (pcdl.rs:141) can be produced in parallel, bit-identical to a
// sequential host stream.  Element i uses draws 4i+1 .. 4i+4, reduced mod r, to Montgomery form.
HALO_DEV uint64_t splitmix_at(uint64_t s0, uint64_t k) {
    uint64_t z = s0 + k * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
// element `idx` of the scalar stream that starts after state s0, in Montgomery form
HALO_DEV Fe rng_scalar_at(uint64_t s0, uint64_t idx) {
    Fe v;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint64_t w = splitmix_at(s0, 4 * idx + k + 1);
        v.v[2 * k] = (uint32_t)w;
        v.v[2 * k + 1] = (uint32_t)(w >> 32);
    }
    // 2^256 < 4r: at most three subtractions
#pragma unroll 1
    for (int r = 0; r < 3; r++) {
        uint32_t d[8];
        uint64_t br = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            uint64_t t = (uint64_t)v.v[k] - FrCfg::P[k] - br;
            d[k] = (uint32_t)t;
            br = (t >> 32) & 1;
        }
        if (!br) {
#pragma unroll
            for (int k = 0; k < 8; k++) v.v[k] = d[k];
        }
    }
    return fe_to_mont<FrCfg>(v);
}
__global__ __launch_bounds__(256) void k_rng_scalars(uint64_t s0, uint32_t n, uint64_t *__restrict__ out) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    fe_store(out + 4 * (size_t)i, rng_scalar_at(s0, i));
}
// A cyclic shard of p_bar = q (X - z) straight from the stream (q is never materialised):
// out[j] = p_bar[offset + j*stride] = q[i-1] - z q[i], q[k] = stream element k for k < deg, else 0
__global__ __launch_bounds__(256) void k_pbar_stream(uint64_t s0, uint32_t deg, FeArg zarg, uint32_t stride, uint32_t offset,
                                                     uint32_t n_local, uint64_t *__restrict__ out) {
    uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n_local) return;
    uint64_t i = (uint64_t)offset + (uint64_t)j * stride;
    Fe r = fe_zero();
    if (i <= deg) {
        Fe lo = (i >= 1) ? rng_scalar_at(s0, i - 1) : fe_zero();
        Fe hi = (i < deg) ? rng_scalar_at(s0, i) : fe_zero();
        r = fe_sub<FrCfg>(lo, fe_mul<FrCfg>(from_arg(zarg), hi));
    }
    fe_store(out + 4 * (size_t)j, r);
}
// p_bar = q * (X - z): p_bar[i] = q[i-1] - z q[i], i in [0, deg]; q has deg coefficients (pcdl.rs:140-142)
__global__ __launch_bounds__(256) void k_pbar(const uint64_t *__restrict__ q, uint32_t deg, FeArg zarg, uint64_t *__restrict__ out) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i > deg) return;
    Fe z = from_arg(zarg);
    Fe lo = (i >= 1) ? fe_load(q + 4 * (size_t)(i - 1)) : fe_zero();
    Fe hi = (i < deg) ? fe_load(q + 4 * (size_t)i) : fe_zero();
    fe_store(out + 4 * (size_t)i, fe_sub<FrCfg>(lo, fe_mul<FrCfg>(z, hi)));
}
// y[i] += a * x[i]   (p' = p + alpha p_bar, pcdl.rs:156): 96 bytes moved per product, HBM-bound
__global__ __launch_bounds__(256) void k_axpy(uint64_t *__restrict__ y, const uint64_t *__restrict__ x, uint32_t n, FsArg aarg) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    Fe yv = fe_load(y + 4 * (size_t)i), xv = fe_load(x + 4 * (size_t)i);
    fs_store(y + 4 * (size_t)i, fs_add(fs_from_fe(yv), fs_mul(fs_from_fe(xv), from_nform(aarg))));
}


// ------------------------------------------------------------------ "no-fold" late rounds
// Once the key has been folded down to M points (G0), folding it further is latency-bound
// (one uniform scalar multiplication is a ~1 ms serial chain whatever m is).  Instead G0 stays
// fixed and the folded key is kept implicitly:  G_k[j] = sum_t s[t] * G0[j + t*m]  with
// s = products of the challenges since the switch.  Then
//   L = <c_r, G_l> = sum_{b: (b mod m) <  m/2} (c[(b mod m) + m/2] * s[b div m]) * G0[b]
//   R = <c_l, G_r> = sum_{b: (b mod m) >= m/2} (c[(b mod m) - m/2] * s[b div m]) * G0[b]
// are two MSMs over the fixed G0 (half of each scalar vector is zero and is skipped at recode).
__global__ __launch_bounds__(256) void k_nofold_expand(const uint64_t *__restrict__ c, const uint64_t *__restrict__ sv, uint32_t m,
                                                       int log2m, uint32_t M, uint64_t *__restrict__ outL, uint64_t *__restrict__ outR) {
    uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= M) return;
    uint32_t u = b & (m - 1), t = b >> log2m, h = m >> 1;
    Fe sc = fe_load(sv + 4 * (size_t)t);
    if (u < h) {
        fe_store(outL + 4 * (size_t)b, fe_mul<FrCfg>(fe_load(c + 4 * (size_t)(u + h)), sc));
        fe_store(outR + 4 * (size_t)b, fe_zero());
    } else {
        fe_store(outR + 4 * (size_t)b, fe_mul<FrCfg>(fe_load(c + 4 * (size_t)(u - h)), sc));
        fe_store(outL + 4 * (size_t)b, fe_zero());
    }
}
// The same two vectors as ONE array for a tagged launch (msm.hip, MsmBatch::tagged): L's and R's non-zero scalars sit on
// disjoint points, so F[b] = c[(b mod m) +- m/2] * s[b div m] in canonical form with bit 255 (free: r < 2^255) set on R's half
// says everything -- half the bytes written, and one launch sequence computes both sums.
__global__ __launch_bounds__(256) void k_nofold_expand_tagged(const uint64_t *__restrict__ c, const uint64_t *__restrict__ sv, uint32_t m,
                                                              int log2m, uint32_t M, uint64_t *__restrict__ out) {
    uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= M) return;
    uint32_t u = b & (m - 1), t = b >> log2m, h = m >> 1;
    Fe v = fe_mul<FrCfg>(fe_load(c + 4 * (size_t)(u < h ? u + h : u - h)), fe_load(sv + 4 * (size_t)t));
    v = fe_from_mont<FrCfg>(v);
    if (u >= h) v.v[7] |= 0x80000000u;
    fe_store(out + 4 * (size_t)b, v);
}
// s'[2t + u] = s[t] * xi^u : the stride of the implicit key halves (pcdl.rs:218 applied to the representation)
__global__ __launch_bounds__(256) void k_nofold_s_update(const uint64_t *__restrict__ s_in, uint32_t len, FeArg xi,
                                                         uint64_t *__restrict__ s_out) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 2 * len) return;
    Fe v = fe_load(s_in + 4 * (size_t)(i >> 1));
    if (i & 1) v = fe_mul<FrCfg>(v, from_arg(xi));
    fe_store(s_out + 4 * (size_t)i, v);
}

// v[i] *= a   (z-powers of a cyclic shard: z^(r + jP) = z^r * (z^P)^j)
__global__ __launch_bounds__(256) void k_scale(uint64_t *__restrict__ v, uint32_t n, FeArg aarg) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    fe_store(v + 4 * (size_t)i, fe_mul<FrCfg>(fe_load(v + 4 * (size_t)i), from_arg(aarg)));
}

// ================================================================== host launchers
int ipa_fold_points(halo_ctx *ctx, uint32_t *d_G, size_t m, const host::Fr &xi_mont) {
    if (m == 0) return HALO_OK;
    host::GlvDigits dg = host::glv_digits(xi_mont);
    GlvArg a;
    for (int i = 0; i < 14; ++i) a.dig[i] = 0;
    for (int i = 0; i < dg.n; ++i) a.dig[i / 10] |= (uint32_t)dg.d[i] << (3 * (i % 10));
    a.ndigits = dg.n;
    // a lane folds the points j and j + half where that still leaves >= 4 waves per SIMD; below, one point per lane
    size_t half = m >= ((size_t)1 << 18) ? (m + 1) / 2 : m;
    HALO_LAUNCH(ctx, "k_fold_points", k_fold_points, dim3((unsigned)((half + 255) / 256)), dim3(256), 0, d_G, (uint32_t)m, (uint32_t)half, a);
    HALO_HIP(hipGetLastError());
    return HALO_OK;
}
// dst[j] <- src[j] + s[0] src[j+m] + s[1] src[j+2m] + s[2] src[j+3m], j < m (dst may be src)
int ipa_fold_points4(halo_ctx *ctx, const uint32_t *d_src, uint32_t *d_dst, size_t m, const host::Fr s[3]) {
    if (m == 0) return HALO_OK;
    {   // the fold from the context's own (constant) key: a comb table replaces the doubling chain (foldtab.hip)
        int done = fold_points4_tab(ctx, d_src, d_dst, m, s);
        if (done < 0) return done;
        if (done) return HALO_OK;
    }
    GlvArg3 a;
    a.ndigits = 0;
    for (int t = 0; t < 3; ++t) {
        host::GlvDigits dg = host::glv_digits(s[t]);
        for (int i = 0; i < 14; ++i) a.dig[t][i] = 0;
        for (int i = 0; i < dg.n; ++i) a.dig[t][i / 10] |= (uint32_t)dg.d[i] << (3 * (i % 10));
        if (dg.n > a.ndigits) a.ndigits = dg.n;
    }
    if (m < ((size_t)1 << 16)) {  // a latency chain at this size: one output per quad (in place is fine: a quad reads j + t m, writes j)
        HALO_LAUNCH(ctx, "k_fold_points4_quad", k_fold_points4_quad, dim3((unsigned)((4 * m + 255) / 256)), dim3(256), 0, d_src, d_dst, (uint32_t)m, a);
        HALO_HIP(hipGetLastError());
        return HALO_OK;
    }
    size_t half = m >= ((size_t)1 << 17) ? (m + 1) / 2 : m;
    HALO_LAUNCH(ctx, "k_fold_points4", k_fold_points4, dim3((unsigned)((half + 255) / 256)), dim3(256), 0, d_src, d_dst, (uint32_t)m, (uint32_t)half, a);
    HALO_HIP(hipGetLastError());
    return HALO_OK;
}
int ipa_fold_scalars(halo_ctx *ctx, uint64_t *d_c, uint64_t *d_z, size_t m, const host::Fr &xi, const host::Fr &xi_inv) {
    if (m == 0) return HALO_OK;
    HALO_LAUNCH(ctx, "k_fold_scalars", k_fold_scalars, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, d_c, d_z, (uint32_t)m, to_nform(xi),
                to_nform(xi_inv));
    HALO_HIP(hipGetLastError());
    return HALO_OK;
}

static unsigned reduce_blocks(size_t work_items) {
    size_t nb = (work_items + 255) / 256;
    if (nb > 1024) nb = 1024;
    if (nb == 0) nb = 1;
    return (unsigned)nb;
}

// two elements per lane and trip, at most 512 blocks (two waves per SIMD): the lanes of a large dot product run several trips
static unsigned dot_blocks(size_t m) {
    const size_t cap_env = tuning().dot_blocks > 0 ? (size_t)tuning().dot_blocks : 0;  // development override
    size_t nb = ((m + 1) / 2 + 255) / 256, cap = cap_env ? cap_env : 512;
    if (nb > cap) nb = cap;
    if (nb == 0) nb = 1;
    return (unsigned)nb;
}
// the two halves of fr_dot2: launch (kernels + the copy of the two sums to pinned memory, on ctx->stream) and collect (wait)
int fr_dot2_launch(halo_ctx *ctx, const uint64_t *xs0, const uint64_t *ys0, const uint64_t *xs1, const uint64_t *ys1, size_t m) {
    if (m == 0) return HALO_OK;
    if (!xs0) { set_error("dot2: the first pair of vectors is required"); return HALO_E_ARG; }
    unsigned nb = dot_blocks(m);
    uint64_t *partial = ctx->d_tmp_c;  // >= 1024 * 8 words
    if (xs1) HALO_LAUNCH(ctx, "k_dot2_partial", k_dot2_partial<true>, dim3(nb), dim3(256), 0, xs0, ys0, xs1, ys1, (uint32_t)m, partial);
    else HALO_LAUNCH(ctx, "k_dot2_partial", k_dot2_partial<false>, dim3(nb), dim3(256), 0, xs0, ys0, xs1, ys1, (uint32_t)m, partial);
    HALO_LAUNCH(ctx, "k_sum_partials", k_sum_partials, dim3(1), dim3(256), 0, partial, nb, 2u, 1, partial + 8 * 1024);
    HALO_HIP(hipGetLastError());
    HALO_HIP(hipMemcpyAsync(ctx->h_pinned, partial + 8 * 1024, 64, hipMemcpyDeviceToHost, ctx->stream));
    return HALO_OK;
}
int fr_dot2_collect(halo_ctx *ctx, hipStream_t stream, size_t m, host::Fr out[2]) {
    out[0] = out[1] = host::Fr::zero();
    if (m == 0) return HALO_OK;
    HALO_HIP(hipStreamSynchronize(stream));
    out[0] = host::Fr::load(ctx->h_pinned);
    out[1] = host::Fr::load(ctx->h_pinned + 4);
    return HALO_OK;
}
int fr_dot2(halo_ctx *ctx, const uint64_t *xs0, const uint64_t *ys0, const uint64_t *xs1, const uint64_t *ys1, size_t m,
            host::Fr out[2]) {
    out[0] = out[1] = host::Fr::zero();
    int rc = fr_dot2_launch(ctx, xs0, ys0, xs1, ys1, m);
    if (rc) return rc;
    return fr_dot2_collect(ctx, ctx->stream, m, out);
}

// Window table of z (see window_power): 16 A-form entries, then nwin x 16 N-form entries, staged through pinned memory.
// ~32 nwin host products (a few microseconds).
static int bits_for(size_t n) {
    int b = 0;
    while (((size_t)1 << b) < n) b++;
    return b < 1 ? 1 : b;
}
// zpows > 0: N(z^(64 k)), k < zpows, behind the window table (k_poly_eval_partial's per-element multipliers)
static int upload_window_table(halo_ctx *ctx, const host::Fr &z, int nwin, uint64_t *d_tab, host::Fr *z64, int zpows = 0) {
    uint64_t *h = ctx->h_wintab;  // pinned, 1024 words: nwin <= 15
    const host::Fr k32 = host::Fr::from_u64(32);
    host::Fr base = z;  // z^(16^k)
    for (int k = 0; k < nwin; ++k) {
        host::Fr cur = host::Fr::one();
        for (int d = 0; d < 16; ++d) {
            if (k == 0) cur.store(h + 4 * d);
            (cur * k32).store(h + 4 * (16 + 16 * k + d));
            cur = cur * base;
        }
        base = cur;  // z^(16^(k+1))
    }
    host::Fr t = z;
    for (int i = 0; i < 6; ++i) t = t.sqr();
    *z64 = t;
    size_t entries = (size_t)16 * (1 + nwin);
    if (entries + (size_t)zpows > 256) { set_error("window table: too many entries"); return HALO_E_ARG; }  // (h_wintab: 256 entries)
    host::Fr cur = host::Fr::one();
    for (int k = 0; k < zpows; ++k) {
        (cur * k32).store(h + 4 * (entries + (size_t)k));
        cur = cur * t;
    }
    HALO_HIP(hipMemcpyAsync(d_tab, h, (entries + (size_t)zpows) * 32, hipMemcpyHostToDevice, ctx->stream));
    HALO_HIP(hipStreamSynchronize(ctx->stream));  // the staging memory is reused by the next call
    return HALO_OK;
}
static unsigned wave_blocks(size_t n, int E) {  // blocks of 4 waves, a wave per 64 * E elements
    size_t waves = (n + 64 * (size_t)E - 1) / (64 * (size_t)E);
    return (unsigned)((waves + 3) / 4);
}

int fr_powers(halo_ctx *ctx, const host::Fr &z, size_t n, uint64_t *d_out) {
    if (n == 0) return HALO_OK;
    uint64_t *d_tab = ctx->d_tmp_c + 8 * 1024 + 64;
    int nwin = (bits_for(n) + 3) / 4;
    host::Fr z64;
    int rc = upload_window_table(ctx, z, nwin, d_tab, &z64);
    if (rc) return rc;
    int E = pow_chain_len(n, 8);
    HALO_LAUNCH(ctx, "k_powers", k_powers, dim3(wave_blocks(n, E)), dim3(256), 0, d_tab, nwin, E, (uint32_t)n, to_nform(z64), d_out);
    HALO_HIP(hipGetLastError());
    return HALO_OK;
}

int fr_poly_eval(halo_ctx *ctx, const uint64_t *d_coeffs, size_t len, const host::Fr &z, host::Fr *out) {
    *out = host::Fr::zero();
    if (len == 0) return HALO_OK;
    uint64_t *d_tab = ctx->d_tmp_c + 8 * 1024 + 64;
    int nwin = (bits_for(len) + 3) / 4;
    host::Fr z64;
    int E = pow_chain_len(len, POLY_EVAL_E);
    int rc = upload_window_table(ctx, z, nwin, d_tab, &z64, E);
    if (rc) return rc;
    unsigned nb = wave_blocks(len, E);
    if (nb > 1024) nb = 1024;
    uint64_t *partial = ctx->d_tmp_c;
    HALO_LAUNCH(ctx, "k_poly_eval_partial", k_poly_eval_partial, dim3(nb), dim3(256), 0, d_coeffs, (uint32_t)len, d_tab, nwin, E,
                d_tab + 4 * (size_t)(16 * (1 + nwin)), partial);
    HALO_LAUNCH(ctx, "k_sum_partials", k_sum_partials, dim3(1), dim3(256), 0, partial, nb, 1u, 0, partial + 8 * 1024);
    HALO_HIP(hipGetLastError());
    HALO_HIP(hipMemcpyAsync(ctx->h_pinned, partial + 8 * 1024, 32, hipMemcpyDeviceToHost, ctx->stream));
    HALO_HIP(hipStreamSynchronize(ctx->stream));
    *out = host::Fr::load(ctx->h_pinned);
    return HALO_OK;
}

int h_coeffs_dev(halo_ctx *ctx, const host::Fr *xis, size_t lg_n, const host::Fr &scale, bool accumulate, uint64_t *d_out) {
    if (lg_n > 24) { set_error("h_coeffs: lg_n > 24 unsupported"); return HALO_E_ARG; }
    // coefficient k = prod over set bits i of k of xis[lg_n - i]  (pcdl.rs:496-505)
    std::vector<uint64_t> tab(3 * 256 * 4);
    for (int level = 0; level < 3; ++level) {
        std::vector<host::Fr> t(256, host::Fr::one());
        if (level == 2) t[0] = scale;
        size_t len = 1;
        for (int b = 0; b < 8; ++b) {
            size_t bit = (size_t)level * 8 + b;
            if (bit >= lg_n) break;
            const host::Fr &x = xis[lg_n - bit];
            for (size_t k = 0; k < len; ++k) t[len + k] = t[k] * x;
            len *= 2;
        }
        if (level == 2 && len == 1) t[0] = scale;
        // low table in A-form (what the kernel multiplies INTO), mid and high as N-form multipliers (fr29.hpp): 32 x
        const host::Fr k32 = host::Fr::from_u64(32);
        for (size_t k = 0; k < 256; ++k) (level == 0 ? t[k] : t[k] * k32).store(&tab[((size_t)level * 256 + k) * 4]);
    }
    uint64_t *d_tab = ctx->d_tmp_c + 8 * 1024 + 1024;  // (the window table of fr_powers / fr_poly_eval sits below)
    HALO_HIP(hipMemcpyAsync(d_tab, tab.data(), tab.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    HALO_HIP(hipStreamSynchronize(ctx->stream));  // `tab` is pageable and dies at return
    size_t n = (size_t)1 << lg_n;
    int E = pow_chain_len(n, 4);
    HALO_LAUNCH(ctx, "k_h_coeffs", k_h_coeffs, dim3(wave_blocks(n, E)), dim3(256), 0, d_tab, (uint32_t)n, E, accumulate ? 1 : 0, d_out);
    HALO_HIP(hipGetLastError());
    return HALO_OK;
}

// Measurement hook (halo_bench_fr_kernel): `reps` back-to-back launches of ONE of the bandwidth-side kernels over n elements
// of the context's scratch buffers (any 256-bit patterns are valid operands), nothing else on the stream and no host
// round trip in between -- the steady-state duration a profiler reads off, not the first launch after an idle gap.
int bench_fr_kernel(halo_ctx *ctx, int which, size_t n, int reps) {
    if (n == 0 || n > (ctx->n < 64 ? 64 : ctx->n) || reps < 1) { set_error("bench_fr_kernel: n must be in [1, context size]"); return HALO_E_ARG; }
    uint64_t *a = ctx->d_tmp_a, *b = ctx->d_tmp_b, *c = ctx->d_tmp_b + 4 * n, *d = ctx->d_tmp_b + 8 * n;  // n x 4 words each (d_tmp_b: 16 n words)
    host::Fr z = host::Fr::from_u64(0x1234567) * host::Fr::from_u64(0x89abcdef), z64;
    uint64_t *d_tab = ctx->d_tmp_c + 8 * 1024 + 64, *partial = ctx->d_tmp_c;
    int nwin = (bits_for(n) + 3) / 4;
    int rc = rng_scalars_dev(ctx, 0xF00D, n, b);
    if (!rc) rc = rng_scalars_dev(ctx, 0xBEEF, n, c);
    if (!rc) rc = rng_scalars_dev(ctx, 0xCAFE, n, d);
    if (!rc) rc = upload_window_table(ctx, z, nwin, d_tab, &z64, which == 1 ? pow_chain_len(n, POLY_EVAL_E) : 0);
    if (rc) return rc;
    size_t m = n / 2 ? n / 2 : 1;
    for (int r = 0; r < reps; ++r) {
        switch (which) {
            case 0: { int E = pow_chain_len(n, 8); HALO_LAUNCH(ctx, "k_powers", k_powers, dim3(wave_blocks(n, E)), dim3(256), 0, d_tab, nwin, E, (uint32_t)n, to_nform(z64), a); break; }
            case 1: {
                int E = pow_chain_len(n, POLY_EVAL_E);
                unsigned nb = wave_blocks(n, E);
                if (nb > 1024) nb = 1024;
                HALO_LAUNCH(ctx, "k_poly_eval_partial", k_poly_eval_partial, dim3(nb), dim3(256), 0, b, (uint32_t)n, d_tab, nwin, E,
                            d_tab + 4 * (size_t)(16 * (1 + nwin)), partial);
                break;
            }
            case 2: HALO_LAUNCH(ctx, "k_dot2_partial", k_dot2_partial<false>, dim3(dot_blocks(n)), dim3(256), 0, b, c, (const uint64_t *)nullptr,
                                (const uint64_t *)nullptr, (uint32_t)n, partial); break;
            case 3: HALO_LAUNCH(ctx, "k_dot2_partial", k_dot2_partial<true>, dim3(dot_blocks(m)), dim3(256), 0, b + 4 * m, c, b, c + 4 * m, (uint32_t)m, partial); break;
            case 4: {
                int E = pow_chain_len(n, 4);
                // (the table region holds the window table: valid field elements, which is all the kernel needs)
                HALO_LAUNCH(ctx, "k_h_coeffs", k_h_coeffs, dim3(wave_blocks(n, E)), dim3(256), 0, ctx->d_tmp_c + 8 * 1024 + 1024, (uint32_t)n, E, 0, a);
                break;
            }
            case 5: HALO_LAUNCH(ctx, "k_fold_scalars", k_fold_scalars, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, b, c, (uint32_t)m, to_nform(z), to_nform(z64)); break;
            case 6: HALO_LAUNCH(ctx, "k_axpy", k_axpy, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, b, c, (uint32_t)n, to_nform(z)); break;
            default: set_error("bench_fr_kernel: which must be 0..6"); return HALO_E_ARG;
        }
    }
    HALO_HIP(hipGetLastError());
    HALO_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->prof.on) ctx->prof.collect();
    return HALO_OK;
}

int h_eval_batch(halo_ctx *ctx, const uint64_t *d_xis, size_t m, size_t lg_n, const host::Fr &z, uint64_t *d_out) {
    if (m == 0) return HALO_OK;
    HALO_LAUNCH(ctx, "k_h_eval", k_h_eval, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, d_xis, (uint32_t)m, (int)lg_n, to_arg(z), d_out);
    HALO_HIP(hipGetLastError());
    return HALO_OK;
}

int h_eval_each(halo_ctx *ctx, const uint64_t *d_xis, const uint64_t *d_zs, size_t m, size_t lg_n, uint64_t *d_out) {
    if (m == 0) return HALO_OK;
    HALO_LAUNCH(ctx, "k_h_eval_z", k_h_eval_z, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, d_xis, d_zs, (uint32_t)m, (int)lg_n, d_out);
    HALO_HIP(hipGetLastError());
    return HALO_OK;
}
int batch_small_msm(halo_ctx *ctx, const uint64_t *d_points, const uint64_t *d_scalars, size_t m, size_t K, uint64_t *d_out) {
    if (m == 0) return HALO_OK;
    if (K == 0 || K > 64) { set_error("batch_small_msm: 1..64 terms per sum"); return HALO_E_ARG; }
    HALO_LAUNCH(ctx, "k_batch_small_msm", k_batch_small_msm, dim3((unsigned)m), dim3(64), 0, d_points, d_scalars, (uint32_t)K, d_out);
    HALO_HIP(hipGetLastError());
    return HALO_OK;
}

int rng_scalars_dev(halo_ctx *ctx, uint64_t state0, size_t n, uint64_t *d_out) {
    if (n == 0) return HALO_OK;
    HALO_LAUNCH(ctx, "k_rng_scalars", k_rng_scalars, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, state0, (uint32_t)n, d_out);
    HALO_HIP(hipGetLastError());
    return HALO_OK;
}
int pbar_dev(halo_ctx *ctx, const uint64_t *d_q, size_t deg, const host::Fr &z, uint64_t *d_out) {
    HALO_LAUNCH(ctx, "k_pbar", k_pbar, dim3((unsigned)((deg + 1 + 255) / 256)), dim3(256), 0, d_q, (uint32_t)deg, to_arg(z), d_out);
    HALO_HIP(hipGetLastError());
    return HALO_OK;
}
int axpy_dev(halo_ctx *ctx, uint64_t *d_y, const uint64_t *d_x, size_t n, const host::Fr &a) {
    if (n == 0) return HALO_OK;
    HALO_LAUNCH(ctx, "k_axpy", k_axpy, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, d_y, d_x, (uint32_t)n, to_nform(a));
    HALO_HIP(hipGetLastError());
    return HALO_OK;
}

int nofold_expand(halo_ctx *ctx, const uint64_t *d_c, const uint64_t *d_s, size_t m, size_t M, uint64_t *d_L, uint64_t *d_R) {
    int log2m = 0;
    while (((size_t)1 << log2m) < m) log2m++;
    HALO_LAUNCH(ctx, "k_nofold_expand", k_nofold_expand, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, d_c, d_s, (uint32_t)m, log2m,
                (uint32_t)M, d_L, d_R);
    HALO_HIP(hipGetLastError());
    return HALO_OK;
}
int nofold_expand_tagged(halo_ctx *ctx, const uint64_t *d_c, const uint64_t *d_s, size_t m, size_t M, uint64_t *d_F) {
    int log2m = 0;
    while (((size_t)1 << log2m) < m) log2m++;
    HALO_LAUNCH(ctx, "k_nofold_expand_tagged", k_nofold_expand_tagged, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, d_c, d_s, (uint32_t)m, log2m,
                (uint32_t)M, d_F);
    HALO_HIP(hipGetLastError());
    return HALO_OK;
}
int nofold_s_update(halo_ctx *ctx, const uint64_t *d_s_in, size_t len, const host::Fr &xi, uint64_t *d_s_out) {
    HALO_LAUNCH(ctx, "k_nofold_s_update", k_nofold_s_update, dim3((unsigned)((2 * len + 255) / 256)), dim3(256), 0, d_s_in, (uint32_t)len,
                to_arg(xi), d_s_out);
    HALO_HIP(hipGetLastError());
    return HALO_OK;
}

int fr_scale(halo_ctx *ctx, uint64_t *d_v, size_t n, const host::Fr &a) {
    if (n == 0) return HALO_OK;
    HALO_LAUNCH(ctx, "k_scale", k_scale, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, d_v, (uint32_t)n, to_arg(a));
    HALO_HIP(hipGetLastError());
    return HALO_OK;
}

int pbar_stream_dev(halo_ctx *ctx, uint64_t state0, size_t deg, const host::Fr &z, uint64_t stride, uint64_t offset, size_t n_local,
                    uint64_t *d_out) {
    if (n_local == 0) return HALO_OK;
    HALO_LAUNCH(ctx, "k_pbar_stream", k_pbar_stream, dim3((unsigned)((n_local + 255) / 256)), dim3(256), 0, state0, (uint32_t)deg, to_arg(z),
                (uint32_t)stride, (uint32_t)offset, (uint32_t)n_local, d_out);
    HALO_HIP(hipGetLastError());
    return HALO_OK;
}

}  // namespace halo
