// libhalo_hip_dev.so: what developers and the test-suite need and a production host does not (include/halo_accumulation_dev.h).
//
//   * the primitive test hooks (one field / group operation per lane, compared with the oracle by tests/test_gpu_parity.py),
//   * halo_bench_fr_kernel (back-to-back launches of one bandwidth-side kernel for the profiler),
//   * the per-context experiment knobs (window bits, task length, sort / fold / IPA strategy),
//   * halo_dev_hook: the fault injectors and forced test paths of csrc/tuning.hpp DevHooks.
//
// It links AGAINST libhalo_hip.so (one copy of the library's state in the process) and holds nothing the product path calls:
// `nm -D libhalo_hip.so` shows no halo_test_* / halo_bench_* / halo_dev_* symbol, and the product library reads no fault
// injector from the environment (tests/test_abi_cpu.py checks both).
#include <cstring>

#include "../../include/halo_accumulation_dev.h"
#include "curve_quad.hpp"
#include "internal.hpp"

namespace halo {

// ------------------------------------------------------------------------------ test hooks
template <class F>
__global__ __launch_bounds__(256) void k_test_field(int op, const uint64_t *a, const uint64_t *b, uint32_t n, uint64_t *out) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    Fe x = fe_load(a + 4 * (size_t)i);
    Fe y = b ? fe_load(b + 4 * (size_t)i) : fe_zero();
    Fe r;
    switch (op) {
        case 0: r = fe_mul<F>(x, y); break;
        case 1: r = fe_add<F>(x, y); break;
        case 2: r = fe_sub<F>(x, y); break;
        case 3: r = fe_is_zero(x) ? fe_zero() : fe_inv<F>(x); break;
        case 4: r = fe_from_mont<F>(x); break;
        default: r = fe_to_mont<F>(x); break;
    }
    fe_store(out + 4 * (size_t)i, r);
}
// the same operations through the native radix-2^29 field (Fq only): in/out in arkworks words
__global__ __launch_bounds__(256) void k_test_field29(int op, const uint64_t *a, const uint64_t *b, uint32_t n, uint64_t *out) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    Fq<2> x = fq_from_words(fe_load(a + 4 * (size_t)i));
    Fq<2> y = b ? fq_from_words(fe_load(b + 4 * (size_t)i)) : fq_zero<2>();
    Fe r;
    switch (op) {
        case 0: r = fq_to_words(fq_mul(x, y)); break;
        case 1: r = fq_to_words(fq_add(x, y)); break;
        case 2: r = fq_to_words(fq_sub<2>(x, y)); break;
        case 3: r = fq_is_zero_modp(x) ? fe_zero() : fq_to_words(fq_inv(x)); break;
        case 6: r = fq_to_words(fq_sqr(x)); break;
        case 7: r = fq_to_words(fq_muls<4>(fq_muls<3>(fq_add(x, y)))); break;  // 12 (x + y), lazy chain
        case 8: r = fq_to_words(fq_tighten(fq_sub_sub2(fq_muls<4>(x), y, x))); break;  // 2x - y
        default: r = fq_to_words(x); break;                                            // round trip
    }
    fe_store(out + 4 * (size_t)i, r);
}
HALO_DEV bool aff_same(const AffN &a, const AffN &b) {
    if (aff_is_inf(a) || aff_is_inf(b)) return aff_is_inf(a) && aff_is_inf(b);
    return fq_eq_modp(a.x, b.x) && fq_eq_modp(a.y, b.y);
}
__global__ __launch_bounds__(256) void k_test_point(int op, const uint64_t *a, const uint64_t *b, uint32_t n, uint64_t *out) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    JacN p = jac_from_words(a + 12 * (size_t)i);
    uint64_t *o = out + 12 * (size_t)i;
    if (op == 0) {
        XyzzN x = jac_to_xyzz(p);
        xyzz_add(x, jac_to_xyzz(jac_from_words(b + 12 * (size_t)i)));
        xyzz_store_jac_words(o, x);
    } else if (op == 1) {
        AffN q = aff_from_words(b + 8 * (size_t)i);
        XyzzN x = jac_to_xyzz(p);
        xyzz_madd(x, q);
        JacN r2 = jac_madd(p, q);  // both mixed-add forms must agree; disagreement poisons the output
        JacN x1; x1.x = fq_widen<8>(fq_mul(x.x, fq_sqr(x.zz))); x1.y = fq_widen<8>(fq_mul(x.y, fq_sqr(x.zzz))); x1.z = fq_widen<4>(x.zzz);
        if (xyzz_is_inf(x)) x1 = jac_inf();
        if (aff_same(jac_to_aff(x1), jac_to_aff(r2))) jac_store_words(o, r2);
        else { AffN bad; bad.x = fq_widen<2>(fq_one()); bad.y = bad.x; jac_store_words(o, jac_from_aff(bad)); }
    } else if (op == 2) {
        JacN r1 = jac_dbl(p);
        XyzzN x = xyzz_dbl(jac_to_xyzz(p));
        JacN x1; x1.x = fq_widen<8>(fq_mul(x.x, fq_sqr(x.zz))); x1.y = fq_widen<8>(fq_mul(x.y, fq_sqr(x.zzz))); x1.z = fq_widen<4>(x.zzz);
        if (xyzz_is_inf(x)) x1 = jac_inf();
        if (aff_same(jac_to_aff(r1), jac_to_aff(x1))) jac_store_words(o, r1);
        else { AffN bad; bad.x = fq_widen<2>(fq_one()); bad.y = bad.x; jac_store_words(o, jac_from_aff(bad)); }
    } else {
        // p * scalar (Montgomery Fr), MSB-first double-and-add on the affine form of p
        Fe k = fe_from_mont<FrCfg>(fe_load(b + 4 * (size_t)i));
        AffN pa = jac_to_aff(p);
        JacN acc = jac_inf();
#pragma unroll 1
        for (int limb = 7; limb >= 0; limb--) {
            uint32_t word = 0;
#pragma unroll
            for (int q = 0; q < 8; q++) word = (q == limb) ? k.v[q] : word;
#pragma unroll 1
            for (int bit = 31; bit >= 0; bit--) {
                acc = jac_dbl(acc);
                if ((word >> bit) & 1u) acc = jac_madd(acc, pa);
            }
        }
        jac_store_words(o, acc);
    }
}

// the quad-parallel forms of curve_quad.hpp, one point per 4 lanes: op 4 = a + b (XYZZ add), op 5 = 2a, op 6 = a + b where
// every fourth pair is replaced by (a, a) so that general additions and doublings share a wave
__global__ __launch_bounds__(256) void k_test_point_quad(int op, const uint64_t *a, const uint64_t *b, uint32_t n, uint64_t *out) {
    uint32_t t = blockIdx.x * 256 + threadIdx.x;
    uint32_t i = t >> 2;
    int ql = (int)(t & 3);
    bool live = i < n;
    if (!live) i = n - 1;  // keep every lane of the wave busy: the quad forms need whole quads
    XyzzN x = jac_to_xyzz(jac_from_words(a + 12 * (size_t)i));
    XyzzN y = jac_to_xyzz(jac_from_words((op == 5 || (op == 6 && (i & 3) == 3) ? a : b) + 12 * (size_t)i));
    if (op == 5) x = xyzz_dbl_quad(x, ql);
    else xyzz_add_quad(x, y, ql);
    if (live && ql == 0) xyzz_store_jac_words(out + 12 * (size_t)i, x);
}

int test_field_op(halo_ctx *ctx, int field, int op, const uint64_t *d_a, const uint64_t *d_b, size_t n, uint64_t *d_out) {
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (field == 2) HALO_LAUNCH(ctx, "k_test_field29", k_test_field29, grid, block, 0, op, d_a, d_b, (uint32_t)n, d_out);
    else if (field == 0) HALO_LAUNCH(ctx, "k_test_field", k_test_field<FqCfg>, grid, block, 0, op, d_a, d_b, (uint32_t)n, d_out);
    else HALO_LAUNCH(ctx, "k_test_field", k_test_field<FrCfg>, grid, block, 0, op, d_a, d_b, (uint32_t)n, d_out);
    HALO_HIP(hipGetLastError());
    return HALO_OK;
}
int test_point_op(halo_ctx *ctx, int op, const uint64_t *d_a, const uint64_t *d_b, size_t n, uint64_t *d_out) {
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (op >= 4) {
        HALO_LAUNCH(ctx, "k_test_point_quad", k_test_point_quad, dim3((unsigned)((4 * n + 255) / 256)), block, 0, op, d_a, d_b, (uint32_t)n, d_out);
        HALO_HIP(hipGetLastError());
        return HALO_OK;
    }
    HALO_LAUNCH(ctx, "k_test_point", k_test_point, grid, block, 0, op, d_a, d_b, (uint32_t)n, d_out);
    HALO_HIP(hipGetLastError());
    return HALO_OK;
}

}  // namespace halo

using namespace halo;

extern "C" {

// what tuning() read from the environment (csrc/tuning.hip), by field name: lets a CPU test pin the parsing
long halo_dev_tuning(const char *name) {
    if (!name) return -1;
    const Tuning &t = tuning();
    if (!std::strcmp(name, "host_split_set")) return t.host_split_set ? 1 : 0;
    if (!std::strcmp(name, "host_pieces")) return t.host_pieces;
    if (!std::strncmp(name, "host_split", 10) && name[10] >= '0' && name[10] <= '3' && !name[11]) return t.host_split[name[10] - '0'];
    if (!std::strcmp(name, "fold_table_after")) return t.fold_table_after;
    if (!std::strcmp(name, "graph_cache")) return t.graph_cache;
    if (!std::strcmp(name, "pow_e")) return t.pow_e;
    if (!std::strcmp(name, "spin_us")) return t.spin_us;
    if (!std::strcmp(name, "graphs")) return t.graphs;
    if (!std::strcmp(name, "memory_budget")) return t.memory_budget_set ? (long)(t.memory_budget >> 20) : -1;  // MiB
    if (!std::strcmp(name, "trace")) return t.trace ? 1 : 0;
    if (!std::strcmp(name, "tagged")) return t.tagged ? 1 : 0;
    return -1;
}

int halo_dev_hook(const char *name, long value) {
    if (!name) { set_error("dev_hook: null name"); return HALO_E_ARG; }
    DevHooks &h = dev_hooks();
    if (!std::strcmp(name, "table_fail")) h.table_fail = (int)value;
    else if (!std::strcmp(name, "force_peer_copy")) h.force_peer_copy = (int)value;
    else if (!std::strcmp(name, "shard_fail_rank")) h.shard_fail_rank = (int)value;
    else if (!std::strcmp(name, "shard_fail_at")) h.shard_fail_at = (int)value;
    else if (!std::strcmp(name, "reset")) h = DevHooks();
    else { set_error("dev_hook: unknown hook (table_fail, force_peer_copy, shard_fail_rank, shard_fail_at, reset)"); return HALO_E_ARG; }
    return HALO_OK;
}

int halo_bench_fr_kernel(halo_ctx *ctx, int which, size_t n, int reps) {
    HALO_CTX(ctx);
    return bench_fr_kernel(ctx, which, n, reps);
}

int halo_test_glv_digits(const uint64_t xi[4], uint8_t out[144], int *n_out) {
    if (!xi || !out || !n_out) { set_error("glv_digits: null pointer"); return HALO_E_ARG; }
    host::GlvDigits dg = host::glv_digits(host::Fr::load(xi));
    for (int i = 0; i < 144; ++i) out[i] = i < dg.n ? dg.d[i] : 0;
    *n_out = dg.n;
    return HALO_OK;
}

int halo_test_fold_digits(const uint64_t s[4], int8_t out[44]) {
    if (!s || !out) { set_error("fold_digits: null pointer"); return HALO_E_ARG; }
    fold_digits_host(host::Fr::load(s), out);
    return HALO_OK;
}

int halo_set_graphs(halo_ctx *ctx, int on) {
    if (!ctx) { set_error("null context"); return HALO_E_ARG; }
    ctx->use_graphs = on != 0;
    for (halo_ctx *sh : ctx->shards) (void)halo_set_graphs(sh, on);  // a multi-device context: its shards run the MSMs
    return HALO_OK;
}

int halo_set_ipa_switch(halo_ctx *ctx, size_t size) {
    if (!ctx) { set_error("null context"); return HALO_E_ARG; }
    ctx->nofold_size = size;
    return HALO_OK;
}

int halo_set_window_bits(halo_ctx *ctx, int c) {
    if (!ctx || (c != 0 && (c < 4 || c > 16))) { set_error("window bits must be 0 or in [4, 16]"); return HALO_E_ARG; }
    ctx->window_bits = c;
    for (halo_ctx *sh : ctx->shards) (void)halo_set_window_bits(sh, c);  // a multi-device context: its shards run the MSMs
    return HALO_OK;
}

int halo_set_reduce_span(halo_ctx *ctx, int span) {
    if (!ctx || span < 0 || span > 512 || (span & (span - 1))) { set_error("reduce span must be 0 or a power of two <= 512"); return HALO_E_ARG; }
    ctx->reduce_span = span;
    for (halo_ctx *sh : ctx->shards) (void)halo_set_reduce_span(sh, span);  // a multi-device context: its shards run the MSMs
    return HALO_OK;
}

int halo_set_sort_mode(halo_ctx *ctx, int mode) {
    if (!ctx || mode < -1 || mode > 1) { set_error("sort mode must be -1 (automatic), 0 (one level) or 1 (two levels)"); return HALO_E_ARG; }
    ctx->sort_two_level = mode;
    for (halo_ctx *sh : ctx->shards) (void)halo_set_sort_mode(sh, mode);  // a multi-device context: its shards run the MSMs
    return HALO_OK;
}

int halo_set_small_path(halo_ctx *ctx, int mode) {
    if (!ctx || mode < -1 || mode > 0) { set_error("small path mode must be -1 (automatic) or 0 (never)"); return HALO_E_ARG; }
    ctx->small_path = mode;
    for (halo_ctx *sh : ctx->shards) (void)halo_set_small_path(sh, mode);  // a multi-device context: its shards run the MSMs
    return HALO_OK;
}

int halo_set_batch_verify(halo_ctx *ctx, int on) {
    if (!ctx) { set_error("null context"); return HALO_E_ARG; }
    ctx->batch_verify = on != 0;
    return HALO_OK;
}

int halo_set_fold_levels(halo_ctx *ctx, int levels) {
    if (!ctx || (levels != 1 && levels != 2)) { set_error("fold levels must be 1 or 2"); return HALO_E_ARG; }
    ctx->fold_levels = levels;
    return HALO_OK;
}

int halo_set_fold_async(halo_ctx *ctx, int mode) {
    if (!ctx || mode < -1 || mode > 1) { set_error("fold async mode must be -1 (automatic), 0 (never) or 1 (wherever possible)"); return HALO_E_ARG; }
    ctx->fold_async = mode;
    return HALO_OK;
}

int halo_set_task_len(halo_ctx *ctx, int len) {
    if (!ctx || !(len == 0 || len == 8 || len == 16 || len == 32 || len == 64)) { set_error("task length must be 0, 8, 16, 32 or 64"); return HALO_E_ARG; }
    ctx->task_len = len;
    for (halo_ctx *sh : ctx->shards) (void)halo_set_task_len(sh, len);  // a multi-device context: its shards run the MSMs
    return HALO_OK;
}

int halo_test_field_op(halo_ctx *ctx, int field, int op, const uint64_t *a, const uint64_t *b, size_t n, uint64_t *out) {
    HALO_CTX(ctx);
    if (n > (ctx->n < 64 ? 64 : ctx->n)) { set_error("test_field_op: n exceeds context size"); return HALO_E_ARG; }
    int rc = upload_words(ctx, ctx->d_tmp_a, a, n * 4);
    if (!rc && b) rc = upload_words(ctx, ctx->d_tmp_b, b, n * 4);
    if (rc) return rc;
    rc = test_field_op(ctx, field, op, ctx->d_tmp_a, b ? ctx->d_tmp_b : nullptr, n, ctx->d_tmp_a + 4 * n);
    if (rc) return rc;
    return download_words(ctx, out, ctx->d_tmp_a + 4 * n, n * 4);
}

int halo_test_point_op(halo_ctx *ctx, int op, const uint64_t *a_jac, const uint64_t *b, size_t n, uint64_t *out_jac) {
    HALO_CTX(ctx);
    if (n > (ctx->n < 64 ? 64 : ctx->n) / 2) { set_error("test_point_op: n exceeds half the context size"); return HALO_E_ARG; }
    size_t bw = (op == 0 || op == 4 || op == 6) ? 12 : (op == 1 ? 8 : 4);
    int rc = upload_words(ctx, ctx->d_tmp_a, a_jac, n * 12);
    if (!rc && b && op != 2 && op != 5) rc = upload_words(ctx, ctx->d_tmp_b, b, n * bw);
    if (rc) return rc;
    // output goes to the upper half of d_tmp_a? keep it simple: a dedicated allocation
    uint64_t *d_out = nullptr;
    HALO_HIP(hipMalloc(&d_out, n * 96));
    rc = test_point_op(ctx, op, ctx->d_tmp_a, ctx->d_tmp_b, n, d_out);
    if (!rc) rc = download_words(ctx, out_jac, d_out, n * 12);
    (void)hipFree(d_out);
    if (rc) return rc;
    for (size_t i = 0; i < n; ++i) host::Point::load(out_jac + 12 * i).store_normalized(out_jac + 12 * i);
    return HALO_OK;
}

}  // extern "C"
