// extern "C" surface of libhalo_hip.so (include/halo_accumulation.h): context, error
// reporting, profiling hooks and the group.rs / pcdl.rs-level entry points.  All compute goes
// to the HIP kernels in msm.hip / ipa.hip; there is no CPU fallback.
#include <cstdio>
#include <cstring>
#include <new>
#include <utility>

#include "internal.hpp"

namespace halo {

static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }
int hip_fail(hipError_t e, const char *what) {
    g_err = std::string("HIP error: ") + hipGetErrorString(e) + " in " + what;
    return HALO_E_DEVICE;
}

int Profiler::find(const char *name) {
    for (size_t i = 0; i < entries.size(); ++i)
        if (std::strcmp(entries[i].name, name) == 0) return (int)i;
    ProfEntry e;
    e.name = name;
    entries.push_back(e);
    return (int)entries.size() - 1;
}
void Profiler::begin(const char *name, hipStream_t s) {
    Pending p;
    p.idx = find(name);
    auto get = [&]() {
        hipEvent_t e;
        if (!pool.empty()) { e = pool.back(); pool.pop_back(); }
        else (void)hipEventCreate(&e);
        return e;
    };
    p.a = get();
    p.b = get();
    (void)hipEventRecord(p.a, s);
    pending.push_back(p);
}
void Profiler::end(hipStream_t s) { (void)hipEventRecord(pending.back().b, s); }
void Profiler::collect() {
    for (auto &p : pending) {
        float ms = 0;
        if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            entries[p.idx].total_ms += ms;
            entries[p.idx].launches += 1;
        }
        pool.push_back(p.a);
        pool.push_back(p.b);
    }
    pending.clear();
}

static int ctx_alloc_common(halo_ctx *ctx, int device, size_t n) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        set_error("no HIP device available: this library has no CPU fallback");
        return HALO_E_DEVICE;
    }
    if (device < 0 || device >= count) { set_error("device index out of range"); return HALO_E_ARG; }
    HALO_HIP(hipSetDevice(device));
    ctx->device = device;
    ctx->n = n;
    alloc_epoch_bump(ctx);
    for (int k = 0; k < HALO_SLOTS; ++k) HALO_HIP(hipStreamCreateWithFlags(&ctx->streams[k], hipStreamNonBlocking));
    ctx->stream = ctx->streams[0];
    if (!ctx->share) {  // (a clone arrives with the key it shares: halo_ctx_clone)
        HALO_HIP(hipMalloc(&ctx->d_bases, (n ? n : 1) * 128));
        ctx->share = std::make_shared<KeyShare>();
        ctx->share->d_bases = ctx->d_bases;
    }
    size_t tn = n < 64 ? 64 : n;
    ctx->tmp_words = tn * 12;
    HALO_HIP(hipMalloc(&ctx->d_tmp_a, tn * 12 * 8));
    HALO_HIP(hipMalloc(&ctx->d_tmp_b, tn * 16 * 8));
    HALO_HIP(hipMalloc(&ctx->d_tmp_c, 16384 * 8));
    HALO_HIP(hipHostMalloc(&ctx->h_pinned, 4096));
    HALO_HIP(hipHostMalloc(&ctx->h_wintab, 8192));
    if (tuning().graphs >= 0) ctx->use_graphs = tuning().graphs != 0;  // HALO_GRAPHS=0: never replay launch graphs
    if (tuning().fold_async > -2) ctx->fold_async = tuning().fold_async;  // -1 automatic, 0 every fold in line, 1 beside the rounds wherever possible (halo_set_fold_async)
    host::g_inv_fermat = tuning().host_inv_fermat;
    return msm_workspace_alloc(ctx, n, 0);
}

static bool is_pow2(size_t n) { return n && !(n & (n - 1)); }

// guard used by every entry point

int upload_words(halo_ctx *ctx, uint64_t *dst, const uint64_t *src, size_t words) {
    if (words == 0) return HALO_OK;
    HALO_HIP(hipMemcpyAsync(dst, src, words * 8, hipMemcpyHostToDevice, ctx->stream));
    HALO_HIP(hipStreamSynchronize(ctx->stream));
    return HALO_OK;
}
int download_words(halo_ctx *ctx, uint64_t *dst, const uint64_t *src, size_t words) {
    if (words == 0) return HALO_OK;
    HALO_HIP(hipMemcpyAsync(dst, src, words * 8, hipMemcpyDeviceToHost, ctx->stream));
    HALO_HIP(hipStreamSynchronize(ctx->stream));
    return HALO_OK;
}


// Key size at which the IPA stops folding G and switches to MSMs over the fixed key (ipa.hip).
#define kNoFoldSize (ctx->nofold_size)
static int ipa_enter_nofold(halo_ipa *st) {
    halo_ctx *ctx = st->ctx;
    st->nofold = true;
    st->M = st->m;
    st->s_len = 1;
    uint64_t *one = ctx->h_pinned + 200;
    host::Fr::one().store(one);
    HALO_HIP(hipMemcpyAsync(st->d_s, one, 32, hipMemcpyHostToDevice, ctx->stream));  // the pinned word always holds 1: no wait needed
    return HALO_OK;
}

static int ipa_begin_general(halo_ctx *ctx, size_t n, const uint64_t *d_coeffs_padded, const host::Fr &z_base, const host::Fr *z_scale,
                             const uint64_t *d_z_vec, halo_ipa **out);
int ipa_begin_dev(halo_ctx *ctx, size_t n, const uint64_t *d_coeffs_padded, const host::Fr &z, halo_ipa **out) {
    return ipa_begin_general(ctx, n, d_coeffs_padded, z, nullptr, nullptr, out);
}
// z vector: d_z_vec if given, else z_base^j (times *z_scale if given)
static void ipa_bufs_free(IpaBuffers &b) {
    (void)hipFree(b.d_G); (void)hipFree(b.d_c); (void)hipFree(b.d_z);
    (void)hipFree(b.d_s); (void)hipFree(b.d_s2); (void)hipFree(b.d_FL); (void)hipFree(b.d_FR);
    (void)hipFree(b.d_pbar);
    b = IpaBuffers();
}
// n x (128 + 32 + 32) bytes of state and four cap_M-element vectors for the no-fold rounds
static int ipa_bufs_alloc(halo_ctx *ctx, IpaBuffers &b, size_t n, size_t M) {
    alloc_epoch_bump(ctx);
    b.cap_n = n;
    b.cap_M = M;
    bool ok = hipMalloc(&b.d_G, n * 128) == hipSuccess && hipMalloc(&b.d_c, n * 32) == hipSuccess && hipMalloc(&b.d_z, n * 32) == hipSuccess &&
              hipMalloc(&b.d_s, M * 32) == hipSuccess && hipMalloc(&b.d_s2, M * 32) == hipSuccess && hipMalloc(&b.d_FL, M * 32) == hipSuccess &&
              hipMalloc(&b.d_FR, M * 32) == hipSuccess;
    if (debug_trace()) fprintf(stderr, "[halo] ipa buffers ctx=%p n=%zu M=%zu G=[%p,+%zu) c=%p z=%p\n", (void *)ctx, n, M, (void *)b.d_G, n * 128, (void *)b.d_c, (void *)b.d_z);
    if (!ok) { ipa_bufs_free(b); set_error("ipa_begin: device allocation failed"); return HALO_E_DEVICE; }
    return HALO_OK;
}
void ipa_bufs_release(halo_ctx *ctx) {
    if (ctx->ipa_bufs.d_G) alloc_epoch_bump(ctx);
    ipa_bufs_free(ctx->ipa_bufs);
}

static int ipa_begin_general(halo_ctx *ctx, size_t n, const uint64_t *d_coeffs_padded, const host::Fr &z_base, const host::Fr *z_scale,
                             const uint64_t *d_z_vec, halo_ipa **out) {
    halo_ipa *st = new (std::nothrow) halo_ipa();
    if (!st) { set_error("out of host memory"); return HALO_E_ARG; }
    st->ctx = ctx;
    st->n = st->m = n;
    int rc = HALO_OK;
    if (hipEventCreateWithFlags(&st->ev, hipEventDisableTiming) != hipSuccess) { delete st; set_error("ipa_begin: event"); return HALO_E_DEVICE; }
    if (hipEventCreateWithFlags(&st->ev_fold, hipEventDisableTiming) != hipSuccess) { (void)hipEventDestroy(st->ev); delete st; set_error("ipa_begin: event"); return HALO_E_DEVICE; }
    do {
        // The context's own buffers (sized for the whole key on first use) serve one state at a time: the opens of a
        // prover loop allocate nothing.  A second concurrent state of the same context gets private buffers.
        // the no-fold vectors span the key the MSMs run over: the whole key when folds are deferred (two levels at a time)
        bool defer = ctx->fold_levels >= 2 && n > kNoFoldSize;
        size_t M = defer ? n : (n < kNoFoldSize ? n : kNoFoldSize);
        IpaBuffers &cb = ctx->ipa_bufs;
        IpaBuffers priv;
        if (!cb.in_use) {
            size_t want_n = ctx->n > n ? ctx->n : n, want_M = (ctx->fold_levels >= 2 || want_n < kNoFoldSize) ? want_n : kNoFoldSize;
            if (want_M < M) want_M = M;
            if (cb.cap_n < n || cb.cap_M < M) {
                for (int k = 0; k < HALO_SLOTS; ++k) (void)hipStreamSynchronize(ctx->streams[k]);
                ipa_bufs_release(ctx);
                rc = ipa_bufs_alloc(ctx, cb, want_n, want_M);
                if (rc) break;
            }
            cb.in_use = true;
            st->borrowed = true;
            priv = cb;
        } else {
            rc = ipa_bufs_alloc(ctx, priv, n, M);
            if (rc) break;
        }
        st->d_G = priv.d_G; st->d_c = priv.d_c; st->d_z = priv.d_z;
        st->d_s = priv.d_s; st->d_s2 = priv.d_s2; st->d_FL = priv.d_FL; st->d_FR = priv.d_FR;
        st->d_pbar = st->borrowed ? priv.d_pbar : nullptr;
        // The key is only copied when it is going to be folded in place round by round; the no-fold forms read the
        // context's own bases until the first real fold writes d_G (and may then use the context's fixed-base table).
        bool read_only_key = defer || n <= kNoFoldSize;
        st->G_src = read_only_key ? ctx->d_bases : st->d_G;
        if ((!read_only_key && hipMemcpyAsync(st->d_G, ctx->d_bases, n * 128, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) ||
            hipMemcpyAsync(st->d_c, d_coeffs_padded, n * 32, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) {
            set_error("ipa_begin: copy failed"); rc = HALO_E_DEVICE; break;
        }
        if (d_z_vec) {
            if (hipMemcpyAsync(st->d_z, d_z_vec, n * 32, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) {
                set_error("ipa_begin: copy failed"); rc = HALO_E_DEVICE; break;
            }
        } else {
            rc = fr_powers(ctx, z_base, n, st->d_z);
            if (!rc && z_scale) rc = fr_scale(ctx, st->d_z, n, *z_scale);
        }
        if (rc) break;
        if (n <= kNoFoldSize || defer) rc = ipa_enter_nofold(st);
        st->deferred = defer;
        st->s_host.assign(1, host::Fr::one());
    } while (0);
    if (rc) { halo_ipa_destroy(st); return rc; }
    ctx->worker.add_hot(1);
    st->counted_hot = true;
    *out = st;
    return HALO_OK;
}

static int upload(halo_ctx *ctx, uint64_t *dst, const uint64_t *src, size_t words) { return upload_words(ctx, dst, src, words); }
static int download(halo_ctx *ctx, uint64_t *dst, const uint64_t *src, size_t words) { return download_words(ctx, dst, src, words); }

}  // namespace halo

using namespace halo;

extern "C" {

const char *halo_last_error(void) { return g_err.c_str(); }

int halo_device_count(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) return 0;
    return count;
}

int halo_ctx_create(int device, const uint64_t *bases_affine, size_t n, halo_ctx **out) {
    if (!out || (n && !bases_affine)) { set_error("ctx_create: null argument"); return HALO_E_ARG; }
    halo_ctx *ctx = new (std::nothrow) halo_ctx();
    if (!ctx) { set_error("out of host memory"); return HALO_E_ARG; }
    int rc = ctx_alloc_common(ctx, device, n);
    if (rc == HALO_OK) rc = upload(ctx, ctx->d_tmp_a, bases_affine, n * 8);
    if (rc == HALO_OK) rc = aff_words_to_native(ctx, ctx->d_tmp_a, n, ctx->d_bases);
    if (rc == HALO_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = HALO_E_DEVICE;
    if (rc != HALO_OK) { halo_ctx_destroy(ctx); return rc; }
    *out = ctx;
    return HALO_OK;
}

int halo_ctx_create_urs(int device, uint64_t first_index, size_t n, halo_ctx **out) {
    if (!out) { set_error("ctx_create_urs: null argument"); return HALO_E_ARG; }
    halo_ctx *ctx = new (std::nothrow) halo_ctx();
    if (!ctx) { set_error("out of host memory"); return HALO_E_ARG; }
    int rc = ctx_alloc_common(ctx, device, n);
    if (rc == HALO_OK) rc = urs_generate(ctx, first_index, 1, n, ctx->d_bases);
    if (rc != HALO_OK) { halo_ctx_destroy(ctx); return rc; }
    *out = ctx;
    return HALO_OK;
}

int halo_ctx_create_urs_strided(int device, uint64_t first_index, uint64_t stride, size_t n, halo_ctx **out) {
    if (!out || stride == 0) { set_error("ctx_create_urs_strided: bad argument"); return HALO_E_ARG; }
    halo_ctx *ctx = new (std::nothrow) halo_ctx();
    if (!ctx) { set_error("out of host memory"); return HALO_E_ARG; }
    int rc = ctx_alloc_common(ctx, device, n);
    if (rc == HALO_OK) rc = urs_generate(ctx, first_index, stride, n, ctx->d_bases);
    if (rc != HALO_OK) { halo_ctx_destroy(ctx); return rc; }
    *out = ctx;
    return HALO_OK;
}

static int ctx_devices_ok(const int *devices, int n_dev) {
    int count = 0;
    if (!devices || n_dev < 1 || n_dev > 64) { set_error("ctx_create_multi: 1..64 device ids"); return HALO_E_ARG; }
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { set_error("no HIP device available: this library has no CPU fallback"); return HALO_E_DEVICE; }
    for (int k = 0; k < n_dev; ++k)
        if (devices[k] < 0 || devices[k] >= count) { set_error("ctx_create_multi: device index out of range"); return HALO_E_ARG; }
    return HALO_OK;
}
int halo_ctx_create_multi(const int *devices, int n_dev, const uint64_t *bases_affine, size_t n, halo_ctx **out) {
    int rc = ctx_devices_ok(devices, n_dev);
    if (rc) return rc;
    rc = halo_ctx_create(devices[0], bases_affine, n, out);
    if (rc || n_dev == 1) return rc;
    rc = multi_attach_shards(*out, devices, n_dev, bases_affine, 0);
    if (rc) { halo_ctx_destroy(*out); *out = nullptr; }
    return rc;
}
int halo_ctx_create_urs_multi(const int *devices, int n_dev, uint64_t first_index, size_t n, halo_ctx **out) {
    int rc = ctx_devices_ok(devices, n_dev);
    if (rc) return rc;
    rc = halo_ctx_create_urs(devices[0], first_index, n, out);
    if (rc || n_dev == 1) return rc;
    rc = multi_attach_shards(*out, devices, n_dev, nullptr, first_index);
    if (rc) { halo_ctx_destroy(*out); *out = nullptr; }
    return rc;
}
int halo_ctx_devices(const halo_ctx *ctx) { return ctx ? (ctx->shards.empty() ? 1 : (int)ctx->shards.size()) : 0; }

// A second context over the SAME resident key (one per host thread: contexts are independent, calls on one context must not
// overlap).  It shares the key and -- as they come into being, whoever builds them -- the fixed-base MSM table and the fold
// table, and owns its streams, workspaces, scratch and IPA buffers (~2.3 GB at 2^20 points against 37 GB for a context of its
// own with both tables).  Tuning knobs are copied from `ctx` at this moment and independent afterwards.
int halo_ctx_clone(halo_ctx *src, halo_ctx **out) {
    if (!src || !out) { set_error("ctx_clone: null argument"); return HALO_E_ARG; }
    if (!src->shards.empty() || src->parent) { set_error("ctx_clone: multi-device contexts and their shards are not cloned"); return HALO_E_ARG; }
    halo_ctx *ctx = new (std::nothrow) halo_ctx();
    if (!ctx) { set_error("out of host memory"); return HALO_E_ARG; }
    ctx->share = src->share;
    { std::lock_guard<std::mutex> lk(ctx->share->mu); ctx->share->users++; }
    ctx->d_bases = ctx->share->d_bases;
    int rc = ctx_alloc_common(ctx, src->device, src->n);
    if (rc != HALO_OK) { halo_ctx_destroy(ctx); return rc; }
    ctx->window_bits = src->window_bits; ctx->reduce_span = src->reduce_span; ctx->sort_two_level = src->sort_two_level;
    ctx->task_len = src->task_len; ctx->table_mode = src->table_mode; ctx->small_path = src->small_path; ctx->use_graphs = src->use_graphs;
    ctx->nofold_size = src->nofold_size; ctx->batch_verify = src->batch_verify; ctx->fold_table_mode = src->fold_table_mode;
    ctx->fold_levels = src->fold_levels; ctx->fold_async = src->fold_async;
    *out = ctx;
    return HALO_OK;
}

void halo_ctx_destroy(halo_ctx *ctx) {
    if (!ctx) return;
    multi_destroy(ctx);
    (void)hipSetDevice(ctx->device);
    for (auto st : ctx->streams) if (st) (void)hipStreamSynchronize(st);
    ctx->prof.collect();
    ctx->worker.stop();
    for (auto e : ctx->prof.pool) (void)hipEventDestroy(e);
    msm_workspace_free(ctx);
    foldtab_release(ctx);
    ipa_bufs_release(ctx);
    if (ctx->share) {  // the key itself: freed by its last user (the tables went above, the same way)
        bool last;
        { std::lock_guard<std::mutex> lk(ctx->share->mu); last = --ctx->share->users == 0; }
        if (last) {  // (tables a user had given up while others still held them went nowhere: they go now)
            (void)hipFree(ctx->share->d_table);
            (void)hipFree(ctx->share->d_foldtab);
            ctx->share->d_table = ctx->share->d_foldtab = nullptr;
            table_budget_release(ctx, ctx->share->budget_held);
            (void)hipFree(ctx->share->d_bases);
        }
    } else {
        (void)hipFree(ctx->d_bases);
    }
    (void)hipFree(ctx->d_tmp_a);
    (void)hipFree(ctx->d_tmp_b);
    (void)hipFree(ctx->d_tmp_c);
    (void)hipFree(ctx->d_poly);
    (void)hipFree(ctx->d_poly2);
    (void)hipFree(ctx->d_verify);
    for (auto p : ctx->d_slot_scalars) (void)hipFree(p);
    for (auto p : ctx->d_batch_scalars) (void)hipFree(p);
    if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
    if (ctx->h_wintab) (void)hipHostFree(ctx->h_wintab);
    for (auto &pair : ctx->ev_piece) for (auto e : pair) if (e) (void)hipEventDestroy(e);
    for (auto st : ctx->streams) if (st) (void)hipStreamDestroy(st);
    delete ctx;
}

size_t halo_ctx_size(const halo_ctx *ctx) { return ctx ? ctx->n : 0; }
void *halo_ctx_bases_dev(halo_ctx *ctx) { return ctx ? ctx->d_bases : nullptr; }
void *halo_ctx_stream(halo_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int halo_ctx_read_bases(halo_ctx *ctx, size_t off, size_t n, uint64_t *out) {
    HALO_CTX(ctx);
    if (off + n > ctx->n || !out) { set_error("read_bases: range"); return HALO_E_ARG; }
    int rc = aff_native_to_words(ctx, ctx->d_bases + 32 * off, n, ctx->d_tmp_a);
    if (rc) return rc;
    return download(ctx, out, ctx->d_tmp_a, n * 8);
}

int halo_public_points(uint64_t S_out[12], uint64_t H_out[12]) {
    host::Point g = host::Point::generator();
    g.mul(host::urs_scalar(0)).store_normalized(S_out);
    g.mul(host::urs_scalar(1)).store_normalized(H_out);
    return HALO_OK;
}

// ------------------------------------------------------------------ group.rs
int halo_msm_dev(halo_ctx *ctx, size_t off, size_t n, const void *d_scalars, int mont, uint64_t out[12]) {
    HALO_CTX(ctx);
    if (off + n > ctx->n || !out || (n && !d_scalars)) { set_error("msm: bad range or null pointer"); return HALO_E_ARG; }
    host::Point r;
    int rc = msm_run(ctx, ctx->d_bases + 32 * off, static_cast<const uint64_t *>(d_scalars), mont != 0, n, &r);
    if (rc) return rc;
    r.store_normalized(out);
    return HALO_OK;
}

int halo_msm_dev_begin(halo_ctx *ctx, int slot, size_t off, size_t n, const void *d_scalars, int mont) {
    HALO_CTX(ctx);
    if (off + n > ctx->n || (n && !d_scalars)) { set_error("msm: bad range or null pointer"); return HALO_E_ARG; }
    if (!ctx->shards.empty()) return multi_begin(ctx, slot, off, n, nullptr, static_cast<const uint64_t *>(d_scalars), mont != 0);
    return msm_enqueue(ctx, slot, ctx->d_bases + 32 * off, static_cast<const uint64_t *>(d_scalars), mont != 0, n);
}
int halo_msm_dev_begin_part(halo_ctx *ctx, int slot, size_t off, size_t n, const void *d_scalars, int mont, int part, int parts) {
    HALO_CTX(ctx);
    if (off + n > ctx->n || (n && !d_scalars)) { set_error("msm: bad range or null pointer"); return HALO_E_ARG; }
    MsmBatch one;
    one.count = 1;
    one.scalars[0] = static_cast<const uint64_t *>(d_scalars);
    one.part = part;
    one.parts = parts;
    return msm_enqueue_batch(ctx, slot, ctx->d_bases + 32 * off, one, mont != 0, n);
}
int halo_msm_dev_end(halo_ctx *ctx, int slot, uint64_t out[12]) {
    HALO_CTX(ctx);
    if (!out) { set_error("msm: null output"); return HALO_E_ARG; }
    host::Point r;
    bool fanned = !ctx->shards.empty() && slot >= 0 && slot < HALO_SLOTS && ctx->fan[slot].active;
    if (fanned && ctx->fan[slot].batch > 0) { set_error("msm: this slot holds a batch (halo_msm_dev_batch_end collects it)"); return HALO_E_ARG; }
    int rc = fanned ? multi_end(ctx, slot, &r) : msm_finish(ctx, slot, &r);
    if (rc) return rc;
    r.store_normalized(out);
    return HALO_OK;
}

int halo_msm_dev_batch_begin(halo_ctx *ctx, int slot, size_t off, size_t n, const void *const *d_scalars, size_t batch, int mont, int part,
                             int parts) {
    HALO_CTX(ctx);
    if (batch < 1 || batch > (size_t)MSM_MAX_BATCH || !d_scalars) { set_error("msm: batch must be in [1, 8]"); return HALO_E_ARG; }
    if (off + n > ctx->n) { set_error("msm: bad range"); return HALO_E_ARG; }
    MsmBatch members;
    members.count = (int)batch;
    members.part = part;
    members.parts = parts;
    for (size_t b = 0; b < batch; ++b) {
        if (n && !d_scalars[b]) { set_error("msm: null scalar pointer in batch"); return HALO_E_ARG; }
        members.scalars[b] = static_cast<const uint64_t *>(d_scalars[b]);
    }
    if (!ctx->shards.empty() && parts == 1) return multi_batch_begin(ctx, slot, off, n, members, mont != 0);
    return msm_enqueue_batch(ctx, slot, ctx->d_bases + 32 * off, members, mont != 0, n);
}
int halo_msm_dev_batch_end(halo_ctx *ctx, int slot, size_t batch, uint64_t *out) {
    HALO_CTX(ctx);
    if (!out || batch < 1 || batch > (size_t)MSM_MAX_BATCH) { set_error("msm: null output or bad batch"); return HALO_E_ARG; }
    host::Point r[MSM_MAX_BATCH];
    bool fanned = !ctx->shards.empty() && slot >= 0 && slot < HALO_SLOTS && ctx->fan[slot].active && ctx->fan[slot].batch > 0;
    int rc = fanned ? multi_batch_end(ctx, slot, r, (int)batch) : msm_finish_batch(ctx, slot, r, (int)batch);
    if (rc) return rc;
    for (size_t b = 0; b < batch; ++b) r[b].store_normalized(out + 12 * b);
    return HALO_OK;
}

// scalars in host memory: copied on the slot's own stream right in front of the launch sequence (no host round trip in
// between), into a per-slot device buffer so that launches on different slots overlap with each other's copies
} // extern "C"
int halo::msm_host_begin(halo_ctx *ctx, int slot, size_t off, size_t n, const uint64_t *scalars, int mont) {
    if (slot < 0 || slot >= HALO_SLOTS) { set_error("msm: slot out of range"); return HALO_E_ARG; }
    if (off + n > ctx->n || (n && !scalars)) { set_error("msm: bad range or null pointer"); return HALO_E_ARG; }
    if (ctx->wss[slot].in_flight) { set_error("msm: slot already has an MSM in flight"); return HALO_E_ARG; }
    if (!ctx->d_slot_scalars[slot]) {
        alloc_epoch_bump(ctx);
        HALO_HIP(hipMalloc(&ctx->d_slot_scalars[slot], (ctx->n < 64 ? 64 : ctx->n) * 32));
    }
    if (n) HALO_HIP(hipMemcpyAsync(ctx->d_slot_scalars[slot], scalars, n * 32, hipMemcpyHostToDevice, ctx->streams[slot]));
    return msm_enqueue(ctx, slot, ctx->d_bases + 32 * off, ctx->d_slot_scalars[slot], mont != 0, n);
}
extern "C" {
int halo_msm_begin(halo_ctx *ctx, int slot, size_t off, size_t n, const uint64_t *scalars, int mont) {
    HALO_CTX(ctx);
    if (!ctx->shards.empty()) {
        if (off + n > ctx->n || (n && !scalars)) { set_error("msm: bad range or null pointer"); return HALO_E_ARG; }
        return multi_begin(ctx, slot, off, n, scalars, nullptr, mont != 0);
    }
    return msm_host_begin(ctx, slot, off, n, scalars, mont);
}
int halo_msm_end(halo_ctx *ctx, int slot, uint64_t out[12]) { return halo_msm_dev_end(ctx, slot, out); }

} // extern "C"
// halo_msm, the call the reference's shim makes (integration/ffi.rs point_dot_affine: a Vec of scalars in pageable memory, one
// synchronous call).  One copy in front of one launch sequence leaves the GPU idle for the whole copy (32 MiB at PCIe speed:
// 0.65 of 1.95 ms at n = 2^20).  An MSM is a sum over index stretches, so a large one over the key's c = 20 table runs as P
// stretches on P slots: stretch k's scalars are copied on slot k's stream directly in front of its launch sequence, the copies
// queue up behind one another on the copy engine and every stretch's kernels start as soon as ITS scalars are there -- under
// the copies of the stretches behind it.  The stretches' points are added on the host in index order.  Same result, bit for
// bit (points written by the library are normalised).  Not taken: multi-device contexts (each shard copies its own block over
// its own link already), contexts without the c = 20 table (the first MSM of a context builds it), any slot busy.
// The stretches of an n-point MSM, in sixteenths: the caller's HALO_HOST_SPLIT at every size, else what was measured best on one
// box (profiles/r05_host_msm_stretches_sweep.txt): 4 + 12 below 2^21 points (2^20: 1.63 ms against 1.92 for one copy + one launch
// sequence; a short first stretch lets the kernels start early, and more stretches only serialise behind one another's bucket
// kernels), 2 + 4 + 4 + 6 from 2^21 points on, where every stretch is a full-size pipeline pass of its own (2^22: 5.2 against 7.7 ms,
// 2^24: 19.9 against 29.0).
struct HostSplit { int pieces; int sixteenths[HALO_SLOTS]; };
static HostSplit host_split_for(size_t n) {
    const Tuning &t = halo::tuning();
    if (t.host_split_set) return HostSplit{t.host_pieces, {t.host_split[0], t.host_split[1], t.host_split[2], t.host_split[3]}};
    if (n >= ((size_t)1 << 21)) return HostSplit{4, {2, 4, 4, 6}};
    return HostSplit{2, {4, 12, 0, 0}};
}
static int host_pieces_wanted(const halo_ctx *ctx, size_t n) {
    const int P = host_split_for(n).pieces;
    if (P < 2 || !ctx->shards.empty() || !ctx->d_table || ctx->tbl.c != 20 || ctx->table_mode == 0 || ctx->window_bits != 0) return 1;
    if (n < ((size_t)1 << 19) || n % 64 != 0) return 1;
    for (int k = 0; k < P; ++k)
        if (ctx->wss[k].in_flight || ctx->wss[k].lent_from >= 0) return 1;
    return P;
}
// `valid` <= n: the host array holds that many scalars, the MSM's remaining scalars are zero (a polynomial shorter than d + 1)
static int msm_host_pieces(halo_ctx *ctx, int P, size_t off, size_t n, const uint64_t *scalars, size_t valid, int mont, host::Point *out) {
    // stretch k takes host_split[k] sixteenths of the points (n is a multiple of 64: every length a multiple of 4), the last one the rest
    size_t lens[HALO_SLOTS], offs[HALO_SLOTS];
    const HostSplit split = host_split_for(n);
    {
        size_t at = 0;
        for (int k = 0; k < P; ++k) {
            lens[k] = k == P - 1 ? n - at : n / 16 * (size_t)split.sixteenths[k];
            offs[k] = at;
            at += lens[k];
        }
    }
    int rc = HALO_OK, started = 0;
    for (int k = 0; k < P && !rc; ++k) {
        const size_t len = lens[k];
        const size_t have = valid > offs[k] ? (valid - offs[k] < len ? valid - offs[k] : len) : 0;  // scalars of this stretch the caller has
        if (!ctx->d_slot_scalars[k]) {
            alloc_epoch_bump(ctx);
            hipError_t e = hipMalloc(&ctx->d_slot_scalars[k], (ctx->n < 64 ? 64 : ctx->n) * 32);
            if (e != hipSuccess) { rc = hip_fail(e, "hipMalloc"); break; }
        }
        hipError_t e = hipSuccess;
        if (have < len) e = hipMemsetAsync(ctx->d_slot_scalars[k] + 4 * have, 0, (len - have) * 32, ctx->streams[k]);
        if (e == hipSuccess && have) e = hipMemcpyAsync(ctx->d_slot_scalars[k], scalars + 4 * offs[k], have * 32, hipMemcpyHostToDevice, ctx->streams[k]);
        if (e != hipSuccess) { rc = hip_fail(e, "hipMemcpyAsync"); break; }
        MsmBatch one;
        one.count = 1;
        one.scalars[0] = ctx->d_slot_scalars[k];
        one.sub = true;
        rc = msm_enqueue_batch(ctx, k, ctx->d_bases + 32 * (off + offs[k]), one, mont != 0, len);
        if (!rc) started = k + 1;
    }
    host::Point acc = host::Point::infinity();
    for (int k = 0; k < started; ++k) {  // (every stretch that went out is collected, whatever happened to the others)
        host::Point r;
        int rk = msm_finish(ctx, k, &r);
        if (rk && !rc) rc = rk;
        acc = acc + r;
    }
    *out = acc;
    return rc;
}
// One synchronous MSM over GS[off, off + n) with `valid` <= n scalars in pageable host memory (the rest zero): in stretches where
// that pays (above), else one copy in front of one launch sequence on slot 0.  halo_msm and pcdl::commit / pedersen::commit with host
// coefficients (pcdl_acc.hip) both end here.
int halo::msm_host_run(halo_ctx *ctx, size_t off, size_t n, const uint64_t *scalars, size_t valid, int mont, host::Point *out) {
    if (off + n > ctx->n || valid > n || (valid && !scalars)) { set_error("msm: bad range or null pointer"); return HALO_E_ARG; }
    const int P = ctx->shards.empty() ? host_pieces_wanted(ctx, n) : 1;
    if (P > 1) return msm_host_pieces(ctx, P, off, n, scalars, valid, mont, out);
    if (!ctx->shards.empty() && valid == n) return multi_host_run(ctx, off, n, scalars, mont != 0, out);
    if (ctx->wss[0].in_flight) { set_error("msm: slot already has an MSM in flight"); return HALO_E_ARG; }
    if (!ctx->d_slot_scalars[0]) {
        alloc_epoch_bump(ctx);
        HALO_HIP(hipMalloc(&ctx->d_slot_scalars[0], (ctx->n < 64 ? 64 : ctx->n) * 32));
    }
    if (valid < n) HALO_HIP(hipMemsetAsync(ctx->d_slot_scalars[0] + 4 * valid, 0, (n - valid) * 32, ctx->streams[0]));
    if (valid) HALO_HIP(hipMemcpyAsync(ctx->d_slot_scalars[0], scalars, valid * 32, hipMemcpyHostToDevice, ctx->streams[0]));
    if (!ctx->shards.empty()) {  // (a multi-device context with a short polynomial: the padded scalars are device-resident now)
        HALO_HIP(hipStreamSynchronize(ctx->streams[0]));
        return msm_run(ctx, ctx->d_bases + 32 * off, ctx->d_slot_scalars[0], mont != 0, n, out);
    }
    BorrowScope scope(ctx);  // synchronous: a large MSM may alternate its pieces over slot 1's workspace
    int rc = msm_enqueue(ctx, 0, ctx->d_bases + 32 * off, ctx->d_slot_scalars[0], mont != 0, n);
    if (rc) return rc;
    return msm_finish(ctx, 0, out);
}
extern "C" {
int halo_msm(halo_ctx *ctx, size_t off, size_t n, const uint64_t *scalars, int mont, uint64_t out[12]) {
    if (!out) { set_error("msm: null output"); return HALO_E_ARG; }
    HALO_CTX(ctx);
    if (off + n > ctx->n || (n && !scalars)) { set_error("msm: bad range or null pointer"); return HALO_E_ARG; }
    host::Point r;
    int rc = msm_host_run(ctx, off, n, scalars, n, mont, &r);
    if (rc) return rc;
    r.store_normalized(out);
    return HALO_OK;
}

int halo_msm_points(halo_ctx *ctx, const uint64_t *pts_jac, const uint64_t *scalars, size_t m, uint64_t out[12]) {
    HALO_CTX(ctx);
    if (m > ctx->n && m > 64) { set_error("msm_points: m exceeds the context's workspace"); return HALO_E_ARG; }
    if (!out || (m && (!pts_jac || !scalars))) { set_error("msm_points: null pointer"); return HALO_E_ARG; }
    int rc = upload(ctx, ctx->d_tmp_a, pts_jac, m * 12);
    if (rc) return rc;
    rc = batch_to_affine(ctx, ctx->d_tmp_a, m, reinterpret_cast<uint32_t *>(ctx->d_tmp_b));
    if (rc) return rc;
    HALO_HIP(hipStreamSynchronize(ctx->stream));
    rc = upload(ctx, ctx->d_tmp_a, scalars, m * 4);
    if (rc) return rc;
    host::Point r;
    rc = msm_run(ctx, reinterpret_cast<const uint32_t *>(ctx->d_tmp_b), ctx->d_tmp_a, true, m, &r);
    if (rc) return rc;
    r.store_normalized(out);
    return HALO_OK;
}

int halo_msm_affine(halo_ctx *ctx, const uint64_t *bases_affine, const uint64_t *scalars, size_t m, int mont, uint64_t out[12]) {
    HALO_CTX(ctx);
    if (!out || (m && (!bases_affine || !scalars))) { set_error("msm_affine: null pointer"); return HALO_E_ARG; }
    // more generators than the context's workspace holds: consecutive chunks, partial sums added on the host
    const size_t cap = ctx->n < 64 ? 64 : ctx->n;
    host::Point acc = host::Point::infinity();
    for (size_t lo = 0; lo < m || lo == 0; lo += cap) {
        size_t len = m - lo < cap ? m - lo : cap;
        // bases: len x 8 words -> native 128-byte entries in d_tmp_b (16 words of room per element); scalars follow in d_tmp_a
        int rc = upload(ctx, ctx->d_tmp_a, bases_affine + 8 * lo, len * 8);
        if (rc) return rc;
        rc = aff_words_to_native(ctx, ctx->d_tmp_a, len, reinterpret_cast<uint32_t *>(ctx->d_tmp_b));
        if (rc) return rc;
        HALO_HIP(hipStreamSynchronize(ctx->stream));
        rc = upload(ctx, ctx->d_tmp_a, scalars + 4 * lo, len * 4);
        if (rc) return rc;
        host::Point r;
        rc = msm_run(ctx, reinterpret_cast<const uint32_t *>(ctx->d_tmp_b), ctx->d_tmp_a, mont != 0, len, &r);
        if (rc) return rc;
        acc = acc + r;
        if (m == 0) break;
    }
    acc.store_normalized(out);
    return HALO_OK;
}

int halo_scalar_dot(halo_ctx *ctx, const uint64_t *xs, const uint64_t *ys, size_t m, uint64_t out[4]) {
    HALO_CTX(ctx);
    if (m > (ctx->n < 64 ? 64 : ctx->n)) { set_error("scalar_dot: m exceeds context size"); return HALO_E_ARG; }
    int rc = upload(ctx, ctx->d_tmp_a, xs, m * 4);
    if (!rc) rc = upload(ctx, ctx->d_tmp_b, ys, m * 4);
    if (rc) return rc;
    host::Fr r[2];
    rc = fr_dot2(ctx, ctx->d_tmp_a, ctx->d_tmp_b, nullptr, nullptr, m, r);
    if (rc) return rc;
    r[0].store(out);
    return HALO_OK;
}

int halo_powers(halo_ctx *ctx, const uint64_t z[4], size_t n, uint64_t *out) {
    HALO_CTX(ctx);
    if (n > (ctx->n < 64 ? 64 : ctx->n)) { set_error("powers: n exceeds context size"); return HALO_E_ARG; }
    int rc = fr_powers(ctx, host::Fr::load(z), n, ctx->d_tmp_a);
    if (rc) return rc;
    return download(ctx, out, ctx->d_tmp_a, n * 4);
}

int halo_poly_eval(halo_ctx *ctx, const uint64_t *coeffs, size_t len, const uint64_t z[4], uint64_t out[4]) {
    HALO_CTX(ctx);
    if (len > (ctx->n < 64 ? 64 : ctx->n)) { set_error("poly_eval: len exceeds context size"); return HALO_E_ARG; }
    int rc = upload(ctx, ctx->d_tmp_a, coeffs, len * 4);
    if (rc) return rc;
    host::Fr r;
    rc = fr_poly_eval(ctx, ctx->d_tmp_a, len, host::Fr::load(z), &r);
    if (rc) return rc;
    r.store(out);
    return HALO_OK;
}

// ------------------------------------------------------------------ h(X)
static std::vector<host::Fr> load_frs(const uint64_t *p, size_t count) {
    std::vector<host::Fr> v(count);
    for (size_t i = 0; i < count; ++i) v[i] = host::Fr::load(p + 4 * i);
    return v;
}

int halo_h_coeffs(halo_ctx *ctx, const uint64_t *xis, size_t lg_n, uint64_t *out) {
    HALO_CTX(ctx);
    size_t n = (size_t)1 << lg_n;
    if (n > (ctx->n < 64 ? 64 : ctx->n)) { set_error("h_coeffs: 2^lg_n exceeds context size"); return HALO_E_ARG; }
    std::vector<host::Fr> x = load_frs(xis, lg_n + 1);
    int rc = h_coeffs_dev(ctx, x.data(), lg_n, host::Fr::one(), false, ctx->d_tmp_a);
    if (rc) return rc;
    return download(ctx, out, ctx->d_tmp_a, n * 4);
}

int halo_h_commit(halo_ctx *ctx, const uint64_t *xis, size_t lg_n, uint64_t out[12]) {
    HALO_CTX(ctx);
    size_t n = (size_t)1 << lg_n;
    if (n > ctx->n) { set_error("h_commit: 2^lg_n exceeds the commitment key"); return HALO_E_ARG; }
    std::vector<host::Fr> x = load_frs(xis, lg_n + 1);
    int rc = h_coeffs_dev(ctx, x.data(), lg_n, host::Fr::one(), false, ctx->d_tmp_a);
    if (rc) return rc;
    host::Point r;
    rc = msm_run(ctx, ctx->d_bases, ctx->d_tmp_a, true, n, &r);
    if (rc) return rc;
    r.store_normalized(out);
    return HALO_OK;
}

int halo_h_eval_batch(halo_ctx *ctx, const uint64_t *xis, size_t m, size_t lg_n, const uint64_t z[4], uint64_t *out) {
    HALO_CTX(ctx);
    if (m * (lg_n + 1) * 4 > ctx->tmp_words) { set_error("h_eval_batch: batch too large for context"); return HALO_E_ARG; }
    int rc = upload(ctx, ctx->d_tmp_a, xis, m * (lg_n + 1) * 4);
    if (rc) return rc;
    rc = h_eval_batch(ctx, ctx->d_tmp_a, m, lg_n, host::Fr::load(z), ctx->d_tmp_b);
    if (rc) return rc;
    return download(ctx, out, ctx->d_tmp_b, m * 4);
}

int halo_h_accumulate(halo_ctx *ctx, const uint64_t *h0, const uint64_t *xis, const uint64_t *alphas, size_t m, size_t lg_n,
                      uint64_t *out) {
    HALO_CTX(ctx);
    size_t n = (size_t)1 << lg_n;
    if (n > (ctx->n < 64 ? 64 : ctx->n)) { set_error("h_accumulate: 2^lg_n exceeds context size"); return HALO_E_ARG; }
    HALO_HIP(hipMemsetAsync(ctx->d_tmp_a, 0, n * 32, ctx->stream));
    if (h0) {
        int rc = upload(ctx, ctx->d_tmp_a, h0, (n < 2 ? n : 2) * 4);
        if (rc) return rc;
    }
    for (size_t i = 0; i < m; ++i) {
        std::vector<host::Fr> x = load_frs(xis + 4 * i * (lg_n + 1), lg_n + 1);
        int rc = h_coeffs_dev(ctx, x.data(), lg_n, host::Fr::load(alphas + 4 * i), true, ctx->d_tmp_a);
        if (rc) return rc;
    }
    return download(ctx, out, ctx->d_tmp_a, n * 4);
}

// ------------------------------------------------------------------ IPA state (pcdl.rs:183-231)
int halo_ipa_begin(halo_ctx *ctx, size_t n, const uint64_t *coeffs, size_t len, const uint64_t z[4], halo_ipa **out) {
    HALO_CTX(ctx);
    if (!out || !z || (len && !coeffs)) { set_error("ipa_begin: null pointer"); return HALO_E_ARG; }
    if (!is_pow2(n)) { set_error("ipa_begin: n is not a power of two"); return HALO_E_ASSERT; }  // pcdl.rs:130
    if (n > ctx->n) { set_error("ipa_begin: n exceeds the commitment key (d <= D)"); return HALO_E_ASSERT; }  // pcdl.rs:132
    if (len > n) { set_error("ipa_begin: more coefficients than n (p.degree() <= d)"); return HALO_E_ASSERT; }  // pcdl.rs:131
    HALO_HIP(hipMemsetAsync(ctx->d_tmp_a, 0, n * 32, ctx->stream));
    int rc = upload(ctx, ctx->d_tmp_a, coeffs, len * 4);
    if (rc) return rc;
    return ipa_begin_dev(ctx, n, ctx->d_tmp_a, host::Fr::load(z), out);
}

static host::Fr fr_pow_u64(const host::Fr &z, uint64_t e) {
    uint64_t ex[4] = {e, 0, 0, 0};
    return z.pow(ex);
}

int halo_ipa_begin_strided(halo_ctx *ctx, size_t n_local, const uint64_t *coeffs_local, size_t len, const uint64_t z[4], uint64_t stride,
                           uint64_t offset, halo_ipa **out) {
    HALO_CTX(ctx);
    if (!out || !z || (len && !coeffs_local) || stride == 0) { set_error("ipa_begin_strided: bad argument"); return HALO_E_ARG; }
    if (!is_pow2(n_local)) { set_error("ipa_begin: n is not a power of two"); return HALO_E_ASSERT; }
    if (n_local > ctx->n) { set_error("ipa_begin: n exceeds the commitment key (d <= D)"); return HALO_E_ASSERT; }
    if (len > n_local) { set_error("ipa_begin: more coefficients than n (p.degree() <= d)"); return HALO_E_ASSERT; }
    HALO_HIP(hipMemsetAsync(ctx->d_tmp_a, 0, n_local * 32, ctx->stream));
    int rc = upload(ctx, ctx->d_tmp_a, coeffs_local, len * 4);
    if (rc) return rc;
    host::Fr zz = host::Fr::load(z);
    host::Fr base = fr_pow_u64(zz, stride), scale = fr_pow_u64(zz, offset);  // z^(offset + j * stride)
    return ipa_begin_general(ctx, n_local, ctx->d_tmp_a, base, &scale, nullptr, out);
}

int halo_ipa_begin_vectors(halo_ctx *ctx, size_t n, const uint64_t *c_vec, const uint64_t *z_vec, halo_ipa **out) {
    HALO_CTX(ctx);
    if (!out || (n && (!c_vec || !z_vec))) { set_error("ipa_begin_vectors: null pointer"); return HALO_E_ARG; }
    if (!is_pow2(n)) { set_error("ipa_begin: n is not a power of two"); return HALO_E_ASSERT; }
    if (n > ctx->n) { set_error("ipa_begin: n exceeds the commitment key (d <= D)"); return HALO_E_ASSERT; }
    int rc = upload(ctx, ctx->d_tmp_a, c_vec, n * 4);
    if (!rc) rc = upload(ctx, ctx->d_tmp_b, z_vec, n * 4);
    if (rc) return rc;
    return ipa_begin_general(ctx, n, ctx->d_tmp_a, host::Fr::one(), nullptr, ctx->d_tmp_b, out);
}

int halo_ipa_dot_cz(halo_ipa *st, uint64_t out[4]) {
    if (!st || !out) { set_error("null ipa state"); return HALO_E_ARG; }
    halo_ctx *ctx = st->ctx;
    HALO_CTX(ctx);
    host::Fr r[2];
    int rc = fr_dot2(ctx, st->d_c, st->d_z, nullptr, nullptr, st->m, r);
    if (rc) return rc;
    r[0].store(out);
    return HALO_OK;
}

// hterm: if non-null, called as hterm(which, dot) -> dot * H' and added to L (0) / R (1), each on the thread that combines it
static int ipa_round_lr_points(halo_ipa *st, host::Fr dots[2], host::Point *Lp, host::Point *Rp, bool with_hterm = false);

int halo_ipa_round_lr_partial(halo_ipa *st, uint64_t L[12], uint64_t R[12], uint64_t dots_out[8]) {
    if (!st || !L || !R || !dots_out) { set_error("null ipa state"); return HALO_E_ARG; }
    halo_ctx *ctx = st->ctx;
    HALO_CTX(ctx);
    host::Fr dots[2];
    host::Point Lp, Rp;
    int rc = ipa_round_lr_points(st, dots, &Lp, &Rp);
    if (rc) return rc;
    Lp.store_normalized(L);
    Rp.store_normalized(R);
    dots[0].store(dots_out);
    dots[1].store(dots_out + 4);
    return HALO_OK;
}

int halo_ipa_round_lr(halo_ipa *st, const uint64_t H_prime[12], uint64_t L[12], uint64_t R[12]) {
    if (!st) { set_error("null ipa state"); return HALO_E_ARG; }
    halo_ctx *ctx = st->ctx;
    HALO_CTX(ctx);
    host::Fr dots[2];
    host::Point Lp, Rp, Hp = host::Point::load(H_prime);
    if (!st->hp_from_scalar && !st->hp_table.matches(Hp)) st->hp_table = host::FixedBaseTable(Hp);  // H' is fixed for a whole open
    int rc = ipa_round_lr_points(st, dots, &Lp, &Rp, true);  // L, R come back with their H' terms, normalised
    if (rc) return rc;
    Lp.store(L);
    Rp.store(R);
    return HALO_OK;
}

} // extern "C"
namespace halo {
const host::FixedBaseTable &public_h_table() {
    static const host::FixedBaseTable tbl = [] {
        uint64_t S[12], H[12];
        halo_public_points(S, H);
        return host::FixedBaseTable(host::Point::load(H));
    }();
    return tbl;
}
const host::FixedBaseTable &public_s_table() {
    static const host::FixedBaseTable tbl = [] {
        uint64_t S[12], H[12];
        halo_public_points(S, H);
        return host::FixedBaseTable(host::Point::load(S));
    }();
    return tbl;
}
void ipa_set_hprime_scalar(halo_ipa *st, const host::Fr &xi0) {
    st->hp_from_scalar = true;
    st->hp_scalar = xi0;
    (void)public_h_table();
}
}  // namespace halo
extern "C" {


// HALO_IPA_TIMING=1: where the host side of a round goes (printed when the state is destroyed; development aid)
namespace {
struct RoundTiming { double enqueue = 0, dots = 0, hterm = 0, wait = 0, combine = 0, fold = 0; long rounds = 0; };
thread_local RoundTiming g_rt;
const bool g_rt_on = tuning().ipa_timing;
inline double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}
// <c_r, G_l>, <c_l, G_r> and the two dot products of one round, without the H' terms
static int ipa_round_lr_points(halo_ipa *st, host::Fr dots[2], host::Point *Lp_out, host::Point *Rp_out, bool with_hterm) {
    halo_ctx *ctx = st->ctx;
    if (st->m < 2) { set_error("ipa_round_lr: no rounds left"); return HALO_E_ARG; }
    size_t m = st->m / 2;
    int rc;
    bool batched = false;
    host::Point Lp, Rp;
    double t0 = g_rt_on ? now_us() : 0;
    // <c_r, G_l> on slot 0 and <c_l, G_r> on slot 1 run concurrently; the other streams first wait
    // for everything queued on stream 0 (the previous round's folds)
    // (a batched round -- see below -- does not use stream 1, and every API call here is GPU idle time in the late rounds:
    // leaving out its two waits and the second event record shortened an open by 0.3 ms)
    // Over the full key with its c = 20 table in place L and R are ONE launch sequence too, of another kind: their non-zero scalars
    // sit on disjoint points, so one array with a set bit per element feeds two bucket sets (MsmBatch::tagged) -- one recode, one
    // sort, one bucket kernel of 13 n additions instead of two of 13 n / 2, one window-sum pass of 2048 waves instead of two of 1024.
    const bool tagged = st->nofold && st->M >= ((size_t)1 << 20) && msm_tagged_ready(ctx, st->G_src, st->M);
    const bool will_batch = (st->nofold && st->M < ((size_t)1 << 20)) || tagged;
    HALO_HIP(hipEventRecord(st->ev, ctx->streams[0]));
    if (!will_batch) HALO_HIP(hipStreamWaitEvent(ctx->streams[1], st->ev, 0));
    HALO_HIP(hipStreamWaitEvent(ctx->streams[2], st->ev, 0));
    // dot_l = <c_r, z_l>, dot_r = <c_l, z_r>   (pcdl.rs:203,207) on a third stream.  In the rounds with large MSMs they are
    // issued BEFORE the MSMs' launches: queued behind them the two small kernels waited for the bucket kernel's waves to
    // drain, and the H' terms, which need only the dot products, were computed after the MSMs instead of under them.  In
    // the late rounds (an MSM of a few 10^4 additions: the GPU is mostly idle) the MSM's launches go first instead: the
    // three API calls of the dot products would only delay its start.
    const int dots_env = tuning().dots_first;  // development switch
    const bool dots_first = dots_env >= 0 ? dots_env != 0 : (st->nofold ? st->M : m) > ((size_t)1 << 16);
    auto launch_dots = [&]() -> int {
        hipStream_t saved = ctx->stream;
        ctx->stream = ctx->streams[2];
        int rcl = fr_dot2_launch(ctx, st->d_c + 4 * m, st->d_z, st->d_c, st->d_z + 4 * m, m);
        ctx->stream = saved;
        return rcl;
    };
    if (dots_first) { int rcl = launch_dots(); if (rcl) return rcl; }
    // the last round of a no-fold phase: its two coefficients travel to the host with the round's results (halo_ipa_finish
    // gets U from this round's own MSMs; pinned words 208..215, read after the wait for stream 0 below)
    const bool last_round = st->nofold && st->M > 1 && st->m == 2;
    st->last_valid = false;
    if (last_round) HALO_HIP(hipMemcpyAsync(ctx->h_pinned + 208, st->d_c, 64, hipMemcpyDeviceToHost, ctx->streams[0]));
    if (st->nofold) {
        rc = tagged ? nofold_expand_tagged(ctx, st->d_c, st->d_s, st->m, st->M, st->d_FL) : nofold_expand(ctx, st->d_c, st->d_s, st->m, st->M, st->d_FL, st->d_FR);
        if (rc) return rc;
        if (!will_batch) {
            HALO_HIP(hipEventRecord(st->ev, ctx->streams[0]));
            HALO_HIP(hipStreamWaitEvent(ctx->streams[1], st->ev, 0));
        }
        // Below the fixed-base table's size both MSMs go out as ONE batched launch (same points, two scalar arrays): two
        // launch sequences on two streams did not overlap -- the second one's 1024-thread sort blocks cannot start on a CU
        // that still holds waves of the first one's bucket kernel -- so a round took two MSM latencies instead of one.
        batched = will_batch;
        if (tagged) {
            MsmBatch both;
            both.count = 1;
            both.scalars[0] = st->d_FL;
            both.tagged = true;
            rc = msm_enqueue_batch(ctx, 0, st->G_src, both, false, st->M);
            if (rc) return rc;
        } else if (batched) {
            MsmBatch both;
            both.count = 2;
            both.scalars[0] = st->d_FL;
            both.scalars[1] = st->d_FR;
            // half of each scalar array is zero: over a 2^16-point key 23 windows of 11 bits beat the table's 20 of 13 (measured,
            // medians of 13 opens on one box: 15.40 / 15.42 -> 15.23 / 15.37 ms; 10, 12 and 14 bits are worse, other key sizes keep the table)
            const bool hint_env = tuning().ipa_c_hint;  // development switch
            if (hint_env && st->M == ((size_t)1 << 16)) both.c_hint = 11;
            rc = msm_enqueue_batch(ctx, 0, st->G_src, both, true, st->M);
            if (rc) return rc;
        } else {
            rc = msm_enqueue(ctx, 0, st->G_src, st->d_FL, true, st->M);
            if (rc) return rc;
            rc = msm_enqueue(ctx, 1, st->G_src, st->d_FR, true, st->M);
        }
    } else {
        rc = msm_enqueue(ctx, 0, st->d_G, st->d_c + 4 * m, true, m);
        if (rc) return rc;
        rc = msm_enqueue(ctx, 1, st->d_G + 32 * m, st->d_c, true, m);
    }
    if (rc) { host::Point dummy; (void)msm_finish(ctx, 0, &dummy); return rc; }
    if (!dots_first) { int rcl = launch_dots(); if (rcl) { host::Point dummy; (void)msm_finish(ctx, 0, &dummy); return rcl; } }
    double t1 = g_rt_on ? now_us() : 0;
    int rcd = fr_dot2_collect(ctx, ctx->streams[2], m, dots);
    // Window combine (~250 doublings), the H' term and the normalisation of L start on the helper thread as soon as L's
    // launches are done, while this thread still waits for R's and then does R's: pure host arithmetic on both sides.
    // the H' terms only need the dot products: computed now, while the MSMs are still running
    double t2 = g_rt_on ? now_us() : 0;
    host::Point hterm[2] = {host::Point::infinity(), host::Point::infinity()};
    if (with_hterm && !rcd)
        for (int k = 0; k < 2; ++k) hterm[k] = st->hp_from_scalar ? public_h_table().mul(dots[k] * st->hp_scalar) : st->hp_table.mul(dots[k]);
    auto finish_one = [ctx, st, last_round, with_hterm, &hterm, batched](int which, host::Point *out) {
        host::Point p;
        if (batched) msm_combine_member(ctx, 0, which, &p);
        else msm_combine(ctx, which, &p, 1);
        if (last_round) (which == 0 ? st->last_L : st->last_R) = p;  // before the H' term
        if (with_hterm) p = (p + hterm[which]).normalized();
        *out = p;
    };
    double t3 = g_rt_on ? now_us() : 0;
    rc = msm_wait(ctx, 0, batched ? 2 : 1);
    double t4 = g_rt_on ? now_us() : 0;
    bool l_started = !rc && !rcd;
    if (l_started) ctx->worker.submit([&finish_one, &Lp] { finish_one(0, &Lp); });
    int rc2 = batched ? HALO_OK : msm_wait(ctx, 1, 1);
    if (!rc && !rc2 && !rcd) finish_one(1, &Rp);
    if (l_started) ctx->worker.wait();
    if (g_rt_on) {
        double t5 = now_us();
        g_rt.enqueue += t1 - t0; g_rt.dots += t2 - t1; g_rt.hterm += t3 - t2; g_rt.wait += t4 - t3; g_rt.combine += t5 - t4; g_rt.rounds++;
    }
    if (rc || rc2 || rcd) return rc ? rc : (rc2 ? rc2 : rcd);
    if (last_round) {
        st->last_c0 = host::Fr::load(ctx->h_pinned + 208);
        st->last_c1 = host::Fr::load(ctx->h_pinned + 212);
        st->last_valid = true;
        st->last_folded = false;
    }
    *Lp_out = Lp;
    *Rp_out = Rp;
    return HALO_OK;
}

static int ipa_round_fold_impl(halo_ipa *st, const uint64_t xi[4], const uint64_t xi_inv[4]);
int halo_ipa_round_fold(halo_ipa *st, const uint64_t xi[4], const uint64_t xi_inv[4]) {
    double t0 = g_rt_on ? now_us() : 0;
    int rc = ipa_round_fold_impl(st, xi, xi_inv);
    if (g_rt_on) g_rt.fold += now_us() - t0;
    return rc;
}
static int ipa_round_fold_impl(halo_ipa *st, const uint64_t xi[4], const uint64_t xi_inv[4]) {
    if (!st) { set_error("null ipa state"); return HALO_E_ARG; }
    halo_ctx *ctx = st->ctx;
    HALO_CTX(ctx);
    if (st->m < 2) { set_error("ipa_round_fold: no rounds left"); return HALO_E_ARG; }
    size_t m = st->m / 2;
    host::Fr x = host::Fr::load(xi), xinv = host::Fr::load(xi_inv);
    int rc;
    if (st->last_valid && st->m == 2) {  // the fold after the last round: halo_ipa_finish needs its challenge, see there
        st->last_xi = x;
        st->last_xi_inv = xinv;
        st->last_folded = true;
    } else {
        st->last_valid = false;
    }
    if (st->nofold) {
        rc = nofold_s_update(ctx, st->d_s, st->s_len, x, st->d_s2);
        if (!rc) { std::swap(st->d_s, st->d_s2); st->s_len *= 2; }
        if (!rc && st->deferred) {  // host copy of the (short) challenge products: s'[2t + u] = s[t] xi^u
            std::vector<host::Fr> &cur = st->fold_pending ? st->tail_host : st->s_host;  // (a fold under way has taken s_host)
            std::vector<host::Fr> s2(2 * cur.size());
            for (size_t t = 0; t < cur.size(); ++t) { s2[2 * t] = cur[t]; s2[2 * t + 1] = cur[t] * x; }
            cur.swap(s2);
        }
    } else {
        rc = ipa_fold_points(ctx, st->d_G, m, x);
    }
    if (!rc) rc = ipa_fold_scalars(ctx, st->d_c, st->d_z, m, x, xinv);
    if (rc) return rc;
    st->m = m;
    if (st->nofold && st->deferred && st->fold_pending && st->tail_host.size() == 4) {
        // Two rounds have run beside the fold launched two rounds ago: switch to its output.  The challenge products since
        // then are s[0..4) (the newest challenge is the lowest index bit and s[0] = 1): nothing to upload, s just gets shorter.
        HALO_HIP(hipStreamWaitEvent(ctx->streams[0], st->ev_fold, 0));
        st->G_src = st->fold_dst;
        st->M = st->fold_m;
        st->s_len = 4;
        st->s_host.swap(st->tail_host);
        st->tail_host.clear();
        st->fold_pending = false;
        st->deferred = st->M > kNoFoldSize;
        if (!st->deferred) return HALO_OK;  // (the key stays as it is for the remaining rounds; s goes on from its 4 entries)
    }
    if (st->nofold && st->deferred && !st->fold_pending && st->s_len == 4) {
        // two rounds are due: G[j] <- G[j] + s1 G[j+m] + s2 G[j+2m] + s3 G[j+3m] with one shared doubling chain
        const size_t Mcur = 4 * m;
        // Measured (tools/open_loop.py, medians of 11, one box): opens of 2^18 points 9.33 -> 8.99 ms with the folds beside the
        // rounds; at 2^19 12.35 -> 12.61 and at 2^20 15.68 -> 16.43 (only the last fold beside: 15.89): there the rounds over the
        // larger key lose more than the hidden fold returns -- a fold's waves sit on every CU (one per SIMD, 232-252 registers,
        // the lane form 27 KB of LDS) and the rounds' 1024-thread sort blocks (140 KB of LDS) cannot start beside them.
        // Hence the automatic mode: opens of at most 2^18 points.
        const bool want_async = ctx->fold_async > 0 || (ctx->fold_async < 0 && st->n <= ((size_t)1 << 18));
        if (want_async && Mcur <= ((size_t)1 << 18) && m >= 64 && st->g_off + m <= (st->borrowed ? ctx->ipa_bufs.cap_n : st->n)) {
            // ... BESIDE the next two rounds (see halo_ipa::fold_pending): on the fourth stream, behind whatever wrote the
            // key it reads (stream 0: an earlier in-line fold; stream 3 itself: the previous fold of this kind)
            uint32_t *dst = st->d_G + 32 * st->g_off;  // (32 words = one 128-byte point, curve.hpp AFF_STRIDE)
            HALO_HIP(hipEventRecord(st->ev, ctx->streams[0]));
            HALO_HIP(hipStreamWaitEvent(ctx->streams[3], st->ev, 0));
            {
                hipStream_t saved = ctx->stream;
                ctx->stream = ctx->streams[3];
                rc = ipa_fold_points4(ctx, st->G_src, dst, m, &st->s_host[1]);
                ctx->stream = saved;
            }
            if (rc) return rc;
            HALO_HIP(hipEventRecord(st->ev_fold, ctx->streams[3]));
            st->fold_pending = true;
            st->fold_dst = dst;
            st->fold_m = m;
            st->g_off += m;
            st->tail_host.assign(1, host::Fr::one());
            return HALO_OK;
        }
        rc = ipa_fold_points4(ctx, st->G_src, st->d_G, m, &st->s_host[1]);
        if (rc) return rc;
        st->G_src = st->d_G;
        st->g_off = m;  // (in place from here on would overwrite this key: later folds go behind it)
        st->deferred = m > kNoFoldSize;  // below the switch size the key stays as it is for the remaining rounds
        st->s_host.assign(1, host::Fr::one());
        if (m > 1) return ipa_enter_nofold(st);
        st->nofold = false;  // the last two rounds: G[0] is U
        st->M = 1;
        st->s_len = 1;
        return HALO_OK;
    }
    if (!st->nofold && m <= kNoFoldSize && m > 1) rc = ipa_enter_nofold(st);
    return rc;
}

int halo_ipa_finish(halo_ipa *st, uint64_t U[12], uint64_t c[4]) {
    if (!st) { set_error("null ipa state"); return HALO_E_ARG; }
    halo_ctx *ctx = st->ctx;
    HALO_CTX(ctx);
    if (st->m != 1) { set_error("ipa_finish: rounds remaining"); return HALO_E_ARG; }
    if (st->fold_pending) {  // (cannot happen with >= 2 rounds behind every fold launched beside them; kept for safety)
        HALO_HIP(hipStreamWaitEvent(ctx->streams[0], st->ev_fold, 0));
        st->fold_pending = false;
    }
    int rc;
    const bool u_from_last_round = tuning().u_from_last_round;  // development switch
    if (st->nofold && st->M > 1 && st->last_valid && st->last_folded && u_from_last_round && !st->last_c0.is_zero() && !st->last_c1.is_zero()) {
        // U = G_final[0] = sum_b s''[b] K[b] with s''[2t + u] = s[t] xi^u (the last fold): U = A + xi B, A / B = the sums of
        // s[t] K[2t] / s[t] K[2t + 1].  The last round had two coefficients left, so its MSMs were exactly L' = c1 A and
        // R' = c0 B (k_nofold_expand with m = 2: FL[2t] = c[1] s[t], FR[2t + 1] = c[0] s[t]): U = L' / c1 + (xi / c0) R' -- two
        // scalar multiples on the host (~80 us, side by side on two threads) instead of one more MSM over the key (0.45 ms at
        // 2^14 points), and c = c0 + xi^-1 c1 (pcdl.rs:222) without another copy.  A zero coefficient takes the MSM below.
        host::Fr inv01 = (st->last_c0 * st->last_c1).inv();  // one inversion for both
        host::Fr a = inv01 * st->last_c0, b = inv01 * st->last_c1 * st->last_xi;  // 1 / c1, xi / c0
        host::Point Lp = st->last_L, Rp = st->last_R, Ua, Ub;
        ctx->worker.submit([&Ua, &Lp, &a] { Ua = Lp.mul(a); });
        Ub = Rp.mul(b);
        ctx->worker.wait();
        (Ua + Ub).store_normalized(U);
        (st->last_c0 + st->last_xi_inv * st->last_c1).store(c);
        if (ctx->prof.on) {
            for (int k = 0; k < 4; ++k) HALO_HIP(hipStreamSynchronize(ctx->streams[k]));
            ctx->prof.collect();
        }
        return HALO_OK;
    }
    if (st->nofold && st->M > 1) {
        // U = G_final[0] = sum_t s[t] * G0[t]
        host::Point Up;
        rc = msm_run(ctx, st->G_src, st->d_s, true, st->M, &Up);
        if (!rc) rc = download(ctx, c, st->d_c, 4);
        if (rc) return rc;
        if (ctx->prof.on) ctx->prof.collect();
        Up.store_normalized(U);
        return HALO_OK;
    }
    uint64_t g[8];
    rc = aff_native_to_words(ctx, st->G_src, 1, ctx->d_tmp_a);
    if (!rc) rc = download(ctx, g, ctx->d_tmp_a, 8);
    if (!rc) rc = download(ctx, c, st->d_c, 4);
    if (rc) return rc;
    if (ctx->prof.on) ctx->prof.collect();
    host::Point::load_affine(g).store_normalized(U);
    return HALO_OK;
}

// hiding branch, sharded (pcdl.rs:140-150): this shard's slice of p_bar = q (X - z) from the rng stream and
// its share <p_bar_loc, G_loc> of C_bar (without the w_bar S term)
int halo_ipa_hiding_partial(halo_ipa *st, uint64_t rng_state, size_t deg, const uint64_t z[4], uint64_t stride, uint64_t offset,
                            uint64_t Cbar_part[12]) {
    if (!st || !z || !Cbar_part || stride == 0) { set_error("ipa_hiding_partial: bad argument"); return HALO_E_ARG; }
    halo_ctx *ctx = st->ctx;
    HALO_CTX(ctx);
    if (st->m != st->n) { set_error("ipa_hiding_partial: rounds already started"); return HALO_E_ARG; }
    if (deg == 0) { set_error("open: hiding needs p.degree() >= 1"); return HALO_E_ASSERT; }
    if (!st->d_pbar) {
        alloc_epoch_bump(ctx);
        size_t cap = st->borrowed ? ctx->ipa_bufs.cap_n : st->n;
        if (hipMalloc(&st->d_pbar, cap * 32) != hipSuccess) { set_error("ipa_hiding_partial: allocation failed"); return HALO_E_DEVICE; }
    }
    int rc = pbar_stream_dev(ctx, rng_state, deg, host::Fr::load(z), stride, offset, st->n, st->d_pbar);
    if (rc) return rc;
    st->pbar_valid = true;
    host::Point part;
    rc = msm_run(ctx, st->G_src, st->d_pbar, true, st->n, &part);
    if (rc) return rc;
    part.store_normalized(Cbar_part);
    return HALO_OK;
}
// p' = p + alpha p_bar on this shard (pcdl.rs:156)
int halo_ipa_apply_hiding(halo_ipa *st, const uint64_t alpha[4]) {
    if (!st || !alpha || !st->d_pbar || !st->pbar_valid) { set_error("ipa_apply_hiding: no p_bar on this state"); return HALO_E_ARG; }
    halo_ctx *ctx = st->ctx;
    HALO_CTX(ctx);
    if (st->m != st->n) { set_error("ipa_apply_hiding: rounds already started"); return HALO_E_ARG; }
    return axpy_dev(ctx, st->d_c, st->d_pbar, st->n, host::Fr::load(alpha));
}

int halo_ipa_finish_z(halo_ipa *st, uint64_t U[12], uint64_t c[4], uint64_t z0[4]) {
    int rc = halo_ipa_finish(st, U, c);
    if (rc) return rc;
    if (!z0) { set_error("ipa_finish_z: null pointer"); return HALO_E_ARG; }
    return download(st->ctx, z0, st->d_z, 4);
}

void halo_ipa_destroy(halo_ipa *st) {
    if (!st) return;
    if (g_rt_on && g_rt.rounds) {
        fprintf(stderr, "[halo] ipa host time over %ld rounds (us): enqueue %.0f  dot launch + wait %.0f  H' terms %.0f  wait for the MSMs %.0f  combine %.0f  fold calls %.0f\n",
                g_rt.rounds, g_rt.enqueue, g_rt.dots, g_rt.hterm, g_rt.wait, g_rt.combine, g_rt.fold);
        g_rt = RoundTiming();
    }
    halo_ctx *ctx = st->ctx;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        for (int k = 0; k < 4; ++k) (void)hipStreamSynchronize(ctx->streams[k]);  // folds, the second MSM, the dot products, a fold beside the rounds
    }
    if (ctx && st->counted_hot) ctx->worker.add_hot(-1);
    if (st->borrowed && ctx) {
        ctx->ipa_bufs.d_pbar = st->d_pbar;  // lazily allocated by the hiding branch of a sharded open: kept with the rest
        ctx->ipa_bufs.in_use = false;
    } else if (st->d_G) {
        if (ctx) alloc_epoch_bump(ctx);
        (void)hipFree(st->d_G);
        (void)hipFree(st->d_c);
        (void)hipFree(st->d_z);
        (void)hipFree(st->d_s); (void)hipFree(st->d_s2); (void)hipFree(st->d_FL); (void)hipFree(st->d_FR);
        (void)hipFree(st->d_pbar);
    }
    if (st->ev) (void)hipEventDestroy(st->ev);
    if (st->ev_fold) (void)hipEventDestroy(st->ev_fold);
    delete st;
}
size_t halo_ipa_len(const halo_ipa *st) { return st ? st->m : 0; }

// ------------------------------------------------------------------ measurement hooks
// (a multi-device context: the shards profile too, and count/get show the sums over parent and shards per kernel name)
static int prof_enable_one(halo_ctx *ctx, int on, bool reset) {
    HALO_HIP(hipSetDevice(ctx->device));
    for (int k = 0; k < HALO_SLOTS; ++k) HALO_HIP(hipStreamSynchronize(ctx->streams[k]));
    ctx->prof.collect();
    if (reset) {
        for (auto &e : ctx->prof.entries) { e.total_ms = 0; e.launches = 0; }
    } else {
        ctx->prof.on = on != 0;
        ctx->prof.dominant_only = on == 2;
    }
    return HALO_OK;
}
static void prof_merge(halo_ctx *ctx) {
    ctx->prof_merged = ctx->prof.entries;
    for (halo_ctx *s : ctx->shards)
        for (const ProfEntry &e : s->prof.entries) {
            bool found = false;
            for (ProfEntry &m : ctx->prof_merged)
                if (std::strcmp(m.name, e.name) == 0) { m.total_ms += e.total_ms; m.launches += e.launches; found = true; break; }
            if (!found) ctx->prof_merged.push_back(e);
        }
}
int halo_prof_enable(halo_ctx *ctx, int on) {
    HALO_CTX(ctx);
    for (halo_ctx *s : ctx->shards) { int rc = prof_enable_one(s, on, false); if (rc) return rc; }
    return prof_enable_one(ctx, on, false);
}
int halo_prof_reset(halo_ctx *ctx) {
    HALO_CTX(ctx);
    for (halo_ctx *s : ctx->shards) { int rc = prof_enable_one(s, 0, true); if (rc) return rc; }
    return prof_enable_one(ctx, 0, true);
}
int halo_prof_count(halo_ctx *ctx) {
    if (!ctx) return 0;
    prof_merge(ctx);
    return (int)ctx->prof_merged.size();
}
int halo_prof_get(halo_ctx *ctx, int i, const char **name, double *total_ms, long *launches) {
    if (!ctx || i < 0 || i >= (int)ctx->prof_merged.size()) { set_error("prof_get: index (call halo_prof_count first)"); return HALO_E_ARG; }
    if (name) *name = ctx->prof_merged[i].name;
    if (total_ms) *total_ms = ctx->prof_merged[i].total_ms;
    if (launches) *launches = ctx->prof_merged[i].launches;
    return HALO_OK;
}
// ---- host steps of a sharded pcdl::open (halo-accumulation_amd/sharded.py) ---------------------
int halo_open_start(const uint64_t C[12], const uint64_t z[4], const uint64_t *v_parts, size_t P, uint64_t v_out[4], uint64_t xi0[4],
                    uint64_t Hp_out[12]) {
    if (!C || !z || !v_parts || !v_out || !xi0 || !Hp_out || P == 0) { set_error("open_start: bad argument"); return HALO_E_ARG; }
    host::Fr v = host::Fr::zero();
    for (size_t i = 0; i < P; ++i) v = v + host::Fr::load(v_parts + 4 * i);  // p(z) = sum of the shards' <c, z>
    host::Transcript t;
    t.point(host::Point::load(C)); t.scalar(host::Fr::load(z)); t.scalar(v);
    host::Fr x0 = t.finish(0);  // pcdl.rs:180
    uint64_t S[12], H[12];
    halo_public_points(S, H);
    host::Point::load(H).mul(x0).store_normalized(Hp_out);  // pcdl.rs:181
    v.store(v_out);
    x0.store(xi0);
    return HALO_OK;
}
// hiding branch, host step (pcdl.rs:147-162): w_bar is the next scalar of the stream after the deg
// coefficients of q; C_bar = sum of the shards' parts + w_bar S; alpha = rho_0(C, z, v, C_bar);
// w' = w_bar alpha + w; C' = C + alpha C_bar - w' S.  *rng_state advances past q and w_bar.
int halo_open_hiding_combine(const uint64_t C[12], const uint64_t z[4], const uint64_t *v_parts, const uint64_t *Cbar_parts, size_t P,
                             const uint64_t w[4], uint64_t *rng_state, size_t deg, uint64_t Cbar[12], uint64_t alpha[4],
                             uint64_t w_prime[4], uint64_t C_prime[12]) {
    if (!C || !z || !v_parts || !Cbar_parts || !w || !rng_state || !Cbar || !alpha || !w_prime || !C_prime || P == 0) {
        set_error("open_hiding_combine: bad argument");
        return HALO_E_ARG;
    }
    host::Fr v = host::Fr::zero();
    host::Point Cb = host::Point::infinity();
    for (size_t i = 0; i < P; ++i) {
        v = v + host::Fr::load(v_parts + 4 * i);
        Cb = Cb + host::Point::load(Cbar_parts + 12 * i);
    }
    host::Rng rng{*rng_state + 4 * (uint64_t)deg * 0x9E3779B97F4A7C15ULL};  // past the deg coefficients of q (pcdl.rs:141)
    host::Fr w_bar = rng.scalar();                                          // pcdl.rs:147
    *rng_state = rng.state;
    uint64_t Sw[12], Hw[12];
    halo_public_points(Sw, Hw);
    host::Point S = host::Point::load(Sw), Cp = host::Point::load(C);
    Cb = (S.mul(w_bar) + Cb).normalized();                                  // pcdl.rs:150 via pedersen.rs:15-17
    host::Transcript t;
    t.point(Cp); t.scalar(host::Fr::load(z)); t.scalar(v); t.point(Cb);
    host::Fr a = t.finish(0);                                               // pcdl.rs:153
    host::Fr wp = w_bar * a + host::Fr::load(w);                            // pcdl.rs:159
    (Cp + Cb.mul(a) - S.mul(wp)).store_normalized(C_prime);                 // pcdl.rs:162
    Cb.store(Cbar);
    a.store(alpha);
    wp.store(w_prime);
    return HALO_OK;
}
// parts: P records of L 12 | R 12 | dot_l 4 | dot_r 4 in rank order
int halo_open_combine(const uint64_t *parts, size_t P, const uint64_t Hp[12], const uint64_t xi_prev[4], uint64_t L[12], uint64_t R[12],
                      uint64_t xi[4], uint64_t xi_inv[4]) {
    if (!parts || !Hp || !xi_prev || !L || !R || !xi || !xi_inv || P == 0) { set_error("open_combine: bad argument"); return HALO_E_ARG; }
    host::Point Lp = host::Point::infinity(), Rp = host::Point::infinity(), H = host::Point::load(Hp);
    // H' is fixed for a whole open: its window table is kept per calling thread (a generic double-and-add took 80 us per
    // term, 20 rounds x 2 terms of every sharded open)
    static thread_local host::FixedBaseTable hp_tab;
    if (!hp_tab.matches(H)) hp_tab = host::FixedBaseTable(H);
    host::Fr dl = host::Fr::zero(), dr = host::Fr::zero();
    for (size_t i = 0; i < P; ++i) {
        const uint64_t *r = parts + 32 * i;
        Lp = Lp + host::Point::load(r);
        Rp = Rp + host::Point::load(r + 12);
        dl = dl + host::Fr::load(r + 24);
        dr = dr + host::Fr::load(r + 28);
    }
    Lp = (Lp + hp_tab.mul(dl)).normalized();  // pcdl.rs:204
    Rp = (Rp + hp_tab.mul(dr)).normalized();  // pcdl.rs:208
    host::Transcript t;
    t.scalar(host::Fr::load(xi_prev)); t.point(Lp); t.point(Rp);
    host::Fr x = t.finish(0);  // pcdl.rs:212
    if (x.is_zero()) { set_error("open: challenge is zero (inverse().unwrap())"); return HALO_E_ASSERT; }
    Lp.store(L); Rp.store(R);
    x.store(xi);
    x.inv().store(xi_inv);
    return HALO_OK;
}

// The last lg P rounds of a sharded open (pcdl.rs:195-227 over the P gathered elements, P <= 64): host arithmetic only --
// a launch sequence per round for a handful of points is all latency, and the throw-away context the Python driver used to
// build for these rounds cost more to create and destroy than the rounds themselves.
int halo_open_tail(const uint64_t *recs, size_t P, const uint64_t Hp[12], const uint64_t xi_prev[4], uint64_t *Ls, uint64_t *Rs,
                   uint64_t U[12], uint64_t c_out[4]) {
    if (!recs || !Hp || !xi_prev || !U || !c_out || P == 0 || (P & (P - 1)) != 0 || P > 64 || (P > 1 && (!Ls || !Rs))) {
        set_error("open_tail: P must be a power of two of at most 64 elements, no null pointers");
        return HALO_E_ARG;
    }
    std::vector<host::Point> G(P);
    std::vector<host::Fr> c(P), z(P);
    for (size_t i = 0; i < P; ++i) {
        const uint64_t *r = recs + 20 * i;  // G_i (Jacobian, 12 words) | c_i | z_i
        G[i] = host::Point::load(r);
        c[i] = host::Fr::load(r + 12);
        z[i] = host::Fr::load(r + 16);
    }
    host::Point H = host::Point::load(Hp);
    host::Fr xi = host::Fr::load(xi_prev);
    size_t round = 0;
    for (size_t m = P / 2; m >= 1; m /= 2, ++round) {
        host::Fr dl = host::Fr::zero(), dr = host::Fr::zero();
        for (size_t j = 0; j < m; ++j) {
            dl = dl + c[m + j] * z[j];       // <c_r, z_l>   (:203)
            dr = dr + c[j] * z[m + j];       // <c_l, z_r>   (:207)
        }
        // the 2 m + 2 scalar multiples of the round (~80 us each) on the host pool; summed in index order
        std::vector<host::Point> term(2 * m + 2);
        pool_run(2 * m + 2, [&](size_t t) {
            if (t < m) term[t] = G[t].mul(c[m + t]);                    // <c_r, G_l>   (:204)
            else if (t < 2 * m) term[t] = G[t].mul(c[t - m]);           // <c_l, G_r>   (:208)
            else term[t] = H.mul(t == 2 * m ? dl : dr);
        });
        host::Point L = host::Point::infinity(), R = host::Point::infinity();
        for (size_t j = 0; j < m; ++j) { L = L + term[j]; R = R + term[m + j]; }
        L = (L + term[2 * m]).normalized();
        R = (R + term[2 * m + 1]).normalized();
        host::Transcript t;
        t.scalar(xi); t.point(L); t.point(R);
        host::Fr x = t.finish(0);  // :212
        if (x.is_zero()) { set_error("open: challenge is zero (inverse().unwrap())"); return HALO_E_ASSERT; }
        host::Fr xinv = x.inv();
        L.store(Ls + 12 * round);
        R.store(Rs + 12 * round);
        pool_run(m, [&](size_t j) {
            G[j] = G[j] + G[m + j].mul(x);   // :218
            c[j] = c[j] + xinv * c[m + j];   // :222
            z[j] = z[j] + x * z[m + j];      // :223
        });
        xi = x;
    }
    G[0].store_normalized(U);  // :230
    c[0].store(c_out);
    return HALO_OK;
}

int halo_rng_scalars_dev(halo_ctx *ctx, uint64_t *rng_state, size_t n, void *d_out) {
    HALO_CTX(ctx);
    if (!rng_state || (n && !d_out)) { set_error("rng_scalars: null pointer"); return HALO_E_ARG; }
    int rc = rng_scalars_dev(ctx, *rng_state, n, static_cast<uint64_t *>(d_out));
    if (rc) return rc;
    HALO_HIP(hipStreamSynchronize(ctx->stream));
    *rng_state += 4 * (uint64_t)n * 0x9E3779B97F4A7C15ULL;  // n scalars = 4n draws of the SplitMix64 stream
    return HALO_OK;
}
int halo_point_sum(const uint64_t *pts_jac, size_t k, uint64_t out[12]) {
    if (!out || (k && !pts_jac)) { set_error("point_sum: null pointer"); return HALO_E_ARG; }
    host::Point acc = host::Point::infinity();
    for (size_t i = 0; i < k; ++i) acc = acc + host::Point::load(pts_jac + 12 * i);  // fixed rank order 0..k-1
    acc.store_normalized(out);
    return HALO_OK;
}


int halo_set_table_mode(halo_ctx *ctx, int mode) {
    if (!ctx || mode < -1 || mode > 0) { set_error("table mode must be -1 (automatic) or 0 (never)"); return HALO_E_ARG; }
    if (mode == 0 && ctx->d_table) {  // "no table memory": a table already built is released, not just left unused
        HALO_CTX(ctx);
        int rc = table_release(ctx);
        if (rc) return rc;
    }
    ctx->table_mode = mode;
    ctx->table_retry_at = 0;
    ctx->table_status = 0;
    for (halo_ctx *sh : ctx->shards) (void)halo_set_table_mode(sh, mode);  // a multi-device context: its shards run the MSMs
    return HALO_OK;
}
int halo_set_fold_table(halo_ctx *ctx, int mode) {
    if (!ctx || mode < -1 || mode > 1) { set_error("fold table mode must be -1 (automatic), 0 (never) or 1 (at the first full-size open)"); return HALO_E_ARG; }
    if (mode == 0) {  // releases the table, and whatever the helper thread of the automatic mode has obtained
        HALO_CTX(ctx);
        for (int k = 0; k < HALO_SLOTS; ++k) HALO_HIP(hipStreamSynchronize(ctx->streams[k]));
        foldtab_release(ctx);
    }
    ctx->fold_table_mode = mode;
    ctx->foldtab_retry_at = 0;  // (a table that could not be had before is considered again)
    ctx->foldtab_status = 0;
    return HALO_OK;
}
}  // extern "C"
// ------------------------------------------------------------------ budget for optional device memory
namespace halo {
namespace {
struct DeviceBudget { size_t budget = 0, used = 0; bool known = false; };
std::mutex g_budget_mu;
DeviceBudget g_budget[64];
// HALO_MEMORY_BUDGET=<bytes>[K|M|G] in the environment (read by tuning()) replaces the default (for hosts that cannot call the setter: the Rust shim)
DeviceBudget &budget_of(int device) {  // (g_budget_mu held; the device is current)
    DeviceBudget &b = g_budget[device & 63];
    if (!b.known) {
        b.known = true;
        size_t free_b = 0, total = 0;
        if (tuning().memory_budget_set) b.budget = tuning().memory_budget;
        else if (hipMemGetInfo(&free_b, &total) == hipSuccess) b.budget = total / 6;  // 48 GB of an MI355X's 288
        else (void)hipGetLastError();
    }
    return b;
}
}  // namespace
bool table_budget_reserve(halo_ctx *ctx, size_t bytes) {
    std::lock_guard<std::mutex> lk(g_budget_mu);
    DeviceBudget &b = budget_of(ctx->device);
    size_t free_b = 0, total = 0;
    if (hipMemGetInfo(&free_b, &total) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (b.used + bytes > b.budget || free_b / 2 < bytes) {
        if (debug_trace()) fprintf(stderr, "[halo] optional table of %zu bytes refused on device %d: %zu of %zu budget bytes in use, %zu free\n", bytes, ctx->device, b.used, b.budget, free_b);
        return false;
    }
    b.used += bytes;
    ctx->share->budget_held += bytes;  // (on the books of the key: a clone may outlive the context that built a table)
    return true;
}
void table_budget_release(halo_ctx *ctx, size_t bytes) {
    std::lock_guard<std::mutex> lk(g_budget_mu);
    DeviceBudget &b = g_budget[ctx->device & 63];
    if (bytes > ctx->share->budget_held) bytes = ctx->share->budget_held;
    ctx->share->budget_held -= bytes;
    b.used = b.used >= bytes ? b.used - bytes : 0;
}
}  // namespace halo
extern "C" {
int halo_set_memory_budget(halo_ctx *ctx, size_t bytes) {
    HALO_CTX(ctx);
    {
        std::lock_guard<std::mutex> lk(g_budget_mu);
        DeviceBudget &b = budget_of(ctx->device);
        b.budget = bytes;
    }
    // a table that was refused is considered again at the next opportunity
    ctx->foldtab_retry_at = 0; ctx->table_retry_at = 0;
    for (halo_ctx *sh : ctx->shards) (void)halo_set_memory_budget(sh, bytes);  // (a shard on another device: that device's budget)
    return HALO_OK;
}
/* what: 0 = bytes of the MSM fixed-base table, 1 = bytes of the fold table, 2 = microseconds its build took, 3 = the budget for optional
 * table memory on this context's device, 4 = bytes of it in use (all contexts of the process), 5 / 6 = status of the fold table / the
 * MSM table: 0 nothing yet, 1 memory requested, 2 built, 3 over the budget, 4 allocation failed (tried again later), 5 off */
size_t halo_ctx_info(const halo_ctx *ctx, int what) {
    if (!ctx) return 0;
    if (what == 0) return ctx->d_table ? (size_t)ctx->tbl.W * ctx->n * 128 : 0;
    if (what == 1) return ctx->foldtab_bytes;
    if (what == 2) return (size_t)(ctx->foldtab_build_ms * 1e3);
    if (what == 3 || what == 4) {
        std::lock_guard<std::mutex> lk(g_budget_mu);
        DeviceBudget &b = g_budget[ctx->device & 63];
        if (!b.known) {  // (the default is a fraction of THIS device's memory: asked once, the caller's current device is put back)
            int prev = -1;
            (void)hipGetDevice(&prev);
            (void)hipSetDevice(ctx->device);
            (void)budget_of(ctx->device);
            if (prev >= 0) (void)hipSetDevice(prev);
        }
        return what == 3 ? b.budget : b.used;
    }
    if (what == 5) return ctx->fold_table_mode == 0 ? 5 : ctx->d_foldtab ? 2 : (size_t)ctx->foldtab_status;
    if (what == 6) return ctx->table_mode == 0 ? 5 : ctx->d_table ? 2 : (size_t)ctx->table_status;
    return 0;
}

// ------------------------------------------------------------------ primitive hooks

}  // extern "C"
