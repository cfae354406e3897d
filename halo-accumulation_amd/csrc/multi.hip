// Multi-device contexts ("halo_ctx_create(device_ids[], n_dev, ...) uploads (and shards) bases once";
// 8e: an MSM is a sum over independent index terms).  One process drives several GPUs:
//
//   * the handle the caller gets is a full context on devices[0] over the whole key -- every entry point of the
//     library works on it unchanged (IPA state, pcdl / acc level, h(X), ...);
//   * it owns one SHARD context per device over that device's index block of the key (block k = [k N / P, (k+1) N / P),
//     multiples of 4; each shard derives or receives only its block and builds its own fixed-base table on first use);
//   * an MSM over GS[off, off + n) -- halo_msm, halo_msm_dev, the begin/end halves and the library's own synchronous
//     MSMs over the key (commit, check, h_commit) -- is cut along the block boundaries, every shard enqueues its stretch
//     on its own device and stream, the shards' window sums are combined by their helper threads in parallel and the
//     P partial points are added on the host in block order (halo_point_sum's rule): bit-identical to the one-device
//     result, which is a normalised point.  Device-resident scalars are read in place by the shard on the same device
//     and copied peer-to-peer (xGMI) for the others; host scalars go straight to each device over its own PCIe link.
//
// There is no collective here: one process owns all partials.  Ranks of a torch.distributed job each hold a plain
// context and all-gather 96 bytes instead (halo-accumulation_amd/sharded.py).
#include <algorithm>

#include "curve.hpp"
#include "internal.hpp"

namespace halo {

static size_t block_lo(size_t N, int P, int k) {
    if (k >= P) return N;
    size_t lo = (size_t)((unsigned __int128)N * (unsigned)k / (unsigned)P);
    return lo / 4 * 4;  // (vector loads of digits want multiples of 4)
}

void multi_destroy(halo_ctx *ctx) {
    for (halo_ctx *s : ctx->shards) halo_ctx_destroy(s);
    ctx->shards.clear();
    ctx->shard_lo.clear();
}

// shards over the blocks of the parent's key: derived on their own device (URS rule) or copied from the host array
int multi_attach_shards(halo_ctx *ctx, const int *devices, int n_dev, const uint64_t *bases_affine, uint64_t first_index) {
    size_t N = ctx->n;
    ctx->shard_lo.resize((size_t)n_dev + 1);
    for (int k = 0; k <= n_dev; ++k) ctx->shard_lo[k] = block_lo(N, n_dev, k);
    for (int k = 0; k < n_dev; ++k) {
        size_t lo = ctx->shard_lo[k], len = ctx->shard_lo[k + 1] - lo;
        halo_ctx *s = nullptr;
        int rc = bases_affine ? halo_ctx_create(devices[k], bases_affine + 8 * lo, len, &s) : halo_ctx_create_urs(devices[k], first_index + lo, len, &s);
        if (rc) { multi_destroy(ctx); return rc; }
        s->parent = ctx;
        ctx->shards.push_back(s);
        if (devices[k] != ctx->device) {  // direct peer copies of device-resident scalars (xGMI); without it HIP stages through the host
            (void)hipSetDevice(devices[k]);
            (void)hipDeviceEnablePeerAccess(ctx->device, 0);
            (void)hipSetDevice(ctx->device);
            (void)hipDeviceEnablePeerAccess(devices[k], 0);
            (void)hipGetLastError();  // "already enabled" is fine
        }
    }
    (void)hipSetDevice(ctx->device);
    return HALO_OK;
}

bool multi_takes(const halo_ctx *ctx, const uint32_t *d_bases, size_t n) {
    return !ctx->shards.empty() && d_bases >= ctx->d_bases && d_bases + AFF_STRIDE * n <= ctx->d_bases + AFF_STRIDE * ctx->n;
}

static int device_of(const void *p, int fallback) {
    hipPointerAttribute_t a;
    if (p && hipPointerGetAttributes(&a, p) == hipSuccess) return a.device;
    (void)hipGetLastError();
    return fallback;
}

// Enqueue the stretches of GS[off, off + n) on the shards' slot `slot`.  Exactly one of host_scalars / dev_scalars is set.
int multi_begin(halo_ctx *ctx, int slot, size_t off, size_t n, const uint64_t *host_scalars, const uint64_t *dev_scalars, bool mont) {
    if (slot < 0 || slot >= HALO_SLOTS) { set_error("msm: slot out of range"); return HALO_E_ARG; }
    halo_ctx::Fan &fan = ctx->fan[slot];
    if (fan.active) { set_error("msm: slot already has an MSM in flight"); return HALO_E_ARG; }
    const int P = (int)ctx->shards.size();
    fan.used.assign((size_t)P, 0);
    int src_dev = dev_scalars ? device_of(dev_scalars, ctx->device) : -1;
    int rc = HALO_OK;
    std::vector<int> rcs((size_t)P, HALO_OK);
    std::vector<std::string> errs((size_t)P);
    for (int k = 0; k < P && !rc; ++k) {
        size_t lo = ctx->shard_lo[k], hi = ctx->shard_lo[k + 1];
        size_t a = std::max(off, lo), b = std::min(off + n, hi);
        if (a >= b) continue;
        halo_ctx *s = ctx->shards[k];
        fan.used[k] = 1;
        if (host_scalars) {
            // every device has its own PCIe link: the copies run in parallel, each issued by its shard's helper thread
            const uint64_t *src = host_scalars + 4 * (a - off);
            s->worker.submit([s, slot, a, b, lo, src, mont, &rcs, &errs, k] {
                (void)hipSetDevice(s->device);
                rcs[k] = msm_host_begin(s, slot, a - lo, b - a, src, mont ? 1 : 0);
                if (rcs[k]) errs[k] = halo_last_error();
            });
            continue;
        }
        // (no early return in here: whatever has been enqueued on other shards is drained below if this one fails)
        const uint64_t *src = dev_scalars + 4 * (a - off);
        hipError_t e = hipSetDevice(s->device);
        const bool staged = src_dev != s->device || dev_hooks().force_peer_copy;  // (development library's hook: the copy path on a one-GPU box)
        if (e == hipSuccess && staged) {  // the scalars live on another GPU: peer copy on this shard's stream, in front of its launches
            if (!s->d_slot_scalars[slot]) {
                alloc_epoch_bump(s);
                e = hipMalloc(&s->d_slot_scalars[slot], (s->n < 64 ? 64 : s->n) * 32);
            }
            if (e == hipSuccess) e = hipMemcpyPeerAsync(s->d_slot_scalars[slot], s->device, src, src_dev, (b - a) * 32, s->streams[slot]);
            src = s->d_slot_scalars[slot];
        }
        // (same device: the shard reads the caller's buffer in place; as for halo_msm_dev on a plain context the caller has
        // synchronised whatever wrote it)
        rc = e != hipSuccess ? hip_fail(e, "multi-device MSM: peer copy of the scalars") : msm_enqueue(s, slot, s->d_bases + 32 * (a - lo), src, mont, b - a);
    }
    if (host_scalars)
        for (int k = 0; k < P; ++k)
            if (fan.used[k]) {
                ctx->shards[k]->worker.wait();
                if (rcs[k] && !rc) { rc = rcs[k]; set_error(errs[k]); }
            }
    (void)hipSetDevice(ctx->device);
    fan.active = true;  // (also after a failure: multi_end drains whatever was enqueued)
    fan.batch = 0;
    if (rc) { host::Point dummy; std::string keep = halo_last_error(); (void)multi_end(ctx, slot, &dummy); set_error(keep); }
    return rc;
}

// Wait for the shards, combine their window sums (each on its own helper thread), add the partials in block order.
int multi_end(halo_ctx *ctx, int slot, host::Point *out) {
    if (slot < 0 || slot >= HALO_SLOTS || !ctx->fan[slot].active) { set_error("msm: nothing in flight on this slot"); return HALO_E_ARG; }
    halo_ctx::Fan &fan = ctx->fan[slot];
    const int P = (int)ctx->shards.size();
    std::vector<host::Point> part((size_t)P, host::Point::infinity());
    std::vector<int> rcs((size_t)P, HALO_OK);
    std::vector<std::string> errs((size_t)P);
    for (int k = 0; k < P; ++k) {
        if (!fan.used[k]) continue;
        halo_ctx *s = ctx->shards[k];
        if (!s->wss[slot].in_flight) { fan.used[k] = 0; continue; }  // (its enqueue failed)
        s->worker.submit([s, slot, k, &part, &rcs, &errs] {
            (void)hipSetDevice(s->device);
            rcs[k] = msm_finish(s, slot, &part[k]);
            if (rcs[k]) errs[k] = halo_last_error();
        });
    }
    int rc = HALO_OK;
    host::Point acc = host::Point::infinity();
    for (int k = 0; k < P; ++k) {
        if (!fan.used[k]) continue;
        ctx->shards[k]->worker.wait();
        if (rcs[k] && !rc) { rc = rcs[k]; set_error(errs[k]); }
        acc = acc + part[k];  // block order 0 .. P-1
    }
    fan.active = false;
    (void)hipSetDevice(ctx->device);
    *out = acc;
    return rc;
}

// Batched form (halo_msm_dev_batch_begin/_end): `members.count` MSMs over GS[off, off + n), one resident scalar array each.
// Every shard runs ITS stretch of all members as one batched launch sequence (which is what makes a 2^17-point block a full
// launch: msm.hip, small-key table plan), and the shards' per-member partials are added in block order.
int multi_batch_begin(halo_ctx *ctx, int slot, size_t off, size_t n, const MsmBatch &members, bool mont) {
    if (slot < 0 || slot >= HALO_SLOTS) { set_error("msm: slot out of range"); return HALO_E_ARG; }
    halo_ctx::Fan &fan = ctx->fan[slot];
    if (fan.active) { set_error("msm: slot already has an MSM in flight"); return HALO_E_ARG; }
    if (members.parts != 1) { set_error("msm: window shards are not split over the devices of a multi-device context"); return HALO_E_ARG; }
    const int P = (int)ctx->shards.size();
    fan.used.assign((size_t)P, 0);
    int rc = HALO_OK;
    for (int k = 0; k < P && !rc; ++k) {
        size_t lo = ctx->shard_lo[k], hi = ctx->shard_lo[k + 1];
        size_t a = std::max(off, lo), b = std::min(off + n, hi);
        if (a >= b) continue;
        halo_ctx *s = ctx->shards[k];
        fan.used[k] = 1;
        hipError_t e = hipSetDevice(s->device);
        MsmBatch mine = members;
        for (int m = 0; m < members.count && e == hipSuccess; ++m) {
            const uint64_t *src = members.scalars[m] + 4 * (a - off);
            int src_dev = device_of(src, ctx->device);
            if (src_dev != s->device || dev_hooks().force_peer_copy) {  // (as multi_begin: peer copy in front of the launches)
                size_t need = (size_t)members.count * (s->n < 64 ? 64 : s->n) * 32;
                if (s->batch_scalars_bytes[slot] < need) {
                    alloc_epoch_bump(s);
                    (void)hipStreamSynchronize(s->streams[slot]);
                    (void)hipFree(s->d_batch_scalars[slot]);
                    s->d_batch_scalars[slot] = nullptr;
                    s->batch_scalars_bytes[slot] = 0;
                    e = hipMalloc(&s->d_batch_scalars[slot], need);
                    if (e == hipSuccess) s->batch_scalars_bytes[slot] = need;
                }
                uint64_t *dst = s->d_batch_scalars[slot] + (size_t)m * (s->n < 64 ? 64 : s->n) * 4;
                if (e == hipSuccess) e = hipMemcpyPeerAsync(dst, s->device, src, src_dev, (b - a) * 32, s->streams[slot]);
                src = dst;
            }
            mine.scalars[m] = src;
            mine.base_off[m] = 0;
        }
        rc = e != hipSuccess ? hip_fail(e, "multi-device MSM: peer copy of the scalars") : msm_enqueue_batch(s, slot, s->d_bases + 32 * (a - lo), mine, mont, b - a);
    }
    (void)hipSetDevice(ctx->device);
    fan.active = true;
    fan.batch = members.count;
    if (rc) { host::Point dummy[MSM_MAX_BATCH]; std::string keep = halo_last_error(); (void)multi_batch_end(ctx, slot, dummy, members.count); set_error(keep); }
    return rc;
}
int multi_batch_end(halo_ctx *ctx, int slot, host::Point *out, int count) {
    if (slot < 0 || slot >= HALO_SLOTS || !ctx->fan[slot].active) { set_error("msm: nothing in flight on this slot"); return HALO_E_ARG; }
    halo_ctx::Fan &fan = ctx->fan[slot];
    if (fan.batch != count) { set_error("msm: this slot holds a batch of a different size"); return HALO_E_ARG; }
    const int P = (int)ctx->shards.size();
    std::vector<host::Point> part((size_t)P * MSM_MAX_BATCH, host::Point::infinity());
    std::vector<int> rcs((size_t)P, HALO_OK);
    std::vector<std::string> errs((size_t)P);
    for (int k = 0; k < P; ++k) {
        if (!fan.used[k]) continue;
        halo_ctx *s = ctx->shards[k];
        if (!s->wss[slot].in_flight) { fan.used[k] = 0; continue; }
        s->worker.submit([s, slot, k, count, &part, &rcs, &errs] {
            (void)hipSetDevice(s->device);
            rcs[k] = msm_finish_batch(s, slot, &part[(size_t)k * MSM_MAX_BATCH], count);
            if (rcs[k]) errs[k] = halo_last_error();
        });
    }
    int rc = HALO_OK;
    for (int m = 0; m < count; ++m) out[m] = host::Point::infinity();
    for (int k = 0; k < P; ++k) {
        if (!fan.used[k]) continue;
        ctx->shards[k]->worker.wait();
        if (rcs[k] && !rc) { rc = rcs[k]; set_error(errs[k]); }
        for (int m = 0; m < count; ++m) out[m] = out[m] + part[(size_t)k * MSM_MAX_BATCH + m];  // block order 0 .. P-1
    }
    fan.active = false;
    fan.batch = 0;
    (void)hipSetDevice(ctx->device);
    return rc;
}

int multi_run(halo_ctx *ctx, size_t off, size_t n, const uint64_t *dev_scalars, bool mont, host::Point *out) {
    // the library's own synchronous MSMs have just written their scalars on the parent's stream
    HALO_HIP(hipStreamSynchronize(ctx->stream));
    int rc = multi_begin(ctx, 0, off, n, nullptr, dev_scalars, mont);
    if (rc) return rc;
    return multi_end(ctx, 0, out);
}

// The synchronous host-scalar form (halo_msm, pcdl::commit with host coefficients): every shard's helper thread runs its block of
// GS[off, off + n) through msm_host_run on its own device -- its scalars over its own PCIe link, in stretches where that pays
// (abi.hip: a shard's block of 2^21 points of an n = 2^24 MSM copies under its own kernels) -- and the partial points are added in
// block order.  Same point as multi_begin + multi_end give.
int multi_host_run(halo_ctx *ctx, size_t off, size_t n, const uint64_t *scalars, bool mont, host::Point *out) {
    if (ctx->fan[0].active) { set_error("msm: slot already has an MSM in flight"); return HALO_E_ARG; }
    const int P = (int)ctx->shards.size();
    std::vector<host::Point> part((size_t)P, host::Point::infinity());
    std::vector<int> rcs((size_t)P, HALO_OK);
    std::vector<char> used((size_t)P, 0);
    std::vector<std::string> errs((size_t)P);
    for (int k = 0; k < P; ++k) {
        size_t lo = ctx->shard_lo[k], hi = ctx->shard_lo[k + 1];
        size_t a = std::max(off, lo), b = std::min(off + n, hi);
        if (a >= b) continue;
        halo_ctx *s = ctx->shards[k];
        used[k] = 1;
        const uint64_t *src = scalars + 4 * (a - off);
        s->worker.submit([s, a, b, lo, src, mont, &rcs, &errs, &part, k] {
            (void)hipSetDevice(s->device);
            rcs[k] = msm_host_run(s, a - lo, b - a, src, b - a, mont ? 1 : 0, &part[k]);
            if (rcs[k]) errs[k] = halo_last_error();
        });
    }
    int rc = HALO_OK;
    for (int k = 0; k < P; ++k)
        if (used[k]) {
            ctx->shards[k]->worker.wait();
            if (rcs[k] && !rc) { rc = rcs[k]; set_error(errs[k]); }
        }
    (void)hipSetDevice(ctx->device);
    if (rc) return rc;
    host::Point acc = host::Point::infinity();
    for (int k = 0; k < P; ++k) acc = acc + part[k];  // block order
    *out = acc;
    return HALO_OK;
}

}  // namespace halo
