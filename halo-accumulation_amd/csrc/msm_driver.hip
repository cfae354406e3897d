// The host side of the MSM pipelines: slot workspaces, the launch driver with its per-key cache of launch graphs
// (msm_enqueue_batch), the wait for the published window sums (msm_wait) and their combination on the host (msm_combine_member).
// The launch sequences themselves are msm_general.hip (msm_enqueue_launches), msm_table.hip (tmsm_enqueue_launches), smsm.hip.
#include <atomic>
#include <cstring>
#include <thread>

#include "msm_kernels.hpp"

namespace halo {

// ------------------------------------------------------------------------------ workspace
static size_t max_counts() { return (size_t)16 * 32768; }  // c = 16 is the largest W*B over c in [4, 16]

struct WorkspaceNeed {
    size_t n, counts, sorted, tasks, hist, windows;
};
static void workspace_release(MsmWorkspace &ws) {
    if (debug_trace()) fprintf(stderr, "[halo] workspace release %p\n", (void *)&ws);
    uint64_t *p64[] = {ws.d_canon, ws.d_winsum};
    uint32_t *p32[] = {ws.d_buckets, ws.d_seg, ws.d_counts, ws.d_starts, ws.d_hist, ws.d_blockoff, ws.d_sorted, ws.d_presort, ws.d_ntask, ws.d_toff,
                       ws.d_tblockoff, ws.d_biglist, ws.d_meta, ws.d_task_g, ws.d_order};
    for (auto p : p64) (void)hipFree(p);
    for (auto p : p32) (void)hipFree(p);
    (void)hipFree(ws.d_fine16);
    if (ws.h_winsum) (void)hipHostFree(ws.h_winsum);
    if (ws.h_done) (void)hipHostFree(ws.h_done);
    for (auto &g : ws.graphs) if (g.exec) (void)hipGraphExecDestroy(g.exec);
    ws = MsmWorkspace();
}
static int workspace_alloc_buffers(MsmWorkspace &ws, const WorkspaceNeed &need);
static int workspace_alloc(MsmWorkspace &ws, const WorkspaceNeed &need) {
    int rc = workspace_alloc_buffers(ws, need);
    if (rc) workspace_release(ws);  // a failed allocation part-way leaves nothing behind
    return rc;
}
static int workspace_alloc_buffers(MsmWorkspace &ws, const WorkspaceNeed &need) {
    ws.cap_n = need.n;
    ws.cap_counts = need.counts;
    ws.cap_sorted = need.sorted;
    ws.cap_tasks = need.tasks;
    ws.cap_hist = need.hist;
    ws.cap_windows = need.windows;
    // the LDS histograms need up to 128 KiB of dynamic LDS per block (160 KiB per CU on gfx950)
    { int rc = msm_general_prepare(); if (rc) return rc; }
    { int rc = msm_table_prepare(); if (rc) return rc; }
    { int rc = smsm_prepare(); if (rc) return rc; }
    HALO_HIP(hipMalloc(&ws.d_canon, ws.cap_sorted * 2 + 64));  // u16 digits, n * W of them
    HALO_HIP(hipMalloc(&ws.d_hist, ws.cap_hist * 4));          // [w][chunk][b]
    HALO_HIP(hipMalloc(&ws.d_counts, ws.cap_counts * 4));
    HALO_HIP(hipMalloc(&ws.d_starts, ws.cap_counts * 4));
    HALO_HIP(hipMalloc(&ws.d_blockoff, 1024 * 4));
    HALO_HIP(hipMalloc(&ws.d_sorted, ws.cap_sorted * 4));
    HALO_HIP(hipMalloc(&ws.d_presort, ws.cap_sorted * 4));  // coarse runs of the two-level sort
    HALO_HIP(hipMalloc(&ws.d_buckets, ws.cap_tasks * XYZZ_WORDS * 4));
    HALO_HIP(hipMalloc(&ws.d_ntask, ws.cap_counts * 4));
    HALO_HIP(hipMalloc(&ws.d_toff, ws.cap_counts * 4));
    HALO_HIP(hipMalloc(&ws.d_tblockoff, 1024 * 4));
    HALO_HIP(hipMalloc(&ws.d_biglist, ws.cap_counts * 4));
    HALO_HIP(hipMalloc(&ws.d_meta, 1024));
    HALO_HIP(hipMalloc(&ws.d_task_g, ws.cap_tasks * 4));
    HALO_HIP(hipMalloc(&ws.d_order, ws.cap_tasks * 16));  // uint4 per task (k_msm_task_order)
    HALO_HIP(hipMalloc(&ws.d_seg, ws.cap_windows * 64 * 2 * XYZZ_WORDS * 4));
    HALO_HIP(hipMalloc(&ws.d_winsum, ws.cap_windows * 12 * 8));
    // (the host reads both while the kernel that writes them is still running: fine-grained coherent, said explicitly)
    HALO_HIP(hipHostMalloc(&ws.h_winsum, ws.cap_windows * 12 * 8, hipHostMallocCoherent));
    HALO_HIP(hipHostMalloc(&ws.h_done, 64, hipHostMallocCoherent));
    *ws.h_done = 0;
    ws.done_expect = 0;
    if (debug_trace())  // address ranges, so that a faulting address can be mapped to a buffer
        fprintf(stderr, "[halo] workspace %p: digits=[%p,+%zu) sorted=[%p,+%zu) presort=[%p,+%zu) partials=[%p,+%zu) hist=[%p,+%zu) counts=%p starts=%p ntask=%p toff=%p task_g=%p order=%p seg=%p winsum=%p\n",
                (void *)&ws, (void *)ws.d_canon, ws.cap_sorted * 2 + 64, (void *)ws.d_sorted, ws.cap_sorted * 4, (void *)ws.d_presort, ws.cap_sorted * 4,
                (void *)ws.d_buckets, ws.cap_tasks * XYZZ_WORDS * 4, (void *)ws.d_hist, ws.cap_hist * 4, (void *)ws.d_counts, (void *)ws.d_starts,
                (void *)ws.d_ntask, (void *)ws.d_toff, (void *)ws.d_task_g, (void *)ws.d_order, (void *)ws.d_seg, (void *)ws.d_winsum);
    return HALO_OK;
}
// capacity for any single MSM of up to n points, whatever the window size
int msm_workspace_alloc(halo_ctx *ctx, size_t n, int slot) {
    if (n < 64) n = 64;
    WorkspaceNeed need;
    need.n = n;
    // sorted entries: n * W; the automatic plan has W <= 32 for n >= 4096 (c >= 8) and W <= 64 below
    need.sorted = n >= 4096 ? n * 32 : n * 64;
    need.counts = max_counts();
    // (a key of the c = 20 table plan: room for the TWO bucket sets of a tagged launch from the start -- growing the workspace at
    // the first tagged launch cost the first open of a context a second of hipFree / hipMalloc)
    if (n >= ((size_t)1 << 20) && need.counts < ((size_t)1 << 20)) need.counts = (size_t)1 << 20;
    {   // tasks: one per non-empty bucket plus entries / kmax (kmax = 16 only below 2^18 points, W <= 32 there)
        size_t small = need.sorted < ((size_t)1 << 23) ? need.sorted : ((size_t)1 << 23);
        size_t extra = need.sorted / KMAX > small / 16 ? need.sorted / KMAX : small / 16;
        need.tasks = need.counts + extra + 1;
    }
    need.hist = (size_t)256 * 32768 + need.counts;  // W * nchunks <= 256 blocks of B <= 32768 counters
    need.windows = 64;
    {   // the table pipeline runs a large MSM in pieces and keeps 2 x 16 window sums per piece
        size_t pieces = (n + TBL_PIECE - 1) / TBL_PIECE;
        if (48 * pieces > need.windows) need.windows = 48 * pieces;  // (or 24 (S, T) pairs per piece: k_msm_reduce_rc)
        if (need.windows < 128) need.windows = 128;                  // (a batch of 8 of the small-key plan: 8 x 8 pairs)
    }
    alloc_epoch_bump(ctx);
    return workspace_alloc(ctx->wss[slot], need);
}
void msm_workspace_free(halo_ctx *ctx) {
    alloc_epoch_bump(ctx);
    table_detach(ctx);
    for (int slot = 0; slot < HALO_SLOTS; ++slot) workspace_release(ctx->wss[slot]);
}

// ------------------------------------------------------------------------------ driver
int msm_run(halo_ctx *ctx, const uint32_t *d_bases, const uint64_t *d_scalars, bool mont, size_t n, host::Point *out) {
    // a multi-device context: a large MSM over its own key goes to the shards (multi.hip); short ones are not worth the fan-out
    if (n >= ((size_t)1 << 16) && multi_takes(ctx, d_bases, n))
        return multi_run(ctx, (size_t)(d_bases - ctx->d_bases) / AFF_STRIDE, n, d_scalars, mont, out);
    BorrowScope scope(ctx);  // synchronous: a large MSM may alternate its pieces over slot 1's workspace (tmsm_enqueue_launches)
    int rc = msm_enqueue(ctx, 0, d_bases, d_scalars, mont, n);
    if (rc) return rc;
    return msm_finish(ctx, 0, out);
}



uint32_t msm_kmax(const halo_ctx *ctx, size_t n) {
    const int small_env = tuning().smsm_kmax;  // development override
    if (ctx->task_len > 0) return (uint32_t)ctx->task_len;
    if (small_env > 0 && n <= ((size_t)1 << 16)) return (uint32_t)small_env;
    const int late_env = tuning().late_kmax;  // development override
    if (n <= ((size_t)1 << 14)) return late_env > 0 ? (uint32_t)late_env : 8u;  // the IPA's late rounds: the chain is the round's latency (measured: 16 -> 8: -0.15 ms per open)
    return n >= ((size_t)1 << 18) ? KMAX : 16u;
}

// what a batch of `count` MSMs of n points needs beyond the slot's current capacity (0 = fits)
// window bits of a launch: the context's forced value, else the caller's hint for these scalars, else the size-based table
int launch_c(const halo_ctx *ctx, const MsmBatch &members) { return ctx->window_bits > 0 ? ctx->window_bits : members.c_hint; }
static bool batch_need(const halo_ctx *ctx, const MsmWorkspace &ws, size_t n, int count, WorkspaceNeed &need, int c_hint) {
    MsmPlan p = msm_plan(n, ctx->window_bits > 0 ? ctx->window_bits : c_hint);
    size_t Wt = (size_t)p.W * count, total = Wt * p.B, sorted = n * Wt;
    if (count > 1 && ctx->n < ((size_t)1 << 20) && ctx->n >= ((size_t)1 << 17)) {  // room for a batch through the small-key table plan
        size_t sets = 1;
        while (sets < (size_t)count) sets <<= 1;
        if (Wt < 32 * sets) Wt = 32 * sets;          // 2 x 16 window sums per bucket set
        if (sorted < n * 30 * (size_t)count) sorted = n * 30 * (size_t)count;  // 15 rows of u32 digits per member (d_canon holds 2 bytes per entry)
        if (total < sets << 16) total = sets << 16;
    }
    size_t hist = (Wt > 256 ? Wt : 256) * (size_t)p.B;
    size_t tasks = total + sorted / msm_kmax(ctx, n) + 1;
    bool grow = n > ws.cap_n || total > ws.cap_counts || sorted > ws.cap_sorted || hist > ws.cap_hist || Wt > ws.cap_windows ||
                tasks > ws.cap_tasks;
    need.n = n > ws.cap_n ? n : ws.cap_n;
    need.counts = total > ws.cap_counts ? total : ws.cap_counts;
    need.sorted = sorted > ws.cap_sorted ? sorted : ws.cap_sorted;
    need.hist = hist > ws.cap_hist ? hist : ws.cap_hist;
    need.windows = Wt > ws.cap_windows ? Wt : ws.cap_windows;
    need.tasks = tasks > ws.cap_tasks ? tasks : ws.cap_tasks;
    return grow;
}

int msm_enqueue(halo_ctx *ctx, int slot, const uint32_t *d_bases, const uint64_t *d_scalars, bool mont, size_t n) {
    MsmBatch one;
    one.count = 1;
    one.scalars[0] = d_scalars;
    return msm_enqueue_batch(ctx, slot, d_bases, one, mont, n);
}

int msm_enqueue_batch(halo_ctx *ctx, int slot, const uint32_t *d_bases, const MsmBatch &members, bool mont, size_t n) {
    if (slot < 0 || slot >= HALO_SLOTS) { set_error("msm: slot out of range"); return HALO_E_ARG; }
    if (members.count < 1 || members.count > MSM_MAX_BATCH) { set_error("msm: batch size must be in [1, 8]"); return HALO_E_ARG; }
    if (!ctx->wss[slot].d_counts) {
        int rc = msm_workspace_alloc(ctx, ctx->wss[0].cap_n, slot);
        if (rc) return rc;
    }
    MsmWorkspace &ws = ctx->wss[slot];
    if (ws.in_flight) { set_error("msm: slot already has an MSM in flight"); return HALO_E_ARG; }
    ws.plan = MsmPlan{0, 0, 0, msm_outputs(members), 0, 0};
    if (members.tagged && (members.count != 1 || mont || !msm_tagged_ready(ctx, d_bases, n))) {
        set_error("msm: a tagged launch is one canonical scalar array over the context's key with the c = 20 table in place (msm_tagged_ready)");
        return HALO_E_ARG;
    }
    if (members.parts < 1 || members.part < 0 || members.part >= members.parts) { set_error("msm: window shard out of range"); return HALO_E_ARG; }
    if (n == 0) { ws.in_flight = true; return HALO_OK; }
    if (members.parts > 1) {  // a shard that owns no window (more shards than windows) contributes the point at infinity
        MsmPlan p = msm_plan(n, launch_c(ctx, members));
        if (p.W * members.part / members.parts == p.W * (members.part + 1) / members.parts) { ws.in_flight = true; return HALO_OK; }
    }
    {
        // A batch lays the members' windows side by side and a forced task length multiplies the tasks: grow
        // this slot's workspace when the launch needs more room than a single automatic-plan MSM of the
        // context's size (the slot is idle here and its stream is drained).
        MsmPlan p = msm_plan(n, launch_c(ctx, members));
        if ((size_t)p.W * members.count * p.B > ((size_t)1 << 22)) {
            set_error("msm: batch too large for this window size (windows * batch * buckets <= 2^22)");
            return HALO_E_ARG;
        }
        WorkspaceNeed need;
        if ((members.count > 1 || ctx->task_len > 0) && batch_need(ctx, ws, n, members.count, need, members.c_hint)) {
            alloc_epoch_bump(ctx);
            workspace_release(ws);
            int rc = workspace_alloc(ws, need);
            if (rc) return rc;
        }
        if (members.tagged) {  // two sets of 2^19 buckets: per-bucket arrays of 2^20, a task per non-empty bucket and per kmax entries beyond
            const TblPlan tp = ctx->tbl;
            size_t counts = 2 * (size_t)tp.B, kmax = ctx->task_len > 0 ? (size_t)ctx->task_len : KMAX;
            size_t tasks = counts + (size_t)tp.W * n / kmax + 1;
            if (counts > ws.cap_counts || tasks > ws.cap_tasks) {
                need.n = ws.cap_n; need.sorted = ws.cap_sorted; need.hist = ws.cap_hist; need.windows = ws.cap_windows;
                need.counts = counts > ws.cap_counts ? counts : ws.cap_counts;
                need.tasks = tasks > ws.cap_tasks ? tasks : ws.cap_tasks;
                alloc_epoch_bump(ctx);
                workspace_release(ws);
                int rc = workspace_alloc(ws, need);
                if (rc) return rc;
            }
        }
    }
    StreamGuard guard(ctx, ctx->streams[slot]);
    if (!ctx->d_table && table_eligible(ctx, d_bases, members, n)) {
        int rc = table_build(ctx);
        if (rc) return rc;
    }
    // A table MSM of more than TBL_PIECE points runs as consecutive pieces.  Inside a synchronous call (msm_run, pcdl::check: the
    // caller has nothing else in flight) the pieces ALTERNATE over this slot's workspace and stream and the neighbouring slot's:
    // piece k + 1 sorts and accumulates while piece k's window sums -- latency chains on a few hundred waves -- finish.
    int partner = -1;
    if (ctx->may_borrow > 0 && members.count == 1 && ctx->d_table && ctx->tbl.c == 20 && n > TBL_PIECE && table_eligible(ctx, d_bases, members, n) &&
        tuning().piece_alternate) {
        int cand = slot ^ 1;
        if (!ctx->wss[cand].in_flight) {
            if (!ctx->wss[cand].d_counts || ctx->wss[cand].cap_n < ws.cap_n) {
                if (ctx->wss[cand].d_counts) { alloc_epoch_bump(ctx); workspace_release(ctx->wss[cand]); }
                if (msm_workspace_alloc(ctx, ws.cap_n, cand) == HALO_OK) partner = cand;
                else (void)hipGetLastError();  // no room for a second workspace: the pieces run one after the other, same result
            } else partner = cand;
        }
        for (int k = 0; k < 2 && partner >= 0; ++k)
            if (!ctx->ev_piece[slot][k] && hipEventCreateWithFlags(&ctx->ev_piece[slot][k], hipEventDisableTiming) != hipSuccess) partner = -1;
    }
    // The launch sequence below is fixed for a given (bases, scalars, n, form, window): the second
    // time the same key arrives it is captured into a hipGraph, afterwards one graph launch replaces
    // ~25 kernel launches (host launch cost matters for the small MSMs of the IPA rounds and of a
    // rank's share of a sharded MSM).  Event profiling needs the individual launches.
    MsmWorkspace::GraphKey key;
    key.bases = d_bases; key.members = members; key.n = n; key.mont = mont ? 1 : 0; key.c = ctx->window_bits; key.span = ctx->reduce_span + 1024 * ctx->task_len + 65536 * (ctx->sort_two_level + 1) + 262144 * (ctx->small_path + 1) + 1048576 * (ctx->table_mode + 1) + 4194304 * (partner + 1);
    bool graphs = ctx->use_graphs && !ctx->prof.on;
    // A graph is kept while this context has not allocated or freed device memory since it was instantiated (its own
    // workspaces, table, IPA buffers: first use only -- the opens of a loop allocate nothing, so their graphs survive), up to
    // MsmWorkspace::GRAPHS keys per slot.  A replay launches exactly the kernels, grids and arguments a fresh
    // enqueue with this key would.  The GPU memory fault of round 1 ("graph REPLAY ... n=262144") was a kernel with
    // SCRATCH inside a replayed graph (k_msm_reduce1: 256 VGPRs, 12 B/lane of spill) after the queue's scratch had been
    // re-assigned -- not a stale pointer: csrc/check_resources.py now fails the build if any kernel uses scratch.
    // (msm_wait returns when the last kernel has PUBLISHED, which is before the graph's last node has retired: the stream is
    // drained before one of its executable graphs is destroyed -- rare paths, both of them)
    auto drain = [&]() {
        (void)hipStreamSynchronize(ctx->streams[slot]);
        if (partner >= 0) (void)hipStreamSynchronize(ctx->streams[partner]);
    };
    if (ws.graph_epoch != ctx->alloc_epoch) {
        bool any = false;
        for (auto &g : ws.graphs) any = any || g.exec;
        if (any) drain();
        for (auto &g : ws.graphs) {
            if (g.exec) (void)hipGraphExecDestroy(g.exec);
            g = MsmWorkspace::CachedGraph();
        }
        ws.graph_epoch = ctx->alloc_epoch;
    }
    const int cache_n = tuning().graph_cache;  // development switch: HALO_GRAPH_CACHE=1 is the single graph per slot of rounds 1-3
    static_assert(MsmWorkspace::GRAPHS == 8, "tuning.hip clamps HALO_GRAPH_CACHE to 8");
    MsmWorkspace::CachedGraph *hit = nullptr;
    for (int k = 0; k < cache_n; ++k)
        if (ws.graphs[k].exec && key == ws.graphs[k].key) hit = &ws.graphs[k];
    if (graphs && hit) {
        if (debug_trace()) fprintf(stderr, "[halo] graph REPLAY ctx=%p slot=%d n=%zu\n", (void *)ctx, slot, n);
        HALO_HIP(hipGraphLaunch(hit->exec, ctx->streams[slot]));
        hit->used = ++ws.graph_clock;
        ws.plan = hit->plan;
        ws.done_expect += (uint32_t)ws.plan.publishers;
        ws.in_flight = true;
        if (partner >= 0) { ws.borrowed = partner; ctx->wss[partner].in_flight = true; ctx->wss[partner].lent_from = slot; }
        return HALO_OK;
    }
    bool capture = false;
    if (graphs) {
        for (int k = 0; k < cache_n; ++k) capture = capture || key == ws.seen[k];
        if (!capture) { ws.seen[ws.seen_at] = key; ws.seen_at = (ws.seen_at + 1) % cache_n; }
    }
    if (debug_trace()) fprintf(stderr, "[halo] msm enqueue ctx=%p slot=%d n=%zu batch=%d part=%d/%d capture=%d\n", (void *)ctx, slot, n, members.count, members.part, members.parts, (int)capture);
    if (capture) HALO_HIP(hipStreamBeginCapture(ctx->streams[slot], hipStreamCaptureModeRelaxed));
    // The launch's last kernel publishes its sums itself (smsm.hip publish(): pinned buffer + pinned counter) unless the event
    // profiler brackets every launch or HALO_DIRECT_RESULTS=0 asks for the copy + stream wait of rounds 1-3 (development switch).
    const bool direct = tuning().direct_results;
    ctx->sink_done = direct && !ctx->prof.on ? ws.h_done : nullptr;
    ctx->sink_publishers = 0;
    int rc = msm_enqueue_launches(ctx, ws, d_bases, members, mont, n, partner);
    ctx->sink_done = nullptr;
    if (!rc) ws.plan.publishers = ctx->sink_publishers;
    if (capture) {
        hipGraph_t graph = nullptr;
        hipError_t e = hipStreamEndCapture(ctx->streams[slot], &graph);
        if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        if (e != hipSuccess) return hip_fail(e, "hipStreamEndCapture");
        MsmWorkspace::CachedGraph *victim = &ws.graphs[0];
        for (int k = 0; k < cache_n; ++k)
            if (!ws.graphs[k].exec) { victim = &ws.graphs[k]; break; }
            else if (ws.graphs[k].used < victim->used) victim = &ws.graphs[k];
        if (victim->exec) { drain(); (void)hipGraphExecDestroy(victim->exec); *victim = MsmWorkspace::CachedGraph(); }
        e = hipGraphInstantiate(&victim->exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess) { victim->exec = nullptr; return hip_fail(e, "hipGraphInstantiate"); }
        victim->key = key;
        victim->plan = ws.plan;
        victim->used = ++ws.graph_clock;
        HALO_HIP(hipGraphLaunch(victim->exec, ctx->streams[slot]));
    }
    if (rc) {
        // part of the sequence may be on its way and will publish: the counter is brought back in step before anyone waits on it
        (void)hipStreamSynchronize(ctx->streams[slot]);
        if (partner >= 0) (void)hipStreamSynchronize(ctx->streams[partner]);
        (void)hipGetLastError();
        ws.done_expect = *(volatile uint32_t *)ws.h_done;
        return rc;
    }
    ws.done_expect += (uint32_t)ws.plan.publishers;
    ws.in_flight = true;
    if (partner >= 0) { ws.borrowed = partner; ctx->wss[partner].in_flight = true; ctx->wss[partner].lent_from = slot; }
    return HALO_OK;
}

int msm_finish(halo_ctx *ctx, int slot, host::Point *out) { return msm_finish_batch(ctx, slot, out, 1); }

int msm_wait(halo_ctx *ctx, int slot, int count) {
    if (slot < 0 || slot >= HALO_SLOTS || !ctx->wss[slot].in_flight) { set_error("msm: nothing in flight on this slot"); return HALO_E_ARG; }
    MsmWorkspace &ws = ctx->wss[slot];
    if (ws.plan.batch != count) { set_error("msm: this slot holds a batch of a different size"); return HALO_E_ARG; }
    ws.in_flight = false;
    if (ws.borrowed >= 0) {  // (the neighbour's launches were joined into this slot's stream: the wait below covers them)
        ctx->wss[ws.borrowed].in_flight = false;
        ctx->wss[ws.borrowed].lent_from = -1;
        ws.borrowed = -1;
    }
    if (ws.plan.W == 0) return HALO_OK;  // n == 0, or a window shard without windows
    if (ws.plan.publishers > 0) {
        // the last kernel's blocks count themselves in as they hand over their sums (smsm.hip publish()): poll the pinned counter
        // instead of waiting for the stream -- and look at the stream now and then, in case a launch has died
        // Waiting costs a core only briefly: the first tuning().spin_us microseconds (50) poll with the spin-wait hint -- the late
        // rounds of an open end within that --, then every poll is followed by sched_yield() (free when nobody else wants the
        // core, and a host running many provers shares it), and a launch that outlasts 2 ms (n >= 2^21, a shard's stretch) sleeps
        // 50 us between polls.
        volatile uint32_t *f = ws.h_done;
        const auto t_begin = std::chrono::steady_clock::now();
        const long spin_us = tuning().spin_us;
        int phase = 0;  // 0 spin, 1 yield, 2 sleep
        for (uint32_t spins = 1;; ++spins) {
            if ((int32_t)(*f - ws.done_expect) >= 0) break;
            if ((spins & (phase == 0 ? 0x3fffu : 0xffu)) == 0 || phase == 2) {
                hipError_t e = hipStreamQuery(ctx->streams[slot]);
                if (e == hipSuccess) {
                    if ((int32_t)(*f - ws.done_expect) >= 0) break;
                    ws.done_expect = *f;
                    set_error("msm: the launch sequence ended without publishing its results");
                    return HALO_E_DEVICE;
                }
                if (e != hipErrorNotReady) return hip_fail(e, "hipStreamQuery");
            }
            if (phase == 0) {
                host::cpu_relax();
                if ((spins & 0x3fu) == 0 &&
                    std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_begin).count() >= spin_us) phase = 1;
            } else if (phase == 1) {
                std::this_thread::yield();
                if ((spins & 0x3fu) == 0 &&
                    std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_begin).count() >= 2000) phase = 2;
            } else {
                std::this_thread::sleep_for(std::chrono::microseconds(50));
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        if (ctx->prof.on) HALO_HIP(hipStreamSynchronize(ctx->streams[slot]));
    } else {
        HALO_HIP(hipStreamSynchronize(ctx->streams[slot]));
    }
    bool others = false;
    for (int k = 0; k < HALO_SLOTS; ++k) others = others || ctx->wss[k].in_flight;
    if (ctx->prof.on && !others) ctx->prof.collect();
    return HALO_OK;
}
// one member of a batched launch (any thread: reads the slot's pinned window sums only)
void msm_combine_member(halo_ctx *ctx, int slot, int b, host::Point *out) {
    const MsmWorkspace &ws = ctx->wss[slot];
    MsmPlan p = ws.plan;
    *out = host::Point::infinity();
    if (p.W == 0) return;
    if (p.table_vw > 0 && p.table_rc) {
        // per piece and set: pairs (S_q, T_q) = (sum E, sum (Q + 1) E) over 64 entries each, the columns' first, then the rows'.
        //   sum_lo (lo + 1) C_lo = sum_q T_q + 64 sum_q q S_q;   sum_hi (hi + 1) R_hi likewise
        //   sum_b (b + 1) B_b = [columns] + 2^lg_cols ([rows] - sum_hi R_hi)
        const int cblocks = (1 << p.table_rc_lg_cols) / 64, rblocks = (1 << p.table_rc_lg_rows) / 64;
        const size_t set_pts = 2 * (size_t)(cblocks + rblocks), piece_pts = set_pts * (size_t)p.table_sets;
        auto weighted = [&](int first, int count, host::Point *plain) {
            host::Point t = host::Point::infinity(), run = host::Point::infinity(), tot = host::Point::infinity();
            for (int q = count - 1; q >= 0; --q) {
                host::Point sq = host::Point::infinity();
                for (int k = 0; k < p.table_pieces; ++k) {
                    const uint64_t *w = ws.h_winsum + 12 * ((size_t)k * piece_pts + (size_t)b * set_pts + 2 * (size_t)(first + q));
                    sq = sq + host::Point::load(w);
                    t = t + host::Point::load(w + 12);
                }
                run = run + sq;
                if (q >= 1) tot = tot + run;  // tot = sum_q q S_q by running sums
            }
            for (int k = 0; k < 6 && !tot.is_inf(); ++k) tot = tot.dbl();
            if (plain) *plain = run;
            return t + tot;
        };
        host::Point s_rows;
        host::Point cols = weighted(0, cblocks, nullptr);
        host::Point rows = weighted(cblocks, rblocks, &s_rows) - s_rows;
        for (int k = 0; k < p.table_rc_lg_cols && !rows.is_inf(); ++k) rows = rows.dbl();
        *out = cols + rows;
        return;
    }
    if (p.table_vw > 0) {
        // virtual window v holds the buckets v 2^b + 1 .. (v + 1) 2^b (b = table_vw_bits): sum_v [ T_v + v 2^b S_v ];
        // a piece (large MSM) or a batch stores sets * V weighted sums, then sets * V plain sums; member b owns set b
        int V = p.table_vw, A = p.table_sets * V;
        host::Point acc = host::Point::infinity(), run = host::Point::infinity(), tot = host::Point::infinity();
        for (int k = 0; k < p.table_pieces; ++k)  // the pieces of a large MSM add up window by window
            for (int v = 0; v < V; ++v) acc = acc + host::Point::load(ws.h_winsum + 12 * ((size_t)k * 2 * A + (size_t)b * V + v));
        for (int v = V - 1; v >= 1; --v) {  // tot = sum_v v S_v by running sums
            for (int k = 0; k < p.table_pieces; ++k) run = run + host::Point::load(ws.h_winsum + 12 * ((size_t)k * 2 * A + A + (size_t)b * V + v));
            tot = tot + run;
        }
        for (int k = 0; k < p.table_vw_bits && !tot.is_inf(); ++k) tot = tot.dbl();
        *out = acc + tot;
        return;
    }
    int Wm = p.w1 - p.w0;
    host::Point acc = host::Point::infinity();
    for (int w = Wm - 1; w >= 0; --w) {
        if (!acc.is_inf())
            for (int k = 0; k < p.c; ++k) acc = acc.dbl();
        acc = acc + host::Point::load(ws.h_winsum + 12 * ((size_t)b * Wm + w));
    }
    if (!acc.is_inf())
        for (int k = 0; k < p.c * p.w0; ++k) acc = acc.dbl();  // a window shard's weight 2^(c * w0)
    *out = acc;
}
// Horner over the window sums the slot's last launch left in pinned memory (call after msm_wait)
void msm_combine(halo_ctx *ctx, int slot, host::Point *out, int count) {
    const MsmWorkspace &ws = ctx->wss[slot];
    MsmPlan p = ws.plan;
    for (int b = 0; b < count; ++b) out[b] = host::Point::infinity();
    if (p.W == 0) return;
    for (int b = 0; b < count; ++b) msm_combine_member(ctx, slot, b, &out[b]);
}
int msm_finish_batch(halo_ctx *ctx, int slot, host::Point *out, int count) {
    for (int b = 0; b < count; ++b) out[b] = host::Point::infinity();
    int rc = msm_wait(ctx, slot, count);
    if (rc) return rc;
    msm_combine(ctx, slot, out, count);
    return HALO_OK;
}

}  // namespace halo
