// Internal declarations shared by the translation units of libhalo_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/halo_accumulation.h"
#include "host_math.hpp"
#include "tuning.hpp"

namespace halo {

void set_error(const std::string &msg);
int hip_fail(hipError_t e, const char *what);

#define HALO_HIP(expr)                                   \
    do {                                                 \
        hipError_t _e = (expr);                          \
        if (_e != hipSuccess) return hip_fail(_e, #expr); \
    } while (0)

struct ProfEntry {
    const char *name;
    double total_ms = 0;
    long launches = 0;
};

struct Profiler {
    bool on = false;
    bool dominant_only = false;  // bracket only k_msm_accumulate / k_fold_points (2 events per MSM)
    std::vector<ProfEntry> entries;
    struct Pending { int idx; hipEvent_t a, b; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> pool;
    int find(const char *name);
    void begin(const char *name, hipStream_t s);
    void end(hipStream_t s);
    void collect();  // needs the stream to be idle
};

// k_msm_accumulate, k_smsm_accumulate, k_fold_points, k_fold_points4
inline bool prof_is_dominant(const char *name) {
    return name[2] == 'm' ? name[6] == 'a' : name[2] == 's' ? name[7] == 'a' : (name[2] == 'f' && name[7] == 'p');
}
// Launch wrapper: brackets the launch with events when profiling is on.
#define HALO_LAUNCH(ctx, name, kernel, grid, block, shmem, ...)                              \
    do {                                                                                     \
        bool _p = (ctx)->prof.on && (!(ctx)->prof.dominant_only || halo::prof_is_dominant(name)); \
        if (_p) (ctx)->prof.begin(name, (ctx)->stream);                                      \
        if (halo::debug_trace()) fprintf(stderr, "[halo] launch %s ctx=%p stream=%p\n", name, (void *)(ctx), (void *)(ctx)->stream); \
        hipLaunchKernelGGL(kernel, grid, block, shmem, (ctx)->stream, __VA_ARGS__);          \
        if (_p) (ctx)->prof.end((ctx)->stream);                                              \
    } while (0)

struct MsmPlan {
    int c;        // window bits
    int W;        // windows = ceil(256 / c)
    uint32_t B;   // buckets per window = 2^(c-1)
    int batch;    // MSMs sharing the launch sequence (their windows are laid side by side: W * batch in all)
    int w0, w1;   // windows [w0, w1) of 0..W are computed (a window-sharded partial); the result carries 2^(c*w0)
    int table_vw = 0;  // > 0: the fixed-base table pipeline ran; h_winsum holds table_vw weighted sums, then table_vw plain sums
    int table_vw_bits = 15;  // log2 of the buckets per virtual window
    int table_pieces = 1;    // a large table MSM runs as this many consecutive pieces: 2 * table_sets * table_vw sums each
    int table_sets = 1;      // bucket sets side by side in a batched table launch (members, rounded up to a power of two)
    bool table_rc = false;   // window sums by rows and columns of the bucket index (msm.hip k_msm_reduce_rc): (S, T) pairs per set
    int table_rc_lg_rows = 0, table_rc_lg_cols = 0;
    // > 0: the launch's last kernel(s) write their sums straight into the slot's pinned buffer and each adds one to the slot's
    // pinned counter when its last block is through (MsmWorkspace::h_done; one per piece); msm_wait polls the counter.  0: a copy
    // and a stream wait.
    int publishers = 0;
};
constexpr int MSM_MAX_BATCH = 8;
// fixed-base table plan of a context (msm.hip: table_plan)
struct TblPlan {
    int c, W, fbits, vw_bits;        // window bits, windows, fine-bucket bits of a coarse range, log2 buckets per virtual window
    uint32_t B, ranges, vw, spread;  // buckets (2^(c-1)), coarse ranges, virtual windows, modulus of the top-window spread (0: none)
    int fold_top;                    // scalars >= 2^254 are recoded as r - s with flipped signs (c = 17: the top window is full)
};
TblPlan table_plan(size_t key_n);
// the members of a batched launch: same n, one scalar array and one base offset (in points) each
struct MsmBatch {
    int count = 1;
    const uint64_t *scalars[MSM_MAX_BATCH] = {};
    uint32_t base_off[MSM_MAX_BATCH] = {};
    int part = 0, parts = 1;  // window shard: this launch computes windows [part*W/parts, (part+1)*W/parts)
    int c_hint = 0;           // window bits the caller knows to be better for THIS launch's scalars (0: the size-based table); halo_set_window_bits wins
    // TWO sums from ONE scalar array (count = 1, canonical form, c = 20 table plan only: msm_tagged_ready): bit 255 of scalar i --
    // free, r < 2^255 -- says which of two bucket sets point i goes to.  One recode / sort / bucket kernel / window-sum pass for
    // both; results as for a batch of two (msm_wait(.., 2), msm_combine_member 0 / 1).  The IPA's L and R over the full key.
    bool tagged = false;
    // A stretch of a LARGER sum over the context's key (halo_msm with host scalars: the stretches run on different slots, each
    // behind the copy of its own scalars): takes the c = 20 table plan although n is below its usual 2^20 points.
    bool sub = false;
};
inline int msm_outputs(const MsmBatch &m) { return m.tagged ? 2 : m.count; }
MsmPlan msm_plan(size_t n, int forced_c);
uint32_t msm_spread(const MsmPlan &p, uint32_t *top_bit);  // top-window spread of the recode (0: none)

struct MsmWorkspace {
    size_t cap_n = 0;
    uint64_t *d_canon = nullptr;     // u16 signed digits, layout [w][i]
    uint32_t *d_hist = nullptr;      // per (window, chunk) bucket histograms / prefixes
    uint32_t *d_counts = nullptr;    // W*B
    uint32_t *d_starts = nullptr;    // W*B exclusive scan within 4096-entry blocks
    uint32_t *d_blockoff = nullptr;  // per 4096-entry block offset
    uint32_t *d_sorted = nullptr;    // n*W entries: point index | sign << 31
    uint32_t *d_presort = nullptr;   // the same entries grouped by bucket range only (two-level sort of large MSMs)
    uint32_t *d_buckets = nullptr;   // one native XYZZ partial (40 words) per task
    uint32_t *d_ntask = nullptr, *d_toff = nullptr, *d_tblockoff = nullptr, *d_biglist = nullptr, *d_meta = nullptr;
    uint32_t *d_task_g = nullptr, *d_order = nullptr;  // per task: bucket | length bin << 24; task records (4 words) by decreasing length
    uint16_t *d_fine16 = nullptr;    // table pipeline: low bucket bits beside d_presort (allocated on first use)
    uint32_t *d_seg = nullptr;       // W*64 x 2 native XYZZ (S, T per 512-bucket segment)
    uint64_t *d_winsum = nullptr;    // W x 12 (Jacobian)
    uint64_t *h_winsum = nullptr;    // pinned
    uint32_t *h_done = nullptr;      // pinned: blocks that have published their sums, ever (see MsmPlan::publishers)
    uint32_t done_expect = 0;        // ... and what it reads once the launches enqueued so far are done
    size_t cap_counts = 0, cap_sorted = 0, cap_tasks = 0, cap_hist = 0, cap_windows = 0;
    MsmPlan plan{};          // plan of the MSM in flight on this slot
    bool in_flight = false;
    int lent_from = -1;      // this slot's workspace and stream run the odd pieces of the large MSM in flight on slot `lent_from`
    int borrowed = -1;       // ... and that slot remembers which one it borrowed (msm_wait gives it back)
    // hipGraph of the launch sequence, replayed while the same (bases, scalars, n, form, window) repeats
    struct GraphKey {
        const void *bases = nullptr;
        MsmBatch members;
        size_t n = 0;
        int mont = 0, c = 0, span = 0;
        bool operator==(const GraphKey &o) const {
            if (!(bases == o.bases && n == o.n && mont == o.mont && c == o.c && span == o.span && members.count == o.members.count &&
                  members.part == o.members.part && members.parts == o.members.parts && members.c_hint == o.members.c_hint && members.tagged == o.members.tagged && members.sub == o.members.sub))
                return false;
            for (int b = 0; b < members.count; ++b)
                if (members.scalars[b] != o.members.scalars[b] || members.base_off[b] != o.members.base_off[b]) return false;
            return true;
        }
    };
    // Launch graphs of this slot, by key: an open's rounds come back with the same few keys open after open (the tagged launch of
    // rounds 0-1, the batches over the 2^18-, 2^16- and 2^14-point keys, the check's MSM), so the slot keeps several -- with ONE
    // graph a slot re-captured and re-instantiated four of them per open and launched the other four rounds kernel by kernel.
    static constexpr int GRAPHS = 8, SEEN = 8;
    struct CachedGraph { GraphKey key; hipGraphExec_t exec = nullptr; MsmPlan plan{}; uint64_t used = 0; };
    CachedGraph graphs[GRAPHS];
    GraphKey seen[SEEN];       // keys that have arrived once (plain launches): the second arrival is captured
    int seen_at = 0;
    uint64_t graph_epoch = 0;  // the context's alloc_epoch the graphs were instantiated under (all go when it moves)
    uint64_t graph_clock = 0;  // least recently used goes first
};

}  // namespace halo

constexpr int HALO_SLOTS = 4;

namespace halo {
// One helper thread per context for pure host arithmetic that would otherwise serialise on the caller's thread
// (the window combine of the second MSM of an IPA round while the caller combines the first); a shard of a multi-device
// context also runs its own HIP calls on it (multi.hip: copies, launches and the stream wait of its stretch of an MSM, on
// its own device).  While an IPA state of the context is alive (`hot` counts them) a round arrives every few hundred
// microseconds and a condition-variable wake-up would cost a good part of what the overlap saves, so after a job
// the thread polls for the next one -- but only for SPIN_US: an idle state (a caller thinking between rounds, a long
// sharded open waiting on a collective) does not hold a core; the thread falls back to the condition variable.
class HostWorker {
   public:
    ~HostWorker() { stop(); }
    void submit(std::function<void()> fn) {
        start();
        job_ = std::move(fn);
        done_.store(false, std::memory_order_relaxed);
        { std::lock_guard<std::mutex> lk(mu_); pending_.store(true, std::memory_order_release); }
        cv_.notify_one();
    }
    // A job of pure host arithmetic ends within tens of microseconds: poll for it.  A job that sits in hipStreamSynchronize
    // for the length of an MSM (multi.hip: a shard's stretch) would make the caller burn a core per shard for milliseconds:
    // after SPIN_US of polling the caller sleeps on a condition variable instead.
    void wait() {
        if (!thread_.joinable()) return;
        auto t0 = std::chrono::steady_clock::now();
        while (!done_.load(std::memory_order_acquire)) {
            if (std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() < SPIN_US) {
                std::this_thread::yield();
                continue;
            }
            std::unique_lock<std::mutex> lk(mu_);
            done_cv_.wait(lk, [this] { return done_.load(std::memory_order_acquire); });
        }
    }
    // one call per IPA state created (+1) / destroyed (-1)
    void add_hot(int delta) {
        int now = hot_.fetch_add(delta, std::memory_order_relaxed) + delta;
        if (delta > 0 && now > 0) start();
    }
    void stop() {
        if (!thread_.joinable()) return;
        { std::lock_guard<std::mutex> lk(mu_); quit_.store(true); }
        cv_.notify_one();
        thread_.join();
    }

   private:
    static constexpr long SPIN_US = 600;
    void start() {
        if (thread_.joinable()) return;
        done_.store(true);
        thread_ = std::thread([this] { loop(); });
    }
    void loop() {
        auto last = std::chrono::steady_clock::now();
        for (;;) {
            if (!pending_.load(std::memory_order_acquire)) {
                if (quit_.load()) return;
                if (hot_.load(std::memory_order_relaxed) > 0 &&
                    std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - last).count() < SPIN_US) {
                    std::this_thread::yield();
                    continue;
                }
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [this] { return pending_.load() || quit_.load(); });
                continue;
            }
            pending_.store(false, std::memory_order_relaxed);
            job_();
            { std::lock_guard<std::mutex> lk(mu_); done_.store(true, std::memory_order_release); }  // (under the lock: a waiter about to sleep cannot miss it)
            done_cv_.notify_all();
            last = std::chrono::steady_clock::now();
        }
    }
    std::thread thread_;
    std::mutex mu_;
    std::condition_variable cv_, done_cv_;
    std::function<void()> job_;
    std::atomic<bool> pending_{false}, done_{true}, quit_{false};
    std::atomic<int> hot_{0};
};
}  // namespace halo

// Device buffers of one pcdl::open (pcdl.rs:183-186 state + the no-fold vectors), owned by the context and reused
// from open to open: the open path performs no device allocation after the first call.
struct IpaBuffers {
    uint32_t *d_G = nullptr;                                       // cap_n x 32 words (native affine, 128-B stride)
    uint64_t *d_c = nullptr, *d_z = nullptr;                       // cap_n x 4
    uint64_t *d_s = nullptr, *d_s2 = nullptr, *d_FL = nullptr, *d_FR = nullptr;  // cap_M x 4
    uint64_t *d_pbar = nullptr;                                    // cap_n x 4, hiding branch of a sharded open (lazy)
    size_t cap_n = 0, cap_M = 0;
    bool in_use = false;
};

// What the contexts over ONE resident key have in common (halo_ctx_clone): the key, its fixed-base MSM table and its fold
// table -- immutable once built -- and the optional-memory bytes they hold on the device's books.  Every context owns a
// reference (a context that was never cloned is the only user of its own); the last one to go frees the memory.  `mu` guards
// the fields; a table is built by ONE context at a time (`*_busy`), the others run table-free meanwhile and adopt it at their
// next opportunity.
struct KeyShare {
    std::mutex mu;
    int users = 1;
    uint32_t *d_bases = nullptr;
    uint32_t *d_table = nullptr;
    halo::TblPlan tbl{};
    uint32_t *d_foldtab = nullptr;
    size_t foldtab_bytes = 0;
    double foldtab_build_ms = 0;
    bool table_busy = false, foldtab_busy = false;
    long full_opens = 0;     // full-size opens over this key by any of its contexts while no fold table existed (the automatic mode builds from tuning().fold_table_after on)
    size_t budget_held = 0;  // bytes of optional memory reserved for this key (abi.hip table_budget_*)
};

struct halo_ctx {
    int device = 0;
    std::shared_ptr<KeyShare> share;  // d_bases / d_table / d_foldtab below are this context's view of it
    hipStream_t stream = nullptr;      // stream the launch macro uses (= streams[slot in use])
    hipStream_t streams[HALO_SLOTS] = {};
    size_t n = 0;
    uint32_t *d_bases = nullptr;  // n x 32 words (AFF_STRIDE): native affine x | y | -y, one 128-byte line per point (curve.hpp AffN)
    uint32_t *sink_done = nullptr;             // while a launch is being enqueued: the counter its last kernel publishes to (null: copy + stream wait)
    int sink_publishers = 0;                   // ... and how many blocks will
    halo::MsmWorkspace wss[HALO_SLOTS];        // slots (workspace + stream) so that independent MSMs can overlap
    halo::Profiler prof;
    std::vector<halo::ProfEntry> prof_merged;  // what halo_prof_count / _get show: this context's entries plus its shards'
    int window_bits = 0;
    int reduce_span = 0;                   // buckets per lane in k_msm_reduce1 (0 = automatic)
    int sort_two_level = -1;               // two-level sort: -1 automatic (n >= 2^17), 0 never, 1 whenever the shape allows
    int task_len = 0;                      // longest chain per lane in k_msm_accumulate (0 = automatic)
    int table_mode = -1;                   // fixed-base tables for MSMs over the context's own bases: -1 automatic (n >= 2^20), 0 never
    uint32_t *d_table = nullptr;           // W x n native affine points: T[w][i] = 2^(c w) G_i (built on first use)
    halo::TblPlan tbl{};                   // the plan d_table was built for
    int small_path = -1;                   // smsm.hip pipeline: -1 automatic (n <= 2^16, one MSM per launch), 0 never
    bool use_graphs = true;                // replay cached hipGraphs for repeated MSM shapes
    size_t nofold_size = (size_t)1 << 14;  // key size at which the IPA stops folding G (0/1 = never)
    bool batch_verify = true;              // succinct checks of >= 64 instances in two device launches (else a host thread pool)
    int fold_table_mode = -1;              // comb table for the first fold of an open (foldtab.hip): -1 after tuning().fold_table_after (8) full-size opens over the key -- memory requested on a helper thread, built at the first later open that finds it; 1 at the next open; 0 never
    uint32_t *d_foldtab = nullptr;         // E[w][d][i - n/4] = d 64^w G_i, 64-byte affine entries
    size_t foldtab_bytes = 0;
    double foldtab_build_ms = 0;
    int foldtab_opens = 0;                 // full-size opens seen while the table did not exist
    // A table that could not be had (budget, allocation) is tried again later, with a doubling back-off, instead of never:
    // memory comes back when other contexts go.  status: 0 nothing yet, 1 memory requested, 2 built, 3 over the
    // memory budget, 4 allocation or build failed, 5 off (halo_ctx_info 5 / 6)
    int foldtab_retry_at = 0, foldtab_backoff = 8, foldtab_status = 0;
    long table_calls = 0, table_retry_at = 0, table_backoff = 64;
    int table_status = 0;
    bool table_said = false, foldtab_said = false;  // the one line on stderr has been printed
    // (the optional-memory bytes are on the books of the key: share->budget_held)
    // automatic mode: the table's 40 GB are requested on a helper thread at the first full-size open (hipMalloc of that size
    // takes 0.5 ms .. 2 s depending on what the driver has at hand) and the table is built at the first later open that
    // finds them there.  state: 0 nothing, 1 running, 2 ready, 3 failed
    std::thread foldtab_alloc_thread;
    std::atomic<int> foldtab_alloc_state{0};
    uint32_t *foldtab_pending = nullptr, *foldtab_pending_tmp = nullptr;
    int fold_levels = 2;                   // halving rounds folded into G at a time (1: every round; 2: every other round, k_fold_points4)
    int fold_async = -1;                   // folds of keys of <= 2^18 points beside the next two rounds: -1 in opens of <= 2^18 points (measured), 0 never, 1 always (halo_set_fold_async)
    // scratch for host-pointer entry points
    uint64_t *d_tmp_a = nullptr, *d_tmp_b = nullptr, *d_tmp_c = nullptr;
    size_t tmp_words = 0;
    uint64_t *h_pinned = nullptr;  // small pinned staging (4 KiB)
    uint64_t *h_wintab = nullptr;  // pinned staging of a window table of powers (ipa.hip upload_window_table), 8 KiB
    uint64_t *d_batch_scalars[HALO_SLOTS] = {};  // a shard's peer copies of a batch's scalar arrays (multi.hip), grown on demand
    size_t batch_scalars_bytes[HALO_SLOTS] = {};
    uint64_t *d_slot_scalars[HALO_SLOTS] = {};  // host-scalar MSMs (halo_msm, halo_msm_begin): one staging buffer of n x 4 words per slot, first use
    uint64_t *d_verify = nullptr;  // staging of the batched verifier (points, scalars, challenges, results), grown on demand
    size_t verify_words = 0;
    // lazily allocated n x 4 polynomial buffers for pcdl::open / acc::prover
    uint64_t *d_poly = nullptr, *d_poly2 = nullptr;
    halo::HostWorker worker;      // host arithmetic overlapped with the caller's (see HostWorker; multi.hip also runs a shard's HIP calls on it)
    IpaBuffers ipa_bufs;          // reused by every halo_ipa of this context (one at a time; a second one allocates its own)
    uint64_t alloc_epoch = 0;     // bumped whenever this context allocates or frees device memory (see msm.hip, launch graphs)
    int may_borrow = 0;           // > 0 inside a synchronous MSM call: a large MSM may run its odd pieces on the neighbouring slot (msm.hip)
    hipEvent_t ev_piece[HALO_SLOTS][2] = {};  // fork / join of those pieces (created on first use)
    // multi-device contexts (multi.hip): one shard context per device over its index block of the key; MSMs over the key fan out
    std::vector<halo_ctx *> shards;
    std::vector<size_t> shard_lo;  // shards.size() + 1 block boundaries
    halo_ctx *parent = nullptr;    // set on a shard
    struct Fan { bool active = false; int batch = 0; std::vector<char> used; };  // batch: members of a batched launch (0: a single MSM)
    Fan fan[HALO_SLOTS];           // which shards hold a stretch of the MSM in flight on each slot
};

struct halo_ipa {
    halo_ctx *ctx = nullptr;
    size_t n = 0, m = 0;  // m = current length (n, n/2, ...)
    uint32_t *d_G = nullptr;  // m x 20 words native affine (in-place)
    const uint32_t *G_src = nullptr;  // where the current key is read from: the context's own table of bases until the first real
                                      // fold (no copy, and its MSMs may use the context's fixed-base table), d_G afterwards
    uint64_t *d_c = nullptr;  // m x 4
    uint64_t *d_z = nullptr;  // m x 4
    halo::host::FixedBaseTable hp_table;  // window table of the H' this open uses (pcdl.rs:181)
    bool hp_from_scalar = false;          // H' = hp_scalar * H: terms k * H' come from the process-wide table of H as (k * hp_scalar) * H
    halo::host::Fr hp_scalar;
    hipEvent_t ev = nullptr;  // orders slot 1's stream after the folds queued on stream 0
    // no-fold mode (ipa.hip): G stays at M points, s holds the challenge products
    bool nofold = false;
    bool deferred = false;                   // the no-fold phase ends in a two-level fold once s_len reaches 4
    std::vector<halo::host::Fr> s_host;      // host copy of s while deferred (<= 4 entries)
    size_t M = 0, s_len = 0;
    uint64_t *d_s = nullptr, *d_s2 = nullptr, *d_FL = nullptr, *d_FR = nullptr;  // M x 4 each
    uint64_t *d_pbar = nullptr;  // n x 4: this shard of p_bar (hiding branch of the sharded open)
    bool pbar_valid = false;     // halo_ipa_hiding_partial has filled d_pbar for this state
    // The LAST round of a no-fold phase already holds U (abi.hip halo_ipa_finish): with two coefficients c0, c1 left, its MSMs are
    // L' = c1 A and R' = c0 B over the even / odd points of the key, and U = G_final[0] = A + xi B.  Kept here: L', R' before
    // their H' terms, c0, c1 (copied to the host with the round's results) and the round's challenge.
    bool last_valid = false, last_folded = false;
    halo::host::Point last_L, last_R;
    halo::host::Fr last_c0, last_c1, last_xi, last_xi_inv;
    // A two-level fold of a key of at most 2^18 points is a latency chain on one wave per SIMD (1.6 and 1.1 ms in a 2^20 open):
    // it runs on the context's fourth stream BESIDE the next two rounds, which keep reading the unfolded key (abi.hip
    // ipa_round_fold_impl).  fold_pending: launched, not yet switched to; fold_dst / fold_m: its output; tail_host: the products
    // of the challenges hashed since (they are the first entries of d_s and the scalars of the fold after this one);
    // g_off: points of d_G already taken by earlier keys (the folds no longer run in place).
    bool fold_pending = false;
    uint32_t *fold_dst = nullptr;
    size_t fold_m = 0, g_off = 0;
    std::vector<halo::host::Fr> tail_host;
    hipEvent_t ev_fold = nullptr;
    bool counted_hot = false;    // this state is counted in ctx->worker's hot count (undone by halo_ipa_destroy)
    bool borrowed = false;       // buffers belong to ctx->ipa_bufs (returned, not freed, by halo_ipa_destroy)
};

namespace halo {

// ---- msm.hip
// scope of a synchronous MSM call (nothing else of the caller's can be in flight on the context's other slots meanwhile)
struct BorrowScope {
    halo_ctx *c;
    explicit BorrowScope(halo_ctx *ctx) : c(ctx) { c->may_borrow++; }
    ~BorrowScope() { c->may_borrow--; }
};
// a context's count of its own device (de)allocations
inline void alloc_epoch_bump(halo_ctx *ctx) { ctx->alloc_epoch++; }
int msm_workspace_alloc(halo_ctx *ctx, size_t n, int slot);
void msm_workspace_free(halo_ctx *ctx);
int table_release(halo_ctx *ctx);  // gives up the context's fixed-base table (every slot idle): freed unless a clone still uses it
void table_detach(halo_ctx *ctx);   // the same without the idle check (context teardown)
// asynchronous halves of msm_run on workspace/stream `slot`
int msm_enqueue(halo_ctx *ctx, int slot, const uint32_t *d_bases, const uint64_t *d_scalars, bool scalars_mont, size_t n);
int msm_finish(halo_ctx *ctx, int slot, host::Point *out);
// the two halves of msm_finish: wait for the slot's launches (HIP), then combine the window sums (pure host arithmetic,
// may run on another thread); batch as msm_finish_batch
int msm_wait(halo_ctx *ctx, int slot, int count);
void msm_combine(halo_ctx *ctx, int slot, host::Point *out, int count);
void msm_combine_member(halo_ctx *ctx, int slot, int b, host::Point *out);
// the same for `members.count` MSMs of n points each issued as ONE launch sequence; out[count]
int msm_enqueue_batch(halo_ctx *ctx, int slot, const uint32_t *d_bases, const MsmBatch &members, bool scalars_mont, size_t n);
bool msm_tagged_ready(const halo_ctx *ctx, const uint32_t *d_bases, size_t n);  // can a `tagged` launch (MsmBatch) over these points run now?
int msm_finish_batch(halo_ctx *ctx, int slot, host::Point *out, int count);
// sum scalars[i] * bases[i]; bases affine (device), scalars device; result host Jacobian (un-normalised)
// bases: native affine table (20 words per point)
int msm_run(halo_ctx *ctx, const uint32_t *d_bases, const uint64_t *d_scalars, bool scalars_mont, size_t n, host::Point *out);
// fn(0) .. fn(m - 1) on a bounded pool of host threads (at most 16; the caller's thread takes part)
inline void pool_run(size_t m, const std::function<void(size_t)> &fn) {
    unsigned hw = std::thread::hardware_concurrency();
    size_t nthreads = hw ? hw : 4;
    if (nthreads > 16) nthreads = 16;
    if (nthreads > m) nthreads = m;
    std::atomic<size_t> next{0};
    auto worker = [&]() { for (size_t i; (i = next.fetch_add(1)) < m;) fn(i); };
    std::vector<std::thread> th;
    for (size_t t = 1; t < nthreads; ++t) th.emplace_back(worker);
    worker();
    for (auto &t : th) t.join();
}
int urs_generate(halo_ctx *ctx, uint64_t first_index, uint64_t stride, size_t n, uint32_t *d_out_native);
int batch_to_affine(halo_ctx *ctx, const uint64_t *d_jac_words, size_t n, uint32_t *d_out_native);
int aff_words_to_native(halo_ctx *ctx, const uint64_t *d_in, size_t n, uint32_t *d_out);
int aff_native_to_words(halo_ctx *ctx, const uint32_t *d_in, size_t n, uint64_t *d_out);
int test_field_op(halo_ctx *ctx, int field, int op, const uint64_t *d_a, const uint64_t *d_b, size_t n, uint64_t *d_out);
int test_point_op(halo_ctx *ctx, int op, const uint64_t *d_a, const uint64_t *d_b, size_t n, uint64_t *d_out);

// ---- abi.hip: the budget for OPTIONAL device memory (the MSM fixed-base table, the fold table and its temporary), per device and
// process-wide: reserve() succeeds if what all contexts of this process hold on ctx's device plus `bytes` stays within the
// device's budget (halo_set_memory_budget; default 1/6 of the device's memory) and at least 2 x bytes are free right now
bool table_budget_reserve(halo_ctx *ctx, size_t bytes);
void table_budget_release(halo_ctx *ctx, size_t bytes);

// ---- foldtab.hip: the first two-level fold of an open from a comb table over the context's key
void foldtab_cancel_alloc(halo_ctx *ctx);  // joins the helper thread and frees what it obtained
int fold_points4_tab(halo_ctx *ctx, const uint32_t *d_src, uint32_t *d_dst, size_t m, const host::Fr s[3]);  // 1 = done, 0 = not applicable
void foldtab_release(halo_ctx *ctx);
void fold_digits_host(const host::Fr &s, int8_t out[44]);

// ---- multi.hip: MSMs over the key of a multi-device context, cut along the shards' blocks
int multi_attach_shards(halo_ctx *ctx, const int *devices, int n_dev, const uint64_t *bases_affine, uint64_t first_index);
void multi_destroy(halo_ctx *ctx);
bool multi_takes(const halo_ctx *ctx, const uint32_t *d_bases, size_t n);  // a stretch of the parent's own key?
int multi_begin(halo_ctx *ctx, int slot, size_t off, size_t n, const uint64_t *host_scalars, const uint64_t *dev_scalars, bool mont);
int multi_end(halo_ctx *ctx, int slot, host::Point *out);
int multi_batch_begin(halo_ctx *ctx, int slot, size_t off, size_t n, const MsmBatch &members, bool mont);
int multi_batch_end(halo_ctx *ctx, int slot, host::Point *out, int count);
int multi_run(halo_ctx *ctx, size_t off, size_t n, const uint64_t *dev_scalars, bool mont, host::Point *out);
int multi_host_run(halo_ctx *ctx, size_t off, size_t n, const uint64_t *scalars, bool mont, host::Point *out);  // synchronous, host scalars, per shard msm_host_run
int msm_host_begin(halo_ctx *ctx, int slot, size_t off, size_t n, const uint64_t *scalars, int mont);  // abi.hip
int msm_host_run(halo_ctx *ctx, size_t off, size_t n, const uint64_t *scalars, size_t valid, int mont, host::Point *out);  // abi.hip: synchronous, host scalars, zero-padded beyond `valid`

// ---- smsm.hip: the 4-launch pipeline for MSMs of up to 2^16 points (digits already in ws.d_canon)
// (winsum / winsum_plain may be pinned host memory; done: the counter to publish to, or null)
int quad_final_enqueue(halo_ctx *ctx, MsmWorkspace &ws, uint32_t Wt, uint32_t nseg, int k, uint64_t *winsum, uint64_t *winsum_plain,
                       const uint32_t *seg = nullptr, uint32_t *done = nullptr);
int rc_mid_enqueue(halo_ctx *ctx, const uint32_t *entries, uint32_t blocks, uint64_t *winsum, uint32_t *done = nullptr, uint32_t *ticket = nullptr);
int smsm_enqueue(halo_ctx *ctx, MsmWorkspace &ws, const uint32_t *d_bases, uint32_t base_off, size_t n, const MsmPlan &p, uint32_t Wt,
                 uint32_t kmax);
int smsm_prepare();

// ---- ipa.hip
int ipa_fold_points(halo_ctx *ctx, uint32_t *d_G, size_t m, const host::Fr &xi_mont);
int ipa_fold_points4(halo_ctx *ctx, const uint32_t *d_src, uint32_t *d_dst, size_t m, const host::Fr s[3]);
int ipa_fold_scalars(halo_ctx *ctx, uint64_t *d_c, uint64_t *d_z, size_t m, const host::Fr &xi, const host::Fr &xi_inv);
// out[0] = <xs0, ys0>, out[1] = <xs1, ys1> (either pair may be null to skip)
int fr_dot2(halo_ctx *ctx, const uint64_t *xs0, const uint64_t *ys0, const uint64_t *xs1, const uint64_t *ys1, size_t m,
            host::Fr out[2]);
int fr_dot2_launch(halo_ctx *ctx, const uint64_t *xs0, const uint64_t *ys0, const uint64_t *xs1, const uint64_t *ys1, size_t m);  // on ctx->stream
int fr_dot2_collect(halo_ctx *ctx, hipStream_t stream, size_t m, host::Fr out[2]);
int fr_powers(halo_ctx *ctx, const host::Fr &z, size_t n, uint64_t *d_out);
// d_v[i] *= a
int fr_scale(halo_ctx *ctx, uint64_t *d_v, size_t n, const host::Fr &a);
int fr_poly_eval(halo_ctx *ctx, const uint64_t *d_coeffs, size_t len, const host::Fr &z, host::Fr *out);
// d_out[k] (+)= scale * prod_{bit i of k} xis[lg_n - i]
int h_coeffs_dev(halo_ctx *ctx, const host::Fr *xis, size_t lg_n, const host::Fr &scale, bool accumulate, uint64_t *d_out);
int h_eval_batch(halo_ctx *ctx, const uint64_t *d_xis, size_t m, size_t lg_n, const host::Fr &z, uint64_t *d_out);
// m polynomials h_i at their own points z_i; m sums of K scalar multiples (canonical scalars, affine points) -> m Jacobian points
int h_eval_each(halo_ctx *ctx, const uint64_t *d_xis, const uint64_t *d_zs, size_t m, size_t lg_n, uint64_t *d_out);
int batch_small_msm(halo_ctx *ctx, const uint64_t *d_points, const uint64_t *d_scalars, size_t m, size_t K, uint64_t *d_out);
// SplitMix64 stream -> n Montgomery scalars (element i = draws 4i+1..4i+4 after state0)
int rng_scalars_dev(halo_ctx *ctx, uint64_t state0, size_t n, uint64_t *d_out);
int pbar_dev(halo_ctx *ctx, const uint64_t *d_q, size_t deg, const host::Fr &z, uint64_t *d_out);
int axpy_dev(halo_ctx *ctx, uint64_t *d_y, const uint64_t *d_x, size_t n, const host::Fr &a);
int pbar_stream_dev(halo_ctx *ctx, uint64_t state0, size_t deg, const host::Fr &z, uint64_t stride, uint64_t offset, size_t n_local,
                    uint64_t *d_out);
int bench_fr_kernel(halo_ctx *ctx, int which, size_t n, int reps);
int nofold_expand(halo_ctx *ctx, const uint64_t *d_c, const uint64_t *d_s, size_t m, size_t M, uint64_t *d_L, uint64_t *d_R);
int nofold_expand_tagged(halo_ctx *ctx, const uint64_t *d_c, const uint64_t *d_s, size_t m, size_t M, uint64_t *d_F);  // one array for a tagged launch
int nofold_s_update(halo_ctx *ctx, const uint64_t *d_s_in, size_t len, const host::Fr &xi, uint64_t *d_s_out);

// ---- abi.hip (device-pointer forms used by pcdl_acc.hip)
// H' = xi0 * H for this state (pcdl.rs:181): lets the rounds use the process-wide window table of H
void ipa_set_hprime_scalar(halo_ipa *st, const host::Fr &xi0);
// process-wide window table of the public point H (consts.rs:45-65), built on first use
const host::FixedBaseTable &public_h_table();
const host::FixedBaseTable &public_s_table();  // (k S and k H from the process-wide window tables: ~14 us against ~80 for a generic double-and-add)
int ipa_begin_dev(halo_ctx *ctx, size_t n, const uint64_t *d_coeffs_padded, const host::Fr &z, halo_ipa **out);
int upload_words(halo_ctx *ctx, uint64_t *dst, const uint64_t *src, size_t words);
int download_words(halo_ctx *ctx, uint64_t *dst, const uint64_t *src, size_t words);

}  // namespace halo

// guard used by every entry point that touches the device
#define HALO_CTX(ctx)                                                 \
    do {                                                              \
        if (!(ctx)) { halo::set_error("null context"); return HALO_E_ARG; } \
        hipError_t _e = hipSetDevice((ctx)->device);                  \
        if (_e != hipSuccess) return halo::hip_fail(_e, "hipSetDevice"); \
    } while (0)
