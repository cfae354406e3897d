// Every environment switch of the library in ONE place, read ONCE (the first call of tuning()), plus the hooks of the
// development library.  INTEGRATION.md section 6 documents the same list for the user.
//
//   * None of the switches changes a result: they select between code paths that the GPU suite holds bit-equal, or print.
//   * The product library (libhalo_hip.so) reads NO fault injector from the environment.  DevHooks are plain fields that only
//     libhalo_hip_dev.so's halo_dev_hook() writes (include/halo_accumulation_dev.h): a process that does not load and call the
//     development library cannot be made to fail or to take a test path by a stray variable.
#pragma once
#include <cstddef>

namespace halo {

struct Tuning {
    // ---- operator-facing
    bool trace = false;            // HALO_TRACE=1          log allocations (address ranges), graph captures and replays to stderr
    bool ipa_timing = false;       // HALO_IPA_TIMING=1     print where the host side of the IPA rounds went when a state is destroyed
    bool memory_budget_set = false;
    size_t memory_budget = 0;      // HALO_MEMORY_BUDGET=48G  optional-memory budget per device for hosts that cannot call halo_set_memory_budget
    int graphs = -1;               // HALO_GRAPHS=0         never replay launch graphs (default: on; halo_set_graphs overrides)
    int fold_async = -2;           // HALO_FOLD_ASYNC=-1|0|1  folds beside the rounds: automatic / never / wherever possible (halo_set_fold_async)
    bool host_split_set = false;   // HALO_HOST_SPLIT="4,12"  a host-scalar halo_msm runs as stretches of these many sixteenths of its points, each behind
    int host_pieces = 2;           //   the copy of its own scalars (at most 4 entries, sum 16; "16": one copy in front of one launch sequence).  Unset:
    int host_split[4] = {4, 12, 0, 0};  //   4,12 below 2^21 points, 2,4,4,6 from there on (abi.hip host_split_for: measured, HISTORY.md)
    int fold_table_after = 8;      // HALO_FOLD_TABLE_AFTER=k  automatic mode builds the fold table after k full-size opens on a key (0: never automatically)
    // ---- development switches for A/B runs on one box (tools/ab.sh, tools/env_ab.sh); defaults are the measured winners
    const char *plan = nullptr;    // HALO_PLAN="16:12,15:12"  window bits by lg n for the general / small pipelines
    int dots_first = -1;           // HALO_DOTS_FIRST=0|1   order of the round's dot products and MSM launches (-1: by size)
    bool ipa_c_hint = true;        // HALO_IPA_C_HINT=0     no 11-bit windows for the half-zero scalars of the 2^16-point rounds
    bool u_from_last_round = true; // HALO_U_FROM_LAST_ROUND=0  U from one more MSM instead of the last round's sums
    int pow_e = 0;                 // HALO_POW_E=4..64      chain length of k_powers / k_h_coeffs (power of two)
    int dot_blocks = 0;            // HALO_DOT_BLOCKS       block cap of k_dot2_partial (default 512)
    int smsm_kmax = 0;             // HALO_SMSM_KMAX        task length of the small pipeline
    int late_kmax = 0;             // HALO_LATE_KMAX        ... of MSMs of <= 2^14 points (default 8)
    bool piece_alternate = true;   // HALO_PIECE_ALTERNATE=0  pieces of a large synchronous MSM one after the other on one stream
    int graph_cache = 8;           // HALO_GRAPH_CACHE=1..8 launch graphs kept per slot
    bool direct_results = true;    // HALO_DIRECT_RESULTS=0 copy kernel + stream wait instead of the last kernel publishing its sums
    bool tagged = true;            // HALO_TAGGED=0         L and R of the first rounds as two launches instead of one tagged one
    bool reduce_rc = true;         // HALO_REDUCE_RC=0      running-sum window sums for the table plans instead of rows / columns
    int reduce1_waves = 1024;      // HALO_REDUCE1_WAVES    wave cap of k_msm_reduce1 (0: none)
    int smsm_wave_task = 0;        // HALO_SMSM_WAVE_TASK   bucket length from which the small pipeline uses wave tasks
    int smsm_waves = 1024;         // HALO_SMSM_WAVES       wave cap of k_smsm_reduce (0: none)
    bool smsm_fused = false;       // HALO_SMSM_FUSED=1     k_smsm_reduce and k_smsm_final as one launch
    bool host_inv_fermat = false;  // HALO_HOST_INV_FERMAT=1  host inverses by the Fermat power instead of division steps
    int spin_us = 50;              // HALO_SPIN_US          how long msm_wait polls before it starts yielding the core
};
const Tuning &tuning();

// Fault injectors and test paths: zero unless the development library sets them.
struct DevHooks {
    int table_fail = 0;        // "table_fail"       the allocation of a fixed-base / fold table reports out-of-memory
    int force_peer_copy = 0;   // "force_peer_copy"  multi-device contexts stage device scalars through a peer copy even on the same GPU
    int shard_fail_rank = -1;  // "shard_fail_rank" / "shard_fail_at": that rank fails locally before its collective number `at`
    int shard_fail_at = -1;
};
DevHooks &dev_hooks();

inline bool debug_trace() { return tuning().trace; }

}  // namespace halo
