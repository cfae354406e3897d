#!/usr/bin/env python3
"""Headline benchmark: Pippenger MSMs/sec at n = 2^20 (BASELINE.json configs[1]) on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--log-n 20] [--open-steps J]

However it is started it prints one line: under torch.distributed.run (WORLD_SIZE set) it is one rank of N; started plainly
with --gpus N > 1 it drives the N GPUs from this ONE process through a multi-device context (as with --one-process).

A step = one MSM over n random scalars (resident in HBM) and the URS bases G_0..G_{n-1}
(derived on the GPU by the reference's main.rs rule).  N > 1 (launched by torch.distributed.run,
one rank per GPU): the SAME n-point MSMs are sharded -- by Pippenger windows (default: every rank keeps
the key and the scalars and does 1/N of the bucket work) or by base/scalar index (--shard index) -- each
rank reduces its share to one point and the partials are all-gathered over RCCL and summed (strong scaling).
Prints ONE JSON line on rank 0.
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def latest_profile(suffix):
    """profiles/rNN_<suffix> of the latest round that has one (PMC counters need rocprofv3 passes of their own -- tools/evidence.sh --
    so the traffic figures of the JSON line are STATIC, taken from the committed pass named beside them)"""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)))
    return files[-1] if files else None


def main():
    # stdout carries exactly ONE line, the JSON: everything else any library prints there during the run (RCCL announces
    # its version on stdout when a communicator is created) is sent to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    # Python's cyclic collector stays off during the run and is called between the legs instead: with the arrays of the earlier
    # legs alive a full collection took 4-6 ms and landed in every fifth open + check (16.7 ms -> 21-23 ms: four of 24 samples in
    # profiles/r04's first bench line); the library's host side is C++, a caller in Rust has no collector at all.
    gc.disable()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--depth", type=int, default=4, help="launch sequences in flight (1..4)")
    ap.add_argument("--batch", type=int, default=0, help="MSMs per launch sequence (1..8; 0 = 1 on one or two GPUs, 4 on more)")
    ap.add_argument("--shard", choices=["window", "index", "auto"], default="auto",
                    help="N > 1: split every MSM by Pippenger windows (key and scalars replicated) or by base/scalar index; "
                         "auto = index blocks while a rank's block has >= 2^17 points (a fixed-base-table MSM), else windows")
    ap.add_argument("--open-steps", type=int, default=12, help="PCDL open+check repetitions at N=1 (0 = skip)")
    ap.add_argument("--cpu-msms", type=int, default=2, help="oracle MSMs timed for cpu_baseline at N=1 (0 = skip)")
    ap.add_argument("--min-seconds", type=float, default=1.0, help="repeat the K-step timed region until this much time is covered; the median repetition is reported")
    ap.add_argument("--one-process", action="store_true",
                    help="--gpus N from ONE process (no torch.distributed): a multi-device context (halo_ctx_create_urs_multi), one "
                         "index-block shard per GPU, partial points added on the host")
    ap.add_argument("--devices", default="", help="--one-process: comma-separated device ids (default 0..N-1; ids may repeat for a rehearsal)")
    ap.add_argument("--host-steps", type=int, default=8, help="MSMs with the scalars in host memory (halo_msm and its begin/end halves) at N=1 (0 = skip)")
    ap.add_argument("--fr-reps", type=int, default=20, help="back-to-back launches of each bandwidth-side Fr kernel at N=1 (0 = skip)")
    ap.add_argument("--asdl-steps", type=int, default=8, help="ASDL chain steps (random_instance + prover + verifier, then one decider) at N=1 (0 = skip)")
    ap.add_argument("--var-steps", type=int, default=60, help="MSMs through the table-free (variable-base) pipeline at N=1 (0 = skip)")
    ap.add_argument("--cpu-log-n", type=int, default=14, help="size of the CPU baseline of open + check and of the ASDL chain (the reference's own D + 1 = 2^14, consts.rs:23)")
    ap.add_argument("--concurrent-opens", type=int, default=2, help="open + check pairs in flight on separate contexts / host threads for the throughput figure at N=1 (0 or 1 = skip)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # HALO_BENCH_BACKEND=gloo lets several ranks share one GPU (rehearsal of the N > 1 path on a 1-GPU box)
    backend = os.environ.get("HALO_BENCH_BACKEND", "nccl")
    gpu = local_rank % max(torch.cuda.device_count(), 1)
    # HALO_BENCH_FORCE_DIST=1: one rank, but with the process group and the all-gather of the N > 1 path (RCCL rehearsal)
    force_dist = world == 1 and os.environ.get("HALO_BENCH_FORCE_DIST") == "1"
    if force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", gpu))
        else:
            dist.init_process_group(backend=backend)
    # --gpus N runs however it is started: under a launcher (WORLD_SIZE in the environment) this process is one rank of N;
    # started plainly it drives all N GPUs itself through a multi-device context (halo_ctx_create_urs_multi) -- no launcher,
    # no re-exec.  With fewer than N GPUs on the box the device ids repeat (a rehearsal: config.sharding says so).
    launched = "WORLD_SIZE" in os.environ
    one_proc = args.one_process or (args.gpus > 1 and not launched)
    assert world == (1 if one_proc else args.gpus), "under torch.distributed.run --nproc-per-node must equal --gpus"
    devices, devices_note = None, ""
    if one_proc:
        have = max(torch.cuda.device_count(), 1)
        devices = [int(x) for x in args.devices.split(",")] if args.devices else [k % have for k in range(args.gpus)]
        assert len(devices) == args.gpus
        if not args.one_process:
            devices_note = "started without a launcher (WORLD_SIZE unset): "
        if len(set(devices)) < len(devices):
            devices_note += "REHEARSAL, device ids repeat (%d GPU(s) visible): " % have
        gpu = devices[0]
    torch.cuda.set_device(gpu)
    dev = torch.device("cuda", gpu)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")  # where the collectives' tensors live

    import halo_accumulation_amd as h
    from halo_accumulation_amd import pcdl
    from halo_accumulation_amd.sharded import ShardedMsm, shard_range

    n = 1 << args.log_n
    per_rank = n // world
    if args.shard == "auto":
        # an index block of >= 2^17 points is a fixed-base-table MSM on its rank (small-key plan below 2^20
        # points, in batches; pieces above 1.3 M); smaller blocks would be latency-bound: window shards of the full MSM then
        args.shard = "index" if per_rank >= (1 << 17) else "window"
    # MSMs per launch, measured per rank on one GPU (tools/time_table_batch.py, tools/sweep_batch.py): index blocks go through
    # the table plan as batches of 8 / 4 / 2 at 2^17 / 2^18 / 2^19 points (count x coarse ranges <= 512), one per launch from 2^20
    if args.batch > 0:
        batch = args.batch
    elif world > 1 and args.shard == "index":
        batch = 8 if per_rank < (1 << 18) else 4 if per_rank < (1 << 19) else 2 if per_rank < (1 << 20) else 1
    else:
        batch = 1 if world <= 2 else 4
    if one_proc and args.batch <= 0:
        # every shard of the multi-device context runs its index block of a batch as one launch (multi.hip multi_batch_begin)
        shard_pts = n // len(devices)
        batch = 8 if shard_pts < (1 << 18) else 4 if shard_pts < (1 << 19) else 2 if shard_pts < (1 << 20) else 1
    window_mode = world > 1 and args.shard == "window"
    if window_mode:
        # every rank holds the whole key and all scalars (128 + 32 MiB at n = 2^20 of 288 GiB) and computes
        # the Pippenger windows [r*W/P, (r+1)*W/P) of every MSM: 1/P of the bucket work at full-size efficiency
        lo, hi, part, parts = 0, n, rank, world
    else:
        # index shard: this rank's block of the key (main.rs:35-45: G_i = hash(i + 2)) and of the scalars
        (lo, hi), part, parts = shard_range(n, rank, world), 0, 1
    cold_path = None
    if world == 1 and not one_proc and args.open_steps > 0 and args.log_n >= 18:
        # THE COLD PATH: a FRESH context in its default configuration does ten open + check pairs, each timed.  The
        # automatic fold table waits for the key's 8th full-size open (csrc/foldtab.hip): a caller with a handful of opens never
        # reserves or pays for it.  Run before the main context exists, closed afterwards: nothing of it is left for the timed legs.
        cold = h._lib.Context(urs_n=n, device=gpu)
        d_cc = torch.empty((n + 2) * 4, dtype=torch.int64, device=dev)
        cold.rng_scalars_dev(0x48414C4F00000003, n + 2, d_cc.data_ptr())
        zw_c = np.ascontiguousarray(d_cc[4 * n:].cpu().numpy().view(np.uint64).reshape(2, 4))
        C_c = pcdl.commit_dev(cold, d_cc.data_ptr(), n, n - 1)
        v_c = cold.poly_eval(np.ascontiguousarray(d_cc[: 4 * n].cpu().numpy().view(np.uint64).reshape(n, 4)), zw_c[0])
        torch.cuda.synchronize()
        ms, st, opt = [], [], []
        for _ in range(10):
            t0 = time.perf_counter()
            p_c = pcdl.open_dev(cold, [1], d_cc.data_ptr(), n, C_c, n - 1, zw_c[0])
            pcdl.check_proof(cold, C_c, n - 1, zw_c[0], v_c, p_c)
            ms.append(round((time.perf_counter() - t0) * 1e3, 3))
            st.append(int(cold.info(5))); opt.append(int(cold.info(1)))
        cold_path = {"first_10_opens_ms": ms, "fold_table_status_after_each": st, "fold_table_bytes_after_each": opt,
                     "sum_ms": round(sum(ms), 3), "fold_table_after_opens": int(os.environ.get("HALO_FOLD_TABLE_AFTER", "8")),
                     "note": "fresh context, default configuration (no halo_set_* call), open + check pairs one at a time, polynomial resident; status: 0 nothing "
                             "yet, 1 memory requested, 2 built -- opens 1-7 never reserve optional memory, the 8th asks for it on a helper thread, "
                             "the first later open that finds it builds the table (that open carries the build)"}
        cold.close()
        del d_cc, cold
        torch.cuda.empty_cache()
    torch.cuda.synchronize()
    t_setup0 = time.perf_counter()
    ctx = h._lib.Context(urs_n=hi - lo, first_index=2 + lo, device=gpu) if not one_proc else h._lib.Context(urs_n=n, devices=devices)
    t_setup_urs = time.perf_counter() - t_setup0  # the key derived on the device (k_urs) + workspace of slot 0
    # scalars: SplitMix64 seed ...02, generated on the device by the library's own
    # generator.  MSM j of a launch takes the j-th block of n scalars of that stream (4 draws per scalar).
    GAMMA, MASK, SEED = 0x9E3779B97F4A7C15, (1 << 64) - 1, 0x48414C4F00000002
    d_sets = []
    for j in range(batch):
        d = torch.empty((hi - lo) * 4, dtype=torch.int64, device=dev)
        ctx.rng_scalars_dev((SEED + 4 * (j * n + lo) * GAMMA) & MASK, hi - lo, d.data_ptr())
        d_sets.append(d)
    ptrs = [d.data_ptr() for d in d_sets]
    # Independent MSMs are pipelined: `batch` of them share one launch sequence (their windows run side by
    # side through every kernel) and up to `depth` launches are in flight on the context's slots, so that
    # one launch's low-occupancy tail (bucket reduce, D2H of the window sums, host Horner) overlaps the
    # next one's sort/accumulate kernels.  Every MSM is completed (and, for N > 1, all-gathered and
    # combined) inside the timed region.
    gather = ShardedMsm(None, h._lib.point_sum, device=coll_dev, always_collective=force_dist)
    if world > 1 or force_dist:
        # first collective now: the communicator's lazy allocations happen before any launch graph exists
        gather.gather_batch([np.zeros(12, dtype=np.uint64)] * batch)

    # one-time cost of the first MSM over the key (fixed-base table build, first-use allocations) -- and the latency of one
    # MSM alone (nothing else in flight, scalars resident): what a single-threaded caller of the reference API sees per call
    t0 = time.perf_counter()
    first = ctx.msm_dev(ptrs[0], hi - lo) if not window_mode else None
    t_first = time.perf_counter() - t0
    solo = []
    if not window_mode:
        for _ in range(12):
            t0 = time.perf_counter()
            again = ctx.msm_dev(ptrs[0], hi - lo)
            solo.append(time.perf_counter() - t0)
        assert again.tolist() == first.tolist()
    cfg = {"depth": args.depth}
    outs = [None] * batch  # latest combined result per scalar set

    def run_steps(k):
        depth = cfg["depth"]
        pending = []   # (slot, members)
        gathers = []   # all-gathers issued and not yet collected: each runs under the launches enqueued after it

        def collect():
            for j, pt in enumerate(gather.gather_finish(gathers.pop(0))):
                outs[j] = pt

        def finish():
            slot, m = pending.pop(0)
            partials = ctx.msm_dev_batch_end(slot, m)
            gathers.append(gather.gather_start([partials[j] for j in range(m)]))  # one all-gather per launch, asynchronous
            while len(gathers) > 1:
                collect()

        launches = 0
        while k > 0:
            m = min(batch, k)
            if len(pending) == depth:
                finish()
            slot = launches % depth
            ctx.msm_dev_batch_begin(slot, ptrs[:m], hi - lo, part=part, parts=parts)  # (a multi-device context fans the batch out over its shards)
            pending.append((slot, m))
            launches += 1
            k -= m
        while pending:
            finish()
        while gathers:
            collect()

    def barrier():
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(3 * args.depth * batch)  # untimed: every slot allocates its workspace and captures its launch graph
    run_steps(args.warmup)
    # The timed region is EXACTLY --steps steps between barrier + synchronize; it is repeated until --min-seconds are covered
    # (a 20-step region lasts 25 ms: one sample says little) and the median repetition is reported, all of them listed.
    reps = []
    while True:
        barrier()
        t0 = time.perf_counter()
        run_steps(args.steps)
        barrier()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        if world > 1 or force_dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)  # the slowest rank's time
        reps.append(float(t.item()))
        if sum(reps) >= args.min_seconds or len(reps) >= 200:
            break
    dt = sorted(reps)[len(reps) // 2]
    out = outs[0]

    # Kernel durations for the roofline: HIP events on the stream each kernel is launched on, in two
    # extra passes of the same loop right after the timed region (the event brackets need individual
    # launches, the timed region replays the launch sequence as a hipGraph): (a) as timed, several
    # launches in flight -- kernels of different slots time-share the CUs and every duration stretches;
    # (b) one launch in flight: the kernel by itself, which is what a roofline fraction is about.
    prof_steps = max(batch, min(args.steps, 8 * batch) // batch * batch)  # whole launches only
    ctx.prof_enable(2)
    ctx.prof_reset()
    barrier()
    run_steps(prof_steps)
    barrier()
    prof = ctx.prof()
    cfg["depth"] = 1
    ctx.prof_reset()
    barrier()
    run_steps(prof_steps)
    barrier()
    prof_solo = ctx.prof()
    cfg["depth"] = args.depth
    ctx.prof_enable(0)

    # what the collective layer itself reports: backend, ranks, and how many distinct devices they sit on
    backend_name, world_reported, distinct_gpus, coll_name = "none", 1, 1, "no"
    if world > 1 or force_dist:
        backend_name, world_reported = dist.get_backend(), dist.get_world_size()
        coll_name = "RCCL" if backend_name == "nccl" else backend_name
        props_me = torch.cuda.get_device_properties(gpu)
        ident = "%s/%s/%d" % (os.uname().nodename, getattr(props_me, "uuid", props_me.name), gpu)
        idents = [None] * world_reported
        dist.all_gather_object(idents, ident)
        distinct_gpus = len(set(idents))
    result = None
    if rank == 0:
        dom = "k_msm_accumulate" if "k_msm_accumulate" in prof_solo else "k_smsm_accumulate"  # n <= 2^16: the small-MSM pipeline
        acc_ms, acc_cnt = prof_solo.get(dom, (0.0, 0))
        # per launch sequence: an MSM of more than 1.3 M points runs the kernel once per piece, and the
        # algorithmic bytes below are those of the whole launch
        n_launches = max(prof_steps // batch, 1)
        if one_proc:
            n_launches *= len(devices)  # every shard runs its own launch sequence per MSM: the figures below are per shard launch
        kern_s = acc_ms / n_launches * 1e-3
        ovl_ms, ovl_cnt = prof.get(dom, (0.0, 0))
        ovl_cnt = n_launches if ovl_cnt else 0
        # 64 B base + 32 B scalar per point, one point out; a launch carries `batch` MSMs (their
        # index block or their 1/parts window share on this rank)
        alg_bytes = batch * (96 * (hi - lo) // parts + 64) if not one_proc else batch * (96 * (n // len(devices)) + 64)
        # HBM-side bytes per launch: PMC counters need a rocprofv3 --pmc pass of their own (tools/evidence.sh), they cannot be
        # read inside this run -- the figure is STATIC, taken from the committed pass named in traffic_source
        traffic, traffic_source = None, None
        pmc = latest_profile("pmc_traffic.json")
        if world == 1 and args.log_n == 20 and pmc:
            pj = json.load(open(pmc))
            traffic = pj["kernels"].get(dom, {}).get("traffic_bytes_per_launch")
            traffic_source = "static: profiles/%s (%s)" % (os.path.basename(pmc), pj.get("note", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes"))
        achieved = alg_bytes / kern_s / 1e9 if kern_s > 0 else 0.0
        # VALU view of the same kernel: one mixed XYZZ addition is 1143 v_mad_u64_u32 on the
        # kernel's hot path (counted in the gfx950 ISA), one per point per window; the issue peak is the measured
        # 5.26 cycles per wave-instruction (profiles/r01_microbench_instr_throughput.txt) on 4 SIMDs per CU
        props = torch.cuda.get_device_properties(gpu)
        # additions per point: 13 with the c = 20 table (>= 2^20 points on this rank), 15 with the small-key table (index
        # blocks of 2^17 .. 2^19 points), 16 for window shards of the general 16-window plan
        plan_w = None
        if args.log_n >= 20:
            shard_pts = (hi - lo) if not one_proc else n // len(devices)
            plan_w = 16 if window_mode else (13 if shard_pts >= (1 << 20) else 15 if shard_pts >= (1 << 17) else None)
        valu = None
        if plan_w and kern_s > 0:
            wave_mads = batch * shard_pts * plan_w / parts * 1143 / 64
            peak = props.multi_processor_count * 4 * 2.4e9 / 5.26  # 2.40 GHz engine clock (the microbenchmark's reading)
            valu = {"unit": "v_mad_u64_u32 wave-instr/s", "achieved": wave_mads / kern_s, "peak": peak, "frac": wave_mads / kern_s / peak}
            # what the SIMDs did during the kernel, from the committed SQ counter pass (static, like `traffic`): vector
            # instructions issued per 4-cycle issue slot over all 1024 SIMDs, the clock the chip held, parked wave cycles
            sq = latest_profile("sq_msm.json")
            if world == 1 and args.log_n == 20 and sq:
                k = json.load(open(sq))["kernels"].get(dom)
                if k:
                    valu["issue_slots_used"] = k["valu_per_quad"]
                    valu["clock_ghz_during_kernel"] = k["clock_ghz"]
                    valu["wave_cycles_parked"] = k["wave_parked"]
                    valu["issue_source"] = "static: profiles/%s (rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAIT_ANY ..., one MSM in flight)" % os.path.basename(sq)
        result = {
            "metric": "MSMs/sec (Pippenger, Pallas, n=2^%d random scalars/URS points, bit-exact vs CPU)" % args.log_n,
            "value": args.steps / dt, "unit": "MSM/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "timed_region": {"repetitions": len(reps), "reported": "median", "seconds_each": [round(x, 6) for x in reps[:32]]},
            "dtype": "u32x8 (256-bit Montgomery integer)", "data": "synthetic",
            "config": {"workload": "Pippenger MSM n=2^%d, bases = URS G_i by main.rs rule, scalars SplitMix64 seed 0x48414C4F00000002" % args.log_n,
                       "sharding": (devices_note + "one process, multi-device context: %d index-block shards on devices %s, partial points added on the host "
                                    "(no collective)" % (len(devices), devices)) if one_proc and args.gpus > 1 else "single GPU" if world == 1 else
                                   ("Pippenger windows split over the ranks (key + scalars replicated), %s all-gather of 96 B partials" % coll_name if window_mode
                                    else "block index shard per rank + %s all-gather of 96 B partials" % coll_name),
                       "collective_backend": backend_name, "world_size": world_reported,
                       "distinct_gpus": len(set(devices)) if one_proc else distinct_gpus,
                       "msms_per_launch": batch, "launches_in_flight": args.depth,
                       "window_bits": "a rank's block of >= 2^20 points: 20 with fixed-base tables over the context's key (13 windows, one set "
                                      "of 2^19 buckets); index blocks of 2^17..2^19 points: 17 (15 windows, 2^16 buckets per MSM of a batch); "
                                      "window shards: 16 (16-window general plan)"},
            "roofline": {"bound": "hbm", "kernel": dom, "symbol": "halo::" + str(dom), "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel_ms": kern_s * 1e3, "kernel_ms_while_%d_launches_in_flight" % args.depth: ovl_ms / max(ovl_cnt, 1),
                         "algorithmic_bytes": alg_bytes, "valu": valu,
                         "note": "integer-VALU-bound kernel: see DESIGN.md for the VALU roofline"},
            "solo_latency_ms": (sorted(solo)[len(solo) // 2] * 1e3 if solo else None),
            "context_setup_ms": {"key_on_device_and_workspace": t_setup_urs * 1e3,
                                 "first_msm_incl_table_build": t_first * 1e3,
                                 "note": "one-time: halo_ctx_create_urs (k_urs over this rank's block + slot-0 workspace), then the first MSM, which "
                                         "builds the fixed-base table (k_table_step per window) before it runs"},
            "table_bytes": ((13 if hi - lo >= (1 << 20) else 15) * 128 * (hi - lo) if (not window_mode and hi - lo >= (1 << 17)) else 0),
            "hbm_roofline_frac_whole_msm": (args.steps / dt) * (96 * n + 64) / (HBM_PEAK_GBS * 1e9),
        }

    if one_proc and args.gpus > 1:
        # cross-check: the same MSM on a plain one-device context
        full = h._lib.Context(urs_n=n, first_index=2, device=gpu)
        ok = full.msm_dev(ptrs[0], n).tolist() == outs[0].tolist()
        full.close()
        result["sharded_equals_single_gpu"] = ok
        assert ok, "multi-device MSM differs from the single-GPU MSM"
    elif world == 1:
        # Everything below runs on the GPU only; the CPU restatement (oracle/) is touched in ONE leg at the very end, which
        # times it on the host cores as the reported baseline and compares what the GPU produced above with it, bit for bit.
        cpu_model = "unknown"
        try:
            cpu_model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
        except Exception:
            pass
        sc_host = np.ascontiguousarray(d_sets[0].cpu().numpy().view(np.uint64).reshape(n, 4))
        gpu_side = {}  # what the CPU leg compares: inputs and GPU outputs of the small (2^cpu_log_n) open / ASDL runs

        def median(xs):
            xs = sorted(xs)
            return xs[len(xs) // 2] if len(xs) % 2 else 0.5 * (xs[len(xs) // 2 - 1] + xs[len(xs) // 2])

        # what the reference's shim gets (integration/ffi.rs: synchronous calls, host Vecs): filled in by the legs below
        dropin = {"note": "the calls integration/ffi.rs makes, as it makes them: halo_msm on scalars in pageable host memory (a fresh buffer per call), the "
                          "open as the patched pcdl::open drives it (halo_ipa_begin on host coefficients, then per round halo_ipa_round_lr_partial + "
                          "halo_open_combine + halo_ipa_round_fold: integration/harness.c), halo_h_accumulate for two instances; results asserted equal"}
        result["dropin"] = dropin
        if cold_path is not None:
            result["cold_path"] = cold_path
        if args.host_steps > 0:
            # the same MSM with the scalars handed over in HOST memory (halo_msm: 32 MiB H2D per MSM at n = 2^20, pageable):
            # the PCIe-inclusive rate; never `value`
            ctx.msm(sc_host)
            t0 = time.perf_counter()
            for _ in range(args.host_steps):
                got_h = ctx.msm(sc_host)
            h2d_dt = (time.perf_counter() - t0) / args.host_steps
            assert got_h.tolist() == out.tolist()
            # the same through the asynchronous halves (halo_msm_begin / halo_msm_end): the copy of the next MSM's scalars runs while
            # the current MSM's kernels do
            K, D = 16, 3
            ctx.msm_begin(0, sc_host); ctx.msm_end(0)
            t0 = time.perf_counter()
            for k in range(K + D):
                if k >= D:
                    got_p = ctx.msm_end(k % D)
                if k < K:
                    ctx.msm_begin(k % D, sc_host)
            pipe_dt = (time.perf_counter() - t0) / K
            assert got_p.tolist() == out.tolist()
            t0 = time.perf_counter()
            for _ in range(4):
                d_tmp = torch.from_numpy(sc_host.view(np.int64)).to(dev)
                torch.cuda.synchronize()
            copy_dt = (time.perf_counter() - t0) / 4
            # ... and from a FRESH pageable buffer every call (a Rust Vec that was just filled: pages the driver has never pinned)
            fresh = []
            for _ in range(max(args.host_steps, 8)):
                buf = sc_host.copy()
                t0 = time.perf_counter()
                got_f = ctx.msm(buf)
                fresh.append(time.perf_counter() - t0)
                assert got_f.tolist() == out.tolist()
                del buf
            dropin["halo_msm_fresh_pageable_ms"] = median(fresh) * 1e3
            dropin["halo_msm_same_buffer_ms"] = h2d_dt * 1e3
            dropin["halo_msm_stretches_sixteenths"] = os.environ.get("HALO_HOST_SPLIT", "4,12 (default)")
            result["end_to_end_host_scalars"] = {"value": 1.0 / h2d_dt, "unit": "MSM/s", "ms": h2d_dt * 1e3,
                                                 "pipelined_value": 1.0 / pipe_dt, "pipelined_ms": pipe_dt * 1e3,
                                                 "h2d_copy_alone_ms": copy_dt * 1e3, "h2d_copy_GBs": sc_host.nbytes / copy_dt / 1e9,
                                                 "note": "halo_msm: %d MiB of scalars copied from pageable host memory each call, one MSM in flight (a call is "
                                                         "the copy + the latency of one MSM); pipelined: halo_msm_begin/_end on %d slots, the next copy under the "
                                                         "current kernels; h2d_copy_alone: the same buffer through torch, for scale" % (sc_host.nbytes >> 20, D)}
        if args.var_steps > 0 and not window_mode:
            # VARIABLE-BASE figure (group.rs:24-26 is VariableBaseMSM::msm_unchecked): the same MSM, same loop, with the fixed-base
            # table switched off (halo_set_table_mode(ctx, 0) releases it) -- the general pipeline, which is what halo_msm_affine,
            # point_dot, every folded key of the IPA and every generator that is not the context's key get.
            ctx.set_table_mode(0)
            var_first = ctx.msm_dev(ptrs[0], n)
            assert var_first.tolist() == out.tolist(), "general pipeline differs from the table pipeline"
            run_steps(3 * args.depth)
            vreps = []
            while sum(vreps) < 0.4 and len(vreps) < 20:
                barrier()
                t0 = time.perf_counter()
                run_steps(args.var_steps)
                barrier()
                vreps.append(time.perf_counter() - t0)
            vdt = median(vreps)
            vsolo = []
            for _ in range(8):
                t0 = time.perf_counter()
                ctx.msm_dev(ptrs[0], n)
                vsolo.append(time.perf_counter() - t0)
            cfg["depth"] = 1
            ctx.prof_enable(2); ctx.prof_reset()
            run_steps(8)
            vprof = ctx.prof()
            ctx.prof_enable(0)
            cfg["depth"] = args.depth
            vk_ms = vprof.get("k_msm_accumulate", (0.0, 0))[0] / 8
            vtraffic, vsrc = None, None
            pmc = latest_profile("pmc_traffic_general.json")
            if args.log_n == 20 and pmc:
                pj = json.load(open(pmc))
                vtraffic = pj["kernels"].get("k_msm_accumulate", {}).get("traffic_bytes_per_launch")
                vsrc = "static: profiles/%s (%s)" % (os.path.basename(pmc), pj.get("note", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes"))
            vach = (96 * n + 64) / (vk_ms * 1e-3) / 1e9 if vk_ms > 0 else 0.0
            result["variable_base"] = {
                "value": args.var_steps / vdt, "unit": "MSM/s", "ms_per_step": vdt / args.var_steps * 1e3, "solo_latency_ms": median(vsolo) * 1e3,
                "steps": args.var_steps, "repetitions": len(vreps), "launches_in_flight": args.depth,
                "roofline": {"bound": "hbm", "kernel": "k_msm_accumulate", "achieved": vach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": vach / HBM_PEAK_GBS,
                             "kernel_ms": vk_ms, "algorithmic_bytes": 96 * n + 64, "traffic": vtraffic, "traffic_source": vsrc},
                "hbm_roofline_frac_whole_msm": (args.var_steps / vdt) * (96 * n + 64) / (HBM_PEAK_GBS * 1e9),
                "note": "halo_set_table_mode(ctx, 0): no fixed-base table -- the general Pippenger pipeline (signed windows of <= 16 bits, one bucket set "
                        "per window), the same %d-point MSM over the same key and scalars, same result (asserted); this is the rate of halo_msm_affine / "
                        "point_dot / the IPA's folded keys, minus their upload" % n}
            ctx.set_table_mode(-1)
            assert ctx.msm_dev(ptrs[0], n).tolist() == out.tolist()  # (the table is rebuilt here, untimed)
        small_n = 1 << min(args.cpu_log_n, args.log_n)
        ctx_small = None
        if (args.open_steps > 0 or args.asdl_steps > 0) and args.cpu_msms > 0:
            ctx_small = ctx if small_n == n else h._lib.Context(urs_n=small_n, device=gpu)
        if args.open_steps > 0:
            # BASELINE configs[2]: pcdl::open + check at the same n.  Compute-only = the polynomial already resident in device
            # memory (halo_pcdl_open_dev); end-to-end = coefficients handed over in pageable host memory (halo_pcdl_open).
            d = n - 1
            d_co = torch.empty((n + 2) * 4, dtype=torch.int64, device=dev)
            ctx.rng_scalars_dev(0x48414C4F00000003, n + 2, d_co.data_ptr())  # seed ...03: n coefficients, then z, w
            co = np.ascontiguousarray(d_co.cpu().numpy().view(np.uint64).reshape(n + 2, 4))
            coeffs, zw = np.ascontiguousarray(co[:n]), np.ascontiguousarray(co[n:])
            C = pcdl.commit_dev(ctx, d_co.data_ptr(), n, d)
            assert C.tolist() == pcdl.commit(ctx, coeffs, d).tolist()
            v = ctx.poly_eval(coeffs, zw[0])

            def one(c_=None):
                p_ = pcdl.open_dev(c_ or ctx, [1], d_co.data_ptr(), n, C, d, zw[0])
                pcdl.check_proof(c_ or ctx, C, d, zw[0], v, p_)
                return p_

            def timed(fn, reps):
                torch.cuda.synchronize()
                gc.collect()  # (the cyclic collector is off for the whole run, see main(): collected here, between the timed loops)
                ts, p_ = [], None
                for _ in range(reps):
                    t0 = time.perf_counter()
                    p_ = fn()
                    ts.append(time.perf_counter() - t0)
                return ts, p_

            # (a) WITHOUT the fold table (halo_set_fold_table(ctx, 0): the price list of the 35 GB): the first open of the context
            # (it allocates the IPA buffers), then the steady state
            ctx.set_fold_table(0)
            (t_first_open,), pi0 = timed(one, 1)
            one()
            ts_plain, pi_plain = timed(one, args.open_steps)
            assert pi_plain.tolist() == pi0.tolist()
            # (b) WITH it, asked for explicitly (mode 1: allocated and built at the next open; the default mode -1 asks for the memory
            # on a helper thread at the first full-size open and builds at the first later open that finds it)
            ctx.set_fold_table(1)
            (t_build_open,), pi_b = timed(one, 1)
            assert pi_b.tolist() == pi0.tolist()
            table_in_place = ctx.info(1) > 0
            ts_a, pi = timed(one, args.open_steps)
            hts, pi_h = timed(lambda: (lambda p_: (pcdl.check_proof(ctx, C, d, zw[0], v, p_), p_)[1])(pcdl.open(ctx, [1], coeffs, C, d, zw[0])), args.open_steps)
            hdt = median(hts)
            assert pi.tolist() == pi_h.tolist() == pi0.tolist()

            def shim_open():  # integration/harness.c (b): the loop of the patched pcdl::open, non-hiding branch (pcdl.rs:165-231)
                lg_ = args.log_n
                v2, xi, Hp = h._lib.open_start(C, zw[0], v)
                st_ = h._lib.Ipa(ctx, n, coeffs, zw[0])
                pb = np.zeros(h._lib.load().halo_proof_words(lg_), dtype=np.uint64)
                pb[1] = lg_
                for r_ in range(lg_):
                    rec = st_.round_lr_partial()
                    L_, R_, xi_n, xi_i = h._lib.open_combine(rec, Hp, xi)
                    pb[2 + 12 * r_: 14 + 12 * r_] = L_
                    pb[2 + 12 * lg_ + 12 * r_: 14 + 12 * lg_ + 12 * r_] = R_
                    xi = xi_n
                    st_.round_fold(xi_n, xi_i)
                U_, c_ = st_.finish()
                st_.close()
                o_ = 2 + 24 * lg_
                pb[o_: o_ + 12] = U_; pb[o_ + 12: o_ + 16] = c_
                pb[o_ + 16: o_ + 28] = pi0[o_ + 16: o_ + 28]  # C_bar = None: the library's encoding of the point at infinity
                return pb

            sts, pi_s = timed(shim_open, max(args.open_steps // 2, 3))
            assert pi_s.tolist() == pi0.tolist(), "the shim's round-by-round open differs from halo_pcdl_open"
            dropin["open_by_rounds_ms"] = median(sts) * 1e3
            dropin["open_one_call_host_polynomial_ms"] = median([t for t in hts]) * 1e3
            dropin["open_note"] = "open_by_rounds: 1 + 3 lg n library calls from Python ctypes (a Rust caller's calls cost less), fold table as the " \
                                  "open leg left it (in place: %s); the *_host_polynomial figure includes the check" % table_in_place
            # AccumulatedHPolys::get_poly for two instances (acc.rs:85-94 through ffi::h_accumulate): h_0 + a_1 h_1 + a_2 h_2, n coefficients back
            xis2 = np.ascontiguousarray(co[: 2 * (args.log_n + 1)].reshape(2, args.log_n + 1, 4))
            al2 = np.ascontiguousarray(co[100:102])
            h02 = np.ascontiguousarray(co[200:202])
            acc1 = ctx.h_accumulate(h02, xis2, al2)
            acc2 = np.zeros_like(acc1)
            hts2, hts3 = [], []
            for _ in range(5):
                t0 = time.perf_counter()
                ctx.h_accumulate(h02, xis2, al2, out=acc2)
                hts2.append(time.perf_counter() - t0)
                t0 = time.perf_counter()
                acc3 = ctx.h_accumulate(h02, xis2, al2)
                hts3.append(time.perf_counter() - t0)
                same3 = bool((acc3 == acc1).all())
                del acc3
            assert bool((acc1 == acc2).all()) and same3
            dropin["h_accumulate_two_instances_ms"] = median(hts2) * 1e3
            dropin["h_accumulate_two_instances_fresh_output_ms"] = median(hts3) * 1e3
            dropin["h_accumulate_note"] = "32 MiB of coefficients come back to pageable host memory: into a buffer the caller reuses / into a freshly " \
                                          "allocated one (its pages are faulted in and pinned during the copy: the operating system's time, not the GPU's)"
            ts_b, _ = timed(one, args.open_steps)  # second set, after the host-path runs
            pooled = sorted(ts_a + ts_b)
            odt = median(pooled)  # ONE median over both sets of samples (not the better of two medians)
            plain_dt = median(ts_plain)
            # check alone, and the kernels of one open with event brackets around every launch
            cts, _ = timed(lambda: pcdl.check_proof(ctx, C, d, zw[0], v, pi), max(args.open_steps, 3))
            ctx.prof_enable(1); ctx.prof_reset()
            pcdl.open_dev(ctx, [1], d_co.data_ptr(), n, C, d, zw[0])
            oprof = ctx.prof()
            ctx.prof_enable(0)
            fold_ms = sum(ms for k, (ms, cnt) in oprof.items() if k.startswith("k_fold_points"))
            by_time = sorted(((ms, k, cnt) for k, (ms, cnt) in oprof.items()), reverse=True)
            # Roofline of the open's dominant kernel = its longest single launch: the first two-level fold of G (pcdl.rs:216-219 applied
            # twice, n -> n / 4 points).  ALGORITHMIC bytes of that fold: n points of 64 B read, n / 4 written = 80 n.
            dom_o = max(((ms / max(cnt, 1), k) for k, (ms, cnt) in oprof.items() if k.startswith("k_fold_points")), default=(0.0, None))
            is_fold = dom_o[1] is not None
            if not is_fold:  # (sizes at which the key is never folded, n <= 2^14: the longest launch is a round's bucket kernel -- 96 B per point)
                dom_o = max(((ms / max(cnt, 1), k) for k, (ms, cnt) in oprof.items()), default=(0.0, None))
            o_roof = None
            if dom_o[1]:
                o_ms, o_alg = dom_o[0], (80 * n if is_fold else 96 * n)
                o_traffic, o_src = None, None
                pmc = latest_profile("pmc_open_loop.json")
                if args.log_n == 20 and table_in_place and pmc:
                    pj = json.load(open(pmc))
                    o_traffic = pj["kernels"].get("k_fold_tab4", {}).get("traffic_bytes_per_launch")
                    o_src = "static: profiles/%s (%s)" % (os.path.basename(pmc), pj.get("note", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/open_loop.py"))
                o_ach = o_alg / (o_ms * 1e-3) / 1e9 if o_ms > 0 else 0.0
                # VALU view: a table fold is ~128 mixed additions per output (2 x 22 comb entries per scalar, three scalars) of ~1143
                # v_mad_u64_u32 each (the XYZZ mixed addition of the bucket kernel), n / 4 outputs
                props_o = torch.cuda.get_device_properties(gpu)
                peak_o = props_o.multi_processor_count * 4 * 2.4e9 / 5.26
                adds_per_out = 3 * 44 + 1 if dom_o[1] == "k_fold_points4_tab" else None
                valu_o = None
                if adds_per_out and o_ms > 0:
                    wm = (n // 4) * adds_per_out * 1143 / 64
                    valu_o = {"unit": "v_mad_u64_u32 wave-instr/s", "achieved": wm / (o_ms * 1e-3), "peak": peak_o, "frac": wm / (o_ms * 1e-3) / peak_o,
                              "mixed_additions_per_output": adds_per_out}
                # (the event profiler's label -> the symbol rocprofv3 prints for the same launch)
                o_symbol = {"k_fold_points4_tab": "halo::k_fold_tab4"}.get(dom_o[1], "halo::" + dom_o[1])
                o_roof = {"bound": "hbm", "kernel": dom_o[1], "symbol": o_symbol, "achieved": o_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": o_ach / HBM_PEAK_GBS,
                          "kernel_ms": o_ms, "algorithmic_bytes": o_alg, "traffic": o_traffic, "traffic_source": o_src, "valu": valu_o,
                          "note": "integer-VALU-bound like the MSM's bucket kernel: the comb-table fold reads its 2.1 GB of table entries as "
                                  "coalesced 4 KiB gathers; see DESIGN.md 4.1"}
            gain = plain_dt - odt
            result["pcdl_open_check"] = {"value": 1.0 / odt, "unit": "open+check/s", "ms": odt * 1e3, "n": n, "hiding": False,
                                          "algorithmic_bytes": 480 * n, "hbm_roofline_frac": (480 * n / odt) / (HBM_PEAK_GBS * 1e9),
                                          "roofline": o_roof,
                                          "check_alone_ms": median(cts) * 1e3,
                                          "kernels_of_one_open_ms": {k: {"ms": round(ms, 4), "launches": cnt} for ms, k, cnt in by_time[:8]},
                                          "k_fold_points_ms_per_open": fold_ms,
                                          "fold_table_in_place": table_in_place, "fold_table_status": ctx.info(5),
                                          "fold_table_bytes": ctx.info(1), "fold_table_build_ms": ctx.info(2) / 1e3,
                                          "without_fold_table_ms": plain_dt * 1e3, "first_open_of_the_context_ms": t_first_open * 1e3,
                                          "open_that_builds_the_table_ms": t_build_open * 1e3,
                                          "fold_table_break_even_opens": (t_build_open - plain_dt) / gain if gain > 0 else None,
                                          "optional_memory_budget_bytes": ctx.info(3), "optional_memory_in_use_bytes": ctx.info(4),
                                          "end_to_end_host_polynomial_ms": hdt * 1e3,
                                          "note": "value / ms: ONE open + check at a time, fold table in place (requested explicitly: halo_set_fold_table(ctx, 1); "
                                                  "%s), polynomial resident in device memory (halo_pcdl_open_dev); without_fold_table_ms: the same with "
                                                  "halo_set_fold_table(ctx, 0); end_to_end: 32 MiB of coefficients copied from pageable host memory per open "
                                                  "(halo_pcdl_open); median of %d samples (two sets of %d, before and after the host-path runs, pooled)"
                                                  % ("in place" if table_in_place else "NOT in place: budget or memory, the figure is the generic fold's", len(pooled), args.open_steps),
                                          "samples_ms": [round(t * 1e3, 3) for t in ts_a + ts_b],
                                          "samples_without_fold_table_ms": [round(t * 1e3, 3) for t in ts_plain]}
            if args.concurrent_opens > 1:
                # THROUGHPUT: independent open + check pairs in flight, each on its own context and host thread (contexts are
                # independent by contract; ctypes releases the GIL).  An open alone leaves the GPU idle between its latency chains;
                # two in flight fill each other's gaps.  The extra contexts are CLONES (halo_ctx_clone): they share the resident
                # key, the MSM table and the fold table of the first one and own only their workspaces and IPA buffers.
                import threading
                T = args.concurrent_opens
                used_before = ctx.info(4)
                extra = [ctx.clone() for _ in range(T - 1)]
                for c_ in extra:
                    assert one(c_).tolist() == pi0.tolist()
                    one(c_)
                ctxs = [ctx] + extra
                J = max(args.open_steps, 4)
                errs = []

                def worker(c_):
                    try:
                        torch.cuda.set_device(gpu)
                        for _ in range(J):
                            p_ = one(c_)
                        if p_.tolist() != pi0.tolist():
                            errs.append("proof differs")
                    except Exception as e:  # noqa: BLE001
                        errs.append(repr(e))

                best = None
                for _ in range(2):
                    ths = [threading.Thread(target=worker, args=(c_,)) for c_ in ctxs]
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for th in ths:
                        th.start()
                    for th in ths:
                        th.join()
                    dtc = time.perf_counter() - t0
                    best = dtc if best is None else min(best, dtc)
                assert not errs, errs
                result["pcdl_open_check"]["in_flight"] = {"pairs_in_flight": T, "value": T * J / best, "unit": "open+check/s", "ms_per_pair": best / (T * J) * 1e3,
                                                          "fold_tables_in_place": [c_.info(1) > 0 for c_ in ctxs],
                                                          "optional_memory_in_use_bytes": ctx.info(4), "optional_memory_added_by_the_clones_bytes": ctx.info(4) - used_before,
                                                          "note": "%d independent open + check pairs at a time, one host thread and one context each -- clones "
                                                                  "(halo_ctx_clone) sharing ONE resident key, MSM table and fold table -- %d pairs per thread, best of 2 "
                                                                  "runs; every proof equals the single-context proof" % (T, J)}
                for c_ in extra:
                    c_.close()
            if ctx_small is not None:
                # the same at the CPU baseline's size (the reference's own D + 1, consts.rs:23): the GPU figure that stands beside it
                ns = small_n
                cs, zs = np.ascontiguousarray(co[:ns]), np.ascontiguousarray(co[n])
                Cs = pcdl.commit(ctx_small, cs, ns - 1)
                vs = ctx_small.poly_eval(cs, zs)
                ps = pcdl.open(ctx_small, [1], cs, Cs, ns - 1, zs)
                ots, cts_s = [], []
                for _ in range(max(args.open_steps, 5)):
                    t0 = time.perf_counter()
                    ps2 = pcdl.open(ctx_small, [1], cs, Cs, ns - 1, zs)
                    t1 = time.perf_counter()
                    pcdl.check_proof(ctx_small, Cs, ns - 1, zs, vs, ps2)
                    t2 = time.perf_counter()
                    ots.append(t1 - t0); cts_s.append(t2 - t1)
                assert ps2.tolist() == ps.tolist()
                gpu_side["open_small"] = {"n": ns, "coeffs": cs, "z": zs, "C": Cs, "v": vs, "proof": ps, "open_ms": median(ots) * 1e3, "check_ms": median(cts_s) * 1e3}
                gpu_side["open_full"] = {"C": C, "z": zw[0], "v": v, "proof": pi, "check_ms": median(cts) * 1e3, "open_ms": odt * 1e3 - median(cts) * 1e3}
        if args.fr_reps > 0:
            # The bandwidth-side Fr kernels, each alone: `fr_reps` back-to-back launches through the library's
            # measurement hook, HIP events around every launch (these include ~2-3 us of dispatch per launch; the rocprofv3
            # durations of the same program, tools/fr_kernels.py, are in profiles/).  Algorithmic bytes per kernel as listed.
            FR = [(0, "k_powers", 32 * n, "valu"), (1, "k_poly_eval_partial", 32 * n, "valu"), (2, "k_dot2_partial", 64 * n, "hbm"),
                  (4, "k_h_coeffs", 32 * n, "valu"), (5, "k_fold_scalars", 192 * (n // 2), "hbm"), (6, "k_axpy", 96 * n, "hbm")]
            hk = {}
            ctx.bench_fr_kernel(5, n, 3)
            for which, name, alg, bound in FR:
                ctx.prof_enable(1); ctx.prof_reset()
                ctx.bench_fr_kernel(which, n, args.fr_reps)
                ms, cnt = ctx.prof()[name]
                ctx.prof_enable(0)
                hk[name] = {"ms": ms / cnt, "algorithmic_bytes": alg, "achieved": alg / (ms / cnt) / 1e6, "unit": "GB/s", "peak": HBM_PEAK_GBS,
                            "frac": alg / (ms / cnt) / 1e6 / HBM_PEAK_GBS, "bound": bound}
            result["hbm_kernels"] = {"n": n, "launches_each": args.fr_reps, "kernels": hk,
                                     "note": "bound = valu: one 255-bit modular product per 32 bytes moved; the bare product loop runs at 8.2 us per "
                                             "2^20 products on this chip (tools/fr29_bench.hip), i.e. 4.1 TB/s-equivalent before any load, store or "
                                             "conversion -- see DESIGN.md 4.4"}
        if args.asdl_steps > 0:
            gc.collect()
            # BASELINE configs[3], the shape of benches/acc.rs:64-98 on a short chain: K x (random_instance + prover), K x verifier,
            # one decider (tests/test_gpu_pcdl_acc.py runs the full 64-step chain)
            from halo_accumulation_amd import acc as A

            def chain(c_, d_, steps, seed):
                rng_a = [seed]
                accs, qss, acc_ = [], [], None
                t0 = time.perf_counter()
                for _ in range(steps):
                    q = A.random_instance(c_, rng_a, d_)
                    qs = [q] if acc_ is None else [A.instance_from_accumulator(c_, acc_, d_), q]
                    acc_ = A.prover(c_, rng_a, d_, qs)
                    accs.append(acc_); qss.append(qs)
                t_chain = time.perf_counter() - t0
                t0 = time.perf_counter()
                for a_, qs in zip(accs, qss):
                    A.verifier(c_, d_, qs, a_)
                t_ver = time.perf_counter() - t0
                t0 = time.perf_counter()
                A.decider(c_, accs[-1])
                return t_chain / steps, t_ver / steps, time.perf_counter() - t0, accs

            t_c, t_v, t_d, _ = chain(ctx, n - 1, args.asdl_steps, 0x48414C4F00000004)
            result["asdl_chain"] = {"steps": args.asdl_steps, "n": n, "instance_plus_prover_ms_each": t_c * 1e3,
                                    "verifier_ms_each": t_v * 1e3, "decider_ms": t_d * 1e3, "all_accepted": True,
                                    "fold_table_in_place": ctx.info(1) > 0}
            if args.concurrent_opens > 1:
                # The same chain with the work of a step on TWO host threads: random_instance for step k + 1 (one hiding open) on a
                # clone of the context while acc::prover of step k (the other hiding open) runs on the context itself -- the two
                # are independent (benches/acc.rs:76-98 runs them one after the other on one thread).  Each thread has its own
                # SplitMix64 stream, so the accumulators differ from the sequential chain's; every one is verified and the last decided.
                import threading
                K = args.asdl_steps
                cl = ctx.clone()
                insts, ready = [None] * K, [threading.Event() for _ in range(K)]
                errs2 = []

                def gen():
                    try:
                        torch.cuda.set_device(gpu)
                        rng_g = [0x48414C4F00000014]
                        for k in range(K):
                            insts[k] = A.random_instance(cl, rng_g, n - 1)
                            ready[k].set()
                    except Exception as e:  # noqa: BLE001
                        errs2.append(repr(e))
                        for ev in ready:
                            ev.set()

                accs2, qss2, acc2 = [], [], None
                rng_p = [0x48414C4F00000024]
                A.random_instance(cl, [1], n - 1)  # (the clone's first open allocates its IPA buffers: untimed)
                tg = threading.Thread(target=gen)
                t0 = time.perf_counter()
                tg.start()
                for k in range(K):
                    ready[k].wait()
                    if errs2:
                        break
                    qs = [insts[k]] if acc2 is None else [A.instance_from_accumulator(ctx, acc2, n - 1), insts[k]]
                    acc2 = A.prover(ctx, rng_p, n - 1, qs)
                    accs2.append(acc2); qss2.append(qs)
                tg.join()
                t_pipe = (time.perf_counter() - t0) / K
                assert not errs2, errs2
                for a_, qs in zip(accs2, qss2):
                    A.verifier(ctx, n - 1, qs, a_)
                A.decider(ctx, accs2[-1])
                cl.close()
                result["asdl_chain"]["two_threads"] = {"instance_plus_prover_ms_each": t_pipe * 1e3, "steps": K, "all_accepted": True,
                                                       "note": "random_instance of step k + 1 on a clone of the context (halo_ctx_clone) while acc::prover of step k "
                                                               "runs on the context: one resident key and one set of tables, two host threads"}
            if ctx_small is not None:
                ks = max(2, min(args.asdl_steps, 2))
                chain(ctx_small, small_n - 1, 1, 0x48414C4F00000004)  # warm-up: buffers, launch graphs
                t_c, t_v, t_d, accs_s = chain(ctx_small, small_n - 1, ks, 0x48414C4F00000004)
                gpu_side["asdl_small"] = {"n": small_n, "steps": ks, "accs": accs_s, "instance_plus_prover_ms_each": t_c * 1e3, "verifier_ms_each": t_v * 1e3,
                                          "decider_ms": t_d * 1e3}
        if args.cpu_msms > 0:
            gc.collect()
            # ---- the CPU leg: the ONLY place that touches oracle/ (the single-thread restatement of the arkworks path the reference
            # runs: "port"), compiled on this host with -march=native.  It is the checker of everything above and the reported baseline.
            import orc  # the oracle: only this cpu_baseline / bit-exactness leg uses it
            build_timings = orc.select_fastest()  # clang / gcc x portable / -march=native builds timed on this host, the fastest one kept
            gs = ctx.read_bases()
            sc_all = sc_host
            assert sc_all.tolist()[:4] == orc.rng_scalars(0x48414C4F00000002, 4)[0].tolist()  # same stream as the tests
            t0 = time.perf_counter()
            for _ in range(args.cpu_msms):
                want = orc.msm_affine(gs, sc_all)
            cpu_dt = (time.perf_counter() - t0) / args.cpu_msms
            assert out.tolist() == want.tolist(), "GPU MSM differs from the CPU restatement"
            result["bit_exact_vs_cpu"] = True
            # the same port on all the host cores this process may use: independent MSMs, one per thread (ctypes releases the
            # GIL); a reported figure next to the single-thread one, which is how the reference runs
            import threading
            try:
                T = len(os.sched_getaffinity(0))
            except Exception:
                T = os.cpu_count() or 1
            T = max(1, min(T, 64))
            box = [None] * T

            def one_cpu(k):
                box[k] = orc.msm_affine(gs, sc_all)

            t0 = time.perf_counter()
            ths = [threading.Thread(target=one_cpu, args=(k,)) for k in range(T)]
            for th in ths:
                th.start()
            for th in ths:
                th.join()
            par_dt = time.perf_counter() - t0
            assert all(b.tolist() == want.tolist() for b in box)
            build = "oracle/halo_cpu.c, %s (the fastest of the builds timed on this host: %s)" % (
                orc.BUILD_FLAGS, ", ".join("%s %.1f ms per 2^13-point MSM" % (k, v * 1e3) for k, v in sorted(build_timings.items())))
            ns_product = orc.ns_per_field_product()
            result["cpu_baseline_all_cores"] = {"value": T / par_dt, "unit": "MSM/s", "cores": T, "kind": "port", "cpu_model": cpu_model,
                                                "sample": "%d concurrent MSMs at n=2^%d, one oracle thread each" % (T, args.log_n)}
            result["cpu_baseline"] = {"value": 1.0 / cpu_dt, "unit": "MSM/s", "cores": 1, "kind": "port", "cpu_model": cpu_model, "build": build,
                                      "sample": "%d full MSM(s) at n=2^%d, oracle/halo_cpu.c msm_bigint_wnaf (c=%d), 1 thread" % (args.cpu_msms, args.log_n, (args.log_n * 69) // 100 + 2),
                                      "host_cpus": os.cpu_count(), "ns_per_field_product": ns_product,
                                      "ns_per_field_product_note": "Fq Montgomery product (4 x 64-bit limbs, ark-ff 0.5 no-carry CIOS restated), throughput over four "
                                                                   "independent chains, measured in this run on this host with the build named above"}
            if "open_small" in gpu_side:
                # pcdl::open + check (pcdl.rs:120-242,323-342) on one core at the reference's own size, the same polynomial the GPU just
                # opened: proofs compared bit for bit; then pcdl::check at the FULL size on the GPU's full-size proof (one MSM of n
                # points + the expansion of h).  A full-size CPU open is not run (minutes): stated as the extrapolation it is.
                g = gpu_side["open_small"]
                ns = g["n"]
                pp_s = orc.make_pp(np.ascontiguousarray(gs[:ns]))
                t0 = time.perf_counter()
                p_cpu, _ = orc.pcdl_open(pp_s, 1, g["coeffs"], g["C"], ns - 1, g["z"])
                t1 = time.perf_counter()
                orc.pcdl_check(pp_s, g["C"], ns - 1, g["z"], g["v"], p_cpu)
                t2 = time.perf_counter()
                assert p_cpu.tolist() == g["proof"].tolist(), "GPU proof differs from the CPU restatement's"
                cb = {"value": 1.0 / (t2 - t0), "unit": "open+check/s", "cores": 1, "kind": "port", "cpu_model": cpu_model, "build": build,
                      "n": ns, "open_ms": (t1 - t0) * 1e3, "check_ms": (t2 - t1) * 1e3,
                      "sample": "one pcdl::open + one pcdl::check at n=2^%d (consts.rs:23: the reference's D + 1), non-hiding, 1 thread" % (ns.bit_length() - 1),
                      "gpu_same_n": {"open_ms": g["open_ms"], "check_ms": g["check_ms"], "value": 1e3 / (g["open_ms"] + g["check_ms"]), "unit": "open+check/s",
                                     "note": "halo_pcdl_open + halo_pcdl_check at the same n, coefficients from host memory, one at a time"},
                      "proof_bit_exact": True}
                if "open_full" in gpu_side and n > ns:
                    f = gpu_side["open_full"]
                    pp_f = orc.make_pp(gs)
                    t0 = time.perf_counter()
                    orc.pcdl_check(pp_f, f["C"], n - 1, f["z"], f["v"], f["proof"])
                    t_chk = time.perf_counter() - t0
                    # open: every term of its cost is linear in n or n lg n (the folds: n scalar multiples in all; the 2 lg n MSMs: 2 n
                    # points in all; p(z), h): scaled by n / n_small from the measured small open -- an EXTRAPOLATION, marked as one
                    ext_open = cb["open_ms"] / 1e3 * (n / ns)
                    cb["full_size"] = {"n": n, "check_ms": t_chk * 1e3, "check_sample": "one full pcdl::check at n=2^%d on the GPU's proof (accepted), 1 thread" % args.log_n,
                                       "gpu_check_ms": f["check_ms"],
                                       "open_ms_extrapolated": ext_open * 1e3, "open_extrapolation": "measured open at n=2^%d x %d (cost linear in n); NOT measured" % (ns.bit_length() - 1, n // ns),
                                       "open_plus_check_per_s_extrapolated": 1.0 / (ext_open + t_chk), "gpu_open_ms": f["open_ms"]}
                result["pcdl_open_check"]["cpu_baseline"] = cb
            if "asdl_small" in gpu_side:
                # benches/acc.rs:64-106 on one core at the same size: K x (random_instance + prover), K x verifier, one decider; the
                # accumulators must equal the GPU's blob for blob (same SplitMix64 stream)
                g = gpu_side["asdl_small"]
                ns, ks = g["n"], g["steps"]
                pp_s = orc.make_pp(np.ascontiguousarray(gs[:ns]))
                lgs = ns.bit_length() - 1
                seed, acc_, accs_c, qss = 0x48414C4F00000004, None, [], []
                t0 = time.perf_counter()
                for _ in range(ks):
                    q, seed = orc.random_instance(pp_s, seed, ns - 1)
                    qs = [q] if acc_ is None else [np.ascontiguousarray(acc_[: orc.instance_words(lgs)]).copy(), q]
                    acc_, seed = orc.acc_prover(pp_s, seed, ns - 1, qs)
                    accs_c.append(acc_); qss.append(qs)
                t_chain = (time.perf_counter() - t0) / ks
                t0 = time.perf_counter()
                for a_, qs in zip(accs_c, qss):
                    orc.acc_verifier(pp_s, ns - 1, qs, a_)
                t_ver = (time.perf_counter() - t0) / ks
                t0 = time.perf_counter()
                orc.acc_decider(pp_s, accs_c[-1])
                t_dec = time.perf_counter() - t0
                assert all(a.tolist() == b.tolist() for a, b in zip(accs_c, g["accs"])), "GPU accumulators differ from the CPU restatement's"
                result["asdl_chain"]["cpu_baseline"] = {
                    "value": 1.0 / t_chain, "unit": "chain steps/s (random_instance + prover)", "cores": 1, "kind": "port", "cpu_model": cpu_model, "build": build, "n": ns,
                    "instance_plus_prover_ms_each": t_chain * 1e3, "verifier_ms_each": t_ver * 1e3, "decider_ms": t_dec * 1e3,
                    "sample": "%d chain steps at n=2^%d (benches/acc.rs:64-106 shape), then %d verifiers and one decider, 1 thread" % (ks, lgs, ks),
                    "gpu_same_n": {"instance_plus_prover_ms_each": g["instance_plus_prover_ms_each"], "verifier_ms_each": g["verifier_ms_each"], "decider_ms": g["decider_ms"]},
                    "accumulators_bit_exact": True}
        if ctx_small is not None and ctx_small is not ctx:
            ctx_small.close()
    if (world > 1 or force_dist) and args.open_steps > 0 and world & (world - 1) == 0 and n % world == 0:
        # BASELINE configs[2] on N GPUs: pcdl::open + check with the key, the coefficients and the z-powers placed cyclically
        # (element i on rank i mod N; sharded.ShardedOpen): per round one all-gather of 256 B per rank, no vector exchange;
        # the check's commitment to h is sharded the same way (halo_pcdl_check_partial).  The same polynomial as at N = 1
        # (seed ...03); a rank's coefficients c[r::N] are handed over in host memory at every open.
        from halo_accumulation_amd.sharded import ShardedOpen, make_allgather
        ag = make_allgather(coll_dev)
        # the collectives of the sharded open / check: libhalo_rccl.so's halo_allgather_rccl when it is built and the backend is
        # RCCL -- the C function pointer goes straight into halo_pcdl_open_sharded, no Python (and no torch) in the collective path,
        # the same symbol a Rust host links; torch.distributed only carries the communicator's 128-byte id to the ranks.
        # HALO_BENCH_NATIVE_GATHER=0 keeps the torch.distributed callback.
        native_gather = None
        from halo_accumulation_amd import rccl as hrccl
        if backend == "nccl" and hrccl.available() and os.environ.get("HALO_BENCH_NATIVE_GATHER", "1") != "0":
            # (optional path: whatever goes wrong here -- on ANY rank -- every rank falls back to the torch.distributed callback;
            # the agreement itself is a collective, so no rank is left alone with a communicator the others do not have)
            ok, box = 1, [None]
            if rank == 0:  # (rank 0 reaches the broadcast whatever happens: the others are waiting in it)
                try:
                    box[0] = hrccl.unique_id()
                except Exception as e:  # noqa: BLE001
                    sys.stderr.write("[bench] native RCCL all-gather not used: %s\n" % e)
            dist.broadcast_object_list(box, src=0)
            if box[0] is None:
                ok = 0
            else:
                try:
                    native_gather = hrccl.RcclGather(box[0], rank, world, device=gpu)
                except Exception as e:  # noqa: BLE001
                    sys.stderr.write("[bench] native RCCL all-gather not used on rank %d: %s\n" % (rank, e))
                    ok = 0
            flag = torch.tensor([ok], dtype=torch.int64, device=coll_dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0 and native_gather is not None:
                native_gather.close()
                native_gather = None
        so = ShardedOpen(h._lib, rank, world, native_gather if native_gather is not None else ag, device=gpu, always_collective=force_dist)
        sctx = so.load_key(n)
        sctx.set_fold_table(1)  # the comb table of the first fold over this rank's shard, built at the first (warm-up) open
        d_co = torch.empty((n + 2) * 4, dtype=torch.int64, device=dev)
        sctx.rng_scalars_dev(0x48414C4F00000003, n + 2, d_co.data_ptr())
        co = d_co.cpu().numpy().view(np.uint64).reshape(n + 2, 4)
        c_local, z_open = np.ascontiguousarray(co[:n][rank::world]), np.ascontiguousarray(co[n])
        del d_co
        C = h._lib.point_sum(ag(sctx.msm(c_local)))  # pcdl::commit, non-hiding: the sum of the ranks' MSMs
        for _ in range(2):
            pi, v_open = so.open(c_local, C, z_open)
            so.check(C, n - 1, z_open, v_open, pi)
        ts = []
        for _ in range(args.open_steps):
            dist.barrier()
            t0 = time.perf_counter()
            pi, v_open = so.open(c_local, C, z_open)
            so.check(C, n - 1, z_open, v_open, pi)
            ts.append(time.perf_counter() - t0)
        tmax = torch.tensor(ts, dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        ts = sorted(tmax.cpu().tolist())
        if rank == 0:
            odt = ts[len(ts) // 2]
            # the sharded proof must be the single-GPU proof: the same open on this rank's GPU alone
            full_o = h._lib.Context(urs_n=n, first_index=2, device=gpu)
            coeffs_all = np.ascontiguousarray(co[:n])
            assert C.tolist() == pcdl.commit(full_o, coeffs_all, n - 1).tolist(), "sharded commitment differs"
            want = pcdl.open(full_o, [1], coeffs_all, C, n - 1, z_open)
            same = want.tolist() == pi.tolist()
            pcdl.check_proof(full_o, C, n - 1, z_open, v_open, pi)
            full_o.close()
            # (HALO_BENCH_FORCE_DIST=1: one rank running the collectives of the N > 1 path -- kept next to the N = 1 figure)
            result["pcdl_open_check" if world > 1 else "pcdl_open_check_collective_path"] = {
                                         "value": 1.0 / odt, "unit": "open+check/s", "ms": odt * 1e3, "n": n, "hiding": False, "ranks": world,
                                         "samples_ms": [round(x * 1e3, 3) for x in ts], "proof_equals_single_gpu": same,
                                         "collectives": ("libhalo_rccl.so halo_allgather_rccl (native: ncclAllGather on the library's own communicator), %d calls"
                                                         % native_gather.calls) if native_gather is not None else "torch.distributed all_gather_into_tensor through a Python callback",
                                         "note": "sharded.ShardedOpen: cyclic shards of G, c, z-powers; per round one all-gather of 256 B per "
                                                 "rank; check = succinct check on every rank + sharded commitment to h; max over ranks "
                                                 "per repetition, median reported; coefficients handed over in host memory"}
            assert same, "sharded open differs from the single-GPU open"
        so.ctx.close()
        if native_gather is not None:
            native_gather.close()
    if world > 1 and rank == 0:
        # cross-check of the sharded results: the same MSMs, unsharded, on this rank's GPU alone
        full = ctx if window_mode else h._lib.Context(urs_n=n, first_index=2, device=gpu)
        ok = True
        for j in sorted({0, batch - 1}):
            d_all = torch.empty(n * 4, dtype=torch.int64, device=dev)
            full.rng_scalars_dev((SEED + 4 * j * n * GAMMA) & MASK, n, d_all.data_ptr())
            ok = ok and full.msm_dev(d_all.data_ptr(), n).tolist() == outs[j].tolist()
        result["sharded_equals_single_gpu"] = ok
        if not window_mode:
            full.close()
        assert ok, "sharded MSM differs from the single-GPU MSM"
    if world > 1 or force_dist:
        dist.barrier()
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(result) + "\n").encode())
    ctx.close()
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
