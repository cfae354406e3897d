#!/usr/bin/env python3
"""Headline benchmark: Pippenger MSMs/sec at n = 2^20 (BASELINE.json configs[1]) on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--log-n 20] [--open-steps J]

A step = one MSM over n random scalars (resident in HBM) and the URS bases G_0..G_{n-1}
(derived on the GPU by the reference's main.rs rule).  N > 1 (launched by torch.distributed.run,
one rank per GPU): the SAME n-point MSMs are sharded -- by Pippenger windows (default: every rank keeps
the key and the scalars and does 1/N of the bucket work) or by base/scalar index (--shard index) -- each
rank reduces its share to one point and the partials are all-gathered over RCCL and summed (strong scaling).
Prints ONE JSON line on rank 0.
"""
import argparse
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


def main():
    # stdout carries exactly ONE line, the JSON: everything else any library prints there during the run (RCCL announces
    # its version on stdout when a communicator is created) is sent to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
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
    ap.add_argument("--open-steps", type=int, default=5, help="PCDL open+check repetitions at N=1 (0 = skip)")
    ap.add_argument("--cpu-msms", type=int, default=2, help="oracle MSMs timed for cpu_baseline at N=1 (0 = skip)")
    ap.add_argument("--min-seconds", type=float, default=1.0, help="repeat the K-step timed region until this much time is covered; the median repetition is reported")
    ap.add_argument("--one-process", action="store_true",
                    help="--gpus N from ONE process (no torch.distributed): a multi-device context (halo_ctx_create_urs_multi), one "
                         "index-block shard per GPU, partial points added on the host")
    ap.add_argument("--devices", default="", help="--one-process: comma-separated device ids (default 0..N-1; ids may repeat for a rehearsal)")
    ap.add_argument("--host-steps", type=int, default=8, help="MSMs with the scalars in host memory (halo_msm and its begin/end halves) at N=1 (0 = skip)")
    ap.add_argument("--fr-reps", type=int, default=20, help="back-to-back launches of each bandwidth-side Fr kernel at N=1 (0 = skip)")
    ap.add_argument("--asdl-steps", type=int, default=8, help="ASDL chain steps (random_instance + prover + verifier, then one decider) at N=1 (0 = skip)")
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
    one_proc = args.one_process
    assert world == (1 if one_proc else args.gpus), "launch with torch.distributed.run --nproc-per-node == --gpus (or --one-process)"
    devices = None
    if one_proc:
        devices = [int(x) for x in args.devices.split(",")] if args.devices else [k % max(torch.cuda.device_count(), 1) for k in range(args.gpus)]
        assert len(devices) == args.gpus
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
        pmc = os.path.join(ROOT, "profiles", "r03_pmc_traffic.json")
        if world == 1 and args.log_n == 20 and os.path.exists(pmc):
            pj = json.load(open(pmc))
            traffic = pj["kernels"].get(dom, {}).get("traffic_bytes_per_launch")
            traffic_source = "static: profiles/r03_pmc_traffic.json (%s)" % pj.get("note", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes")
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
            sq = os.path.join(ROOT, "profiles", "r03_sq_msm.json")
            if world == 1 and args.log_n == 20 and os.path.exists(sq):
                k = json.load(open(sq))["kernels"].get(dom)
                if k:
                    valu["issue_slots_used"] = k["valu_per_quad"]
                    valu["clock_ghz_during_kernel"] = k["clock_ghz"]
                    valu["wave_cycles_parked"] = k["wave_parked"]
                    valu["issue_source"] = "static: profiles/r03_sq_msm.json (rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAIT_ANY ..., one MSM in flight)"
        result = {
            "metric": "MSMs/sec (Pippenger, Pallas, n=2^%d random scalars/URS points, bit-exact vs CPU)" % args.log_n,
            "value": args.steps / dt, "unit": "MSM/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "timed_region": {"repetitions": len(reps), "reported": "median", "seconds_each": [round(x, 6) for x in reps[:32]]},
            "dtype": "u32x8 (256-bit Montgomery integer)", "data": "synthetic",
            "config": {"workload": "Pippenger MSM n=2^%d, bases = URS G_i by main.rs rule, scalars SplitMix64 seed 0x48414C4F00000002" % args.log_n,
                       "sharding": ("one process, multi-device context: %d index-block shards on devices %s, partial points added on the host "
                                    "(no collective)" % (len(devices), devices)) if one_proc and args.gpus > 1 else "single GPU" if world == 1 else
                                   ("Pippenger windows split over the ranks (key + scalars replicated), %s all-gather of 96 B partials" % coll_name if window_mode
                                    else "block index shard per rank + %s all-gather of 96 B partials" % coll_name),
                       "collective_backend": backend_name, "world_size": world_reported,
                       "distinct_gpus": len(set(devices)) if one_proc else distinct_gpus,
                       "msms_per_launch": batch, "launches_in_flight": args.depth,
                       "window_bits": "a rank's block of >= 2^20 points: 20 with fixed-base tables over the context's key (13 windows, one set "
                                      "of 2^19 buckets); index blocks of 2^17..2^19 points: 17 (15 windows, 2^16 buckets per MSM of a batch); "
                                      "window shards: 16 (16-window general plan)"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
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
        # bit-exactness in the same run + CPU baseline (oracle = single-thread port of the arkworks path)
        gs = ctx.read_bases()
        cpu_model = "unknown"
        try:
            cpu_model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
        except Exception:
            pass
        sc_host = np.ascontiguousarray(d_sets[0].cpu().numpy().view(np.uint64).reshape(n, 4))
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
            result["end_to_end_host_scalars"] = {"value": 1.0 / h2d_dt, "unit": "MSM/s", "ms": h2d_dt * 1e3,
                                                 "pipelined_value": 1.0 / pipe_dt, "pipelined_ms": pipe_dt * 1e3,
                                                 "h2d_copy_alone_ms": copy_dt * 1e3, "h2d_copy_GBs": sc_host.nbytes / copy_dt / 1e9,
                                                 "note": "halo_msm: %d MiB of scalars copied from pageable host memory each call, one MSM in flight (a call is "
                                                         "the copy + the latency of one MSM); pipelined: halo_msm_begin/_end on %d slots, the next copy under the "
                                                         "current kernels; h2d_copy_alone: the same buffer through torch, for scale" % (sc_host.nbytes >> 20, D)}
        if args.cpu_msms > 0:
            import orc  # the oracle: only this cpu_baseline / bit-exactness leg uses it
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

            def one(k):
                box[k] = orc.msm_affine(gs, sc_all)

            t0 = time.perf_counter()
            ths = [threading.Thread(target=one, args=(k,)) for k in range(T)]
            for th in ths:
                th.start()
            for th in ths:
                th.join()
            par_dt = time.perf_counter() - t0
            assert all(b.tolist() == want.tolist() for b in box)
            result["cpu_baseline_all_cores"] = {"value": T / par_dt, "unit": "MSM/s", "cores": T, "kind": "port", "cpu_model": cpu_model,
                                                "sample": "%d concurrent MSMs at n=2^%d, one oracle thread each" % (T, args.log_n)}
            result["cpu_baseline"] = {"value": 1.0 / cpu_dt, "unit": "MSM/s", "cores": 1, "kind": "port", "cpu_model": cpu_model,
                                      "sample": "%d full MSM(s) at n=2^%d, oracle/halo_cpu.c msm_bigint_wnaf (c=%d), 1 thread" % (args.cpu_msms, args.log_n, (args.log_n * 69) // 100 + 2),
                                      "host_cpus": os.cpu_count()}
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
            # warm-up: the first full-size open of a context asks for the fold table's memory on a helper thread, the first
            # later open that finds it there builds the table (foldtab.hip); the timed opens run with the table in place
            pi = pcdl.open_dev(ctx, [1], d_co.data_ptr(), n, C, d, zw[0])
            for _ in range(40):
                pi = pcdl.open_dev(ctx, [1], d_co.data_ptr(), n, C, d, zw[0])
                if ctx.info(1) or n < (1 << 18) or n > (1 << 21):
                    break
                time.sleep(0.1)
            pi = pcdl.open_dev(ctx, [1], d_co.data_ptr(), n, C, d, zw[0])
            v = ctx.poly_eval(coeffs, zw[0])

            def timed(fn):
                torch.cuda.synchronize()
                ts = []
                for _ in range(args.open_steps):
                    t0 = time.perf_counter()
                    p_ = fn()
                    pcdl.check_proof(ctx, C, d, zw[0], v, p_)
                    ts.append(time.perf_counter() - t0)
                return sorted(ts)[len(ts) // 2], p_, ts

            _, pi, ts_a = timed(lambda: pcdl.open_dev(ctx, [1], d_co.data_ptr(), n, C, d, zw[0]))
            hdt, pi_h, _ = timed(lambda: pcdl.open(ctx, [1], coeffs, C, d, zw[0]))
            assert pi.tolist() == pi_h.tolist()
            _, _, ts_b = timed(lambda: pcdl.open_dev(ctx, [1], d_co.data_ptr(), n, C, d, zw[0]))  # second set, after the host path
            pooled = sorted(ts_a + ts_b)
            odt = pooled[len(pooled) // 2]  # ONE median over both sets of samples (not the better of two medians)
            ctx.prof_enable(2); ctx.prof_reset()  # one more open with event brackets around the fold kernels
            pcdl.open_dev(ctx, [1], d_co.data_ptr(), n, C, d, zw[0])
            prof = ctx.prof()
            ctx.prof_enable(0)
            fold_ms = sum(ms for k, (ms, cnt) in prof.items() if k.startswith("k_fold_points"))
            result["pcdl_open_check"] = {"value": 1.0 / odt, "unit": "open+check/s", "ms": odt * 1e3, "n": n, "hiding": False,
                                          "algorithmic_bytes": 480 * n, "hbm_roofline_frac": (480 * n / odt) / (HBM_PEAK_GBS * 1e9),
                                          "k_fold_points_ms_per_open": fold_ms,
                                          "fold_table_bytes": ctx.info(1), "fold_table_build_ms": ctx.info(2) / 1e3,
                                          "end_to_end_host_polynomial_ms": hdt * 1e3,
                                          "note": "value: polynomial resident in device memory (halo_pcdl_open_dev); end_to_end: 32 MiB of "
                                                  "coefficients copied from pageable host memory per open (halo_pcdl_open); median of %d samples "
                                                  "(two sets of %d, before and after the host-path runs, pooled)" % (len(pooled), args.open_steps),
                                          "samples_ms": [round(t * 1e3, 3) for t in ts_a + ts_b]}
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
            # BASELINE configs[3], the shape of benches/acc.rs:64-98 on a short chain: K x (random_instance + prover), K x verifier,
            # one decider (tests/test_gpu_pcdl_acc.py runs the full 64-step chain)
            from halo_accumulation_amd import acc as A
            d = n - 1
            rng_a = [0x48414C4F00000004]
            accs, qss, acc_ = [], [], None
            t0 = time.perf_counter()
            for _ in range(args.asdl_steps):
                q = A.random_instance(ctx, rng_a, d)
                qs = [q] if acc_ is None else [A.instance_from_accumulator(ctx, acc_, d), q]
                acc_ = A.prover(ctx, rng_a, d, qs)
                accs.append(acc_); qss.append(qs)
            t_chain = time.perf_counter() - t0
            t0 = time.perf_counter()
            for a_, qs in zip(accs, qss):
                A.verifier(ctx, d, qs, a_)
            t_ver = time.perf_counter() - t0
            t0 = time.perf_counter()
            A.decider(ctx, accs[-1])
            t_dec = time.perf_counter() - t0
            result["asdl_chain"] = {"steps": args.asdl_steps, "n": n, "instance_plus_prover_ms_each": t_chain / args.asdl_steps * 1e3,
                                    "verifier_ms_each": t_ver / args.asdl_steps * 1e3, "decider_ms": t_dec * 1e3, "all_accepted": True}
    if (world > 1 or force_dist) and args.open_steps > 0 and world & (world - 1) == 0 and n % world == 0:
        # BASELINE configs[2] on N GPUs: pcdl::open + check with the key, the coefficients and the z-powers placed cyclically
        # (element i on rank i mod N; sharded.ShardedOpen): per round one all-gather of 256 B per rank, no vector exchange;
        # the check's commitment to h is sharded the same way (halo_pcdl_check_partial).  The same polynomial as at N = 1
        # (seed ...03); a rank's coefficients c[r::N] are handed over in host memory at every open.
        from halo_accumulation_amd.sharded import ShardedOpen, make_allgather
        ag = make_allgather(coll_dev)
        so = ShardedOpen(h._lib, rank, world, ag, device=gpu, always_collective=force_dist)
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
                                         "note": "sharded.ShardedOpen: cyclic shards of G, c, z-powers; per round one all-gather of 256 B per "
                                                 "rank; check = succinct check on every rank + sharded commitment to h; max over ranks "
                                                 "per repetition, median reported; coefficients handed over in host memory"}
            assert same, "sharded open differs from the single-GPU open"
        so.ctx.close()
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
