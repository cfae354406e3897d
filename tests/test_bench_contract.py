"""bench.py contract: one JSON line with the driver's keys plus the roofline and cpu_baseline objects."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_emits_contract_json():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--log-n", "14", "--steps", "6", "--warmup", "1",
                          "--cpu-msms", "1", "--open-steps", "1", "--asdl-steps", "2", "--min-seconds", "0.05"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "bench.py must print exactly ONE line on stdout"
    r = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in r, k
    assert r["n_gpus"] == 1 and r["steps"] == 6 and r["warmup"] == 1 and r["higher_is_better"] is True and r["vs_baseline"] is None
    assert r["data"] == "synthetic" and "workload" in r["config"] and "model" not in r["config"]
    assert r["value"] > 0 and abs(r["value"] * r["ms_per_step"] / 1e3 - 1.0) < 1e-6
    rf = r["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and rf["achieved"] > 0
    cb = r["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("port", "reference") and cb["cores"] == 1 and cb["value"] > 0
    assert r["bit_exact_vs_cpu"] is True
    assert r["pcdl_open_check"]["value"] > 0 and r["pcdl_open_check"]["end_to_end_host_polynomial_ms"] > 0
    assert "fold_table_bytes" in r["pcdl_open_check"]
    # the second half of the metric carries its own roofline and CPU baseline, the MSM a variable-base block
    oc = r["pcdl_open_check"]
    for k in ("without_fold_table_ms", "first_open_of_the_context_ms", "fold_table_in_place", "check_alone_ms", "roofline", "cpu_baseline"):
        assert k in oc, k
    assert oc["without_fold_table_ms"] > 0 and oc["roofline"]["kernel"].startswith("k_") and oc["roofline"]["achieved"] > 0
    assert abs(oc["roofline"]["frac"] - oc["roofline"]["achieved"] / 8000.0) < 1e-12 and "traffic" in oc["roofline"]
    ocb = oc["cpu_baseline"]
    assert ocb["kind"] == "port" and ocb["cores"] == 1 and ocb["value"] > 0 and ocb["proof_bit_exact"] is True and ocb["gpu_same_n"]["value"] > 0
    assert ocb["unit"] == "open+check/s" and "sample" in ocb and "-march=native" in ocb["build"] and "-march=x86-64-v2" in ocb["build"]
    assert oc["in_flight"]["pairs_in_flight"] == 2 and oc["in_flight"]["value"] > 0 and oc["in_flight"]["optional_memory_added_by_the_clones_bytes"] == 0
    assert r["asdl_chain"]["two_threads"]["all_accepted"] is True and r["asdl_chain"]["two_threads"]["instance_plus_prover_ms_each"] > 0
    acb = r["asdl_chain"]["cpu_baseline"]
    assert acb["kind"] == "port" and acb["cores"] == 1 and acb["value"] > 0 and acb["accumulators_bit_exact"] is True
    assert acb["gpu_same_n"]["instance_plus_prover_ms_each"] > 0 and acb["decider_ms"] > 0
    vb = r["variable_base"]
    assert vb["value"] > 0 and vb["roofline"]["kernel"] == "k_msm_accumulate" or vb["roofline"]["kernel_ms"] == 0  # (2^14: the small pipeline has no k_msm_accumulate)
    assert abs(vb["roofline"]["frac"] - vb["roofline"]["achieved"] / 8000.0) < 1e-12 and "traffic" in vb["roofline"]
    assert r["timed_region"]["repetitions"] >= 1 and r["timed_region"]["reported"] == "median"
    assert r["end_to_end_host_scalars"]["value"] > 0 and r["asdl_chain"]["all_accepted"] is True
    assert "cpu_model" in cb and r["cpu_baseline_all_cores"]["cores"] >= 1
    assert r["solo_latency_ms"] > 0 and r["context_setup_ms"]["first_msm_incl_table_build"] > 0 and "table_bytes" in r
    assert r["end_to_end_host_scalars"]["pipelined_value"] > 0
    hk = r["hbm_kernels"]["kernels"]
    assert set(hk) == {"k_powers", "k_poly_eval_partial", "k_dot2_partial", "k_h_coeffs", "k_fold_scalars", "k_axpy"}
    assert all(v["achieved"] > 0 and abs(v["frac"] - v["achieved"] / 8000.0) < 1e-12 for v in hk.values())


@pytest.mark.gpu
def test_bench_two_ranks_report_msms_and_the_sharded_open():
    """N = 2 as the driver launches it (one process per rank; gloo so that both ranks can share the one GPU of the test box):
    one JSON line from rank 0 with the whole-job MSM rate and open + check over the cyclically sharded key, whose proof the
    bench itself compares with the single-GPU open."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HALO_BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--log-n", "14", "--steps", "8", "--warmup", "2",
                          "--open-steps", "2", "--min-seconds", "0"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["value"] > 0 and r["sharded_equals_single_gpu"] is True
    oc = r["pcdl_open_check"]
    assert oc["ranks"] == 2 and oc["value"] > 0 and oc["proof_equals_single_gpu"] is True and len(oc["samples_ms"]) == 2


@pytest.mark.gpu
def test_bench_collective_path_over_rccl_with_one_rank():
    """The N > 1 code path with its collectives on RCCL (backend "nccl"), as far as one GPU can take it: one rank that still
    runs every all-gather of the sharded MSM and of the sharded open + check (HALO_BENCH_FORCE_DIST=1)."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HALO_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    env.pop("HALO_BENCH_BACKEND", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--log-n", "14", "--steps", "6", "--warmup", "1", "--cpu-msms", "0",
                          "--open-steps", "2", "--asdl-steps", "0", "--host-steps", "0", "--fr-reps", "0", "--min-seconds", "0"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["config"]["collective_backend"] == "nccl" and r["value"] > 0
    oc = r["pcdl_open_check_collective_path"]
    assert oc["ranks"] == 1 and oc["proof_equals_single_gpu"] is True and oc["value"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [["--devices", "0,0"], []])
def test_bench_gpus_2_runs_without_a_launcher(extra):
    """`python bench.py --gpus N` started plainly (no torch.distributed.run, WORLD_SIZE unset) must not die on
    an assert: it takes the one-process path (multi-device context over devices 0..N-1) and says so; with one GPU on the box
    the device ids repeat and the line says REHEARSAL."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "HALO_BENCH_BACKEND", "HALO_BENCH_FORCE_DIST")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--log-n", "17", "--steps", "8", "--warmup", "2",
                          "--min-seconds", "0"] + extra, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["value"] > 0 and r["sharded_equals_single_gpu"] is True
    sh = r["config"]["sharding"]
    assert "one process, multi-device context" in sh and "started without a launcher" in sh
    import torch
    if torch.cuda.device_count() < 2 or extra:
        assert "REHEARSAL" in sh and r["config"]["distinct_gpus"] == 1


def test_bench_does_not_touch_the_oracle_outside_the_cpu_leg():
    src = open(os.path.join(ROOT, "bench.py")).read()
    head, _, tail = src.partition("if args.cpu_msms > 0:")
    assert "import orc" not in head and "orc." not in head, "only the cpu_baseline leg may use oracle/"
    assert "import orc" in tail
