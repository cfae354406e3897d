"""N > 1 path of the sharded MSM on CPU: world_size 2, 3 (uneven shards) and 8 (the node size) over gloo.

The per-rank Pippenger runs on the GPU in production; here the partial comes from the oracle so
that the sharding, the all-gather and the fixed-order combine are exercised without a device."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n, q):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import halo_accumulation_amd as h
    from halo_accumulation_amd.sharded import ShardedMsm, shard_range
    import orc
    lo, hi = shard_range(n, rank, world)
    gs = orc.urs_affine(2 + lo, hi - lo)              # this rank's block of the key
    sc, _ = orc.rng_scalars(0x48414C4F00000005, n)    # every rank derives the same scalar stream
    msm = ShardedMsm(lambda: orc.msm_affine(gs, np.ascontiguousarray(sc[lo:hi])), h._lib.point_sum)
    out = msm()
    # batched form: three partials (the local one, twice, and the point at infinity) in one collective
    inf = np.array([1, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0], dtype=np.uint64)
    part = orc.msm_affine(gs, np.ascontiguousarray(sc[lo:hi]))
    batch = msm.gather_batch([part, inf, part])
    assert batch[0].tolist() == out.tolist() == batch[2].tolist()
    assert orc.point_canonical(batch[1]) is None
    # asynchronous form: two collectives in flight, finished in issue order (bench.py keeps one under the next launch)
    h1, h2 = msm.gather_start([part]), msm.gather_start([inf, part])
    assert msm.gather_finish(h1)[0].tolist() == out.tolist()
    r2 = msm.gather_finish(h2)
    assert orc.point_canonical(r2[0]) is None and r2[1].tolist() == out.tolist()
    q.put((rank, out.tolist(), (lo, hi)))
    dist.barrier()
    dist.destroy_process_group()


def _window_worker(rank, world, port, n, q):
    """window shards: every rank holds the whole key, its partial covers its scalar windows only"""
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import halo_accumulation_amd as h
    from halo_accumulation_amd.sharded import ShardedMsm, window_range
    import orc
    gs = orc.urs_affine(2, n)
    sc, _ = orc.rng_scalars(0x48414C4F00000005, n)
    c, W = 16, 16
    w0, w1 = window_range(W, rank, world)
    ints = [orc.fr_from_mont(x) for x in sc]
    mine = [((v >> (c * w0)) & ((1 << (c * (w1 - w0))) - 1)) << (c * w0) for v in ints]  # this rank's windows of every scalar
    msm = ShardedMsm(lambda: orc.msm_affine(gs, orc.scalars_to_mont(mine)), h._lib.point_sum)
    q.put((rank, msm().tolist(), (w0, w1)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_window_sharded_msm_gloo(world):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    n = 64
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_window_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = orc.msm_affine(orc.urs_affine(2, n), orc.rng_scalars(0x48414C4F00000005, n)[0]).tolist()
    ranges = sorted(r[2] for r in res)
    assert ranges[0][0] == 0 and ranges[-1][1] == 16 and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    for rank, out, _ in res:
        assert out == want, "rank %d disagrees" % rank


@pytest.mark.parametrize("world,n", [(2, 256), (3, 101), (8, 203)])
def test_sharded_msm_gloo(world, n):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    gs = orc.urs_affine(2, n)
    sc, _ = orc.rng_scalars(0x48414C4F00000005, n)
    want = orc.msm_affine(gs, sc).tolist()
    ranges = sorted(r[2] for r in res)
    assert ranges[0][0] == 0 and ranges[-1][1] == n and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    for rank, out, _ in res:
        assert out == want, "rank %d disagrees" % rank


def test_shard_range_partition():
    from halo_accumulation_amd.sharded import shard_range
    for n in (0, 1, 7, 8, 1 << 20):
        for world in (1, 2, 3, 8):
            parts = [shard_range(n, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1


def _status_worker(rank, world, port, q):
    """The collective rule of the sharded open's by-rounds driver (ShardedOpen._gather / _defer), without a device: a rank
    whose local step raises still enters the all-gather with a status word, every rank raises after that collective."""
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import halo_accumulation_amd as h
    from halo_accumulation_amd.sharded import ShardedOpen, make_allgather
    calls = [0]
    ag = make_allgather()

    def allgather(arr):
        calls[0] += 1
        return ag(arr)

    so = ShardedOpen(h._lib, rank, world, allgather)
    so._pending = None
    log = []
    # 1. nobody fails: the records come back in rank order, status stripped
    got = so._gather(3, lambda: np.array([rank, 10 * rank, 7], dtype=np.uint64))
    log.append(got.tolist() == [[r, 10 * r, 7] for r in range(world)])
    # 2. the last rank's local step raises: every rank raises after this one collective (the failing rank its own exception)
    def local():
        if rank == world - 1:
            raise ValueError("device lost on rank %d" % rank)
        return np.zeros(5, dtype=np.uint64)
    before = calls[0]
    try:
        so._gather(5, local)
        log.append("returned")
    except ValueError as e:
        log.append(("ValueError", str(e), calls[0] - before))
    except h._lib.HaloError as e:
        log.append(("HaloError", "rank %d" % (world - 1) in str(e), calls[0] - before))
    # 3. a failure between two collectives (a fold) rides into the next one
    so._defer(lambda: (_ for _ in ()).throw(RuntimeError("fold failed")) if rank == 0 else None)
    before = calls[0]
    try:
        so._gather(2, lambda: np.ones(2, dtype=np.uint64))
        log.append("returned")
    except RuntimeError as e:
        log.append((type(e).__name__, calls[0] - before))
    # 4. and the group is still in step afterwards
    got = so._gather(1, lambda: np.array([rank + 1], dtype=np.uint64))
    log.append(got.reshape(-1).tolist() == list(range(1, world + 1)))
    q.put((rank, log))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_failing_rank_reaches_the_collective_and_every_rank_raises(world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_status_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = dict(q.get(timeout=120) for _ in range(world))
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    for rank in range(world):
        log = res[rank]
        assert log[0] is True and log[3] is True
        if rank == world - 1:
            assert log[1] == ("ValueError", "device lost on rank %d" % rank, 1)
        else:
            assert log[1] == ("HaloError", True, 1)
        assert log[2] == ("RuntimeError" if rank == 0 else "HaloError", 1)
