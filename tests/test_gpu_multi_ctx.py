"""Multi-device contexts (halo_ctx_create_urs_multi / halo_ctx_create_multi): one process, one shard context per device id
over that device's index block of the key; MSMs over the key fan out and the partial points are added on the host.

The GPU box has one MI355X: the same device id is passed 2 / 4 / 8 times -- every code path of a real multi-GPU node runs
(per-shard contexts, streams, tables, helper threads, block-order combine) except the peer copy between distinct devices,
which degenerates to reading the scalars in place.  Results must equal the plain one-device context's, which the other
tests pin on the oracle."""
import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hal():
    import halo_accumulation_amd as h
    return h._lib


@pytest.mark.parametrize("P", [2, 3, 8])
def test_multi_ctx_small_key_matches_oracle(hal, P):
    n = 1 << 17
    c = hal.Context(urs_n=n, devices=[0] * P)
    try:
        assert c.n_devices == P and c.size == n
        gs = c.read_bases()
        sc, _ = orc.rng_scalars(0x3317 + P, n)
        want = orc.msm_affine(gs, sc)
        assert c.msm(sc).tolist() == want.tolist()                    # host scalars: per-shard copies on helper threads
        # a stretch that starts and ends inside blocks, and a short one (below the fan-out threshold of the library's own MSMs)
        off, m = 12345 * 4, (1 << 16) + 4444
        assert c.msm(sc[:m], off=off).tolist() == orc.msm_affine(gs[off:off + m], sc[:m]).tolist()
        assert c.msm(sc[:1000], off=7).tolist() == orc.msm_affine(gs[7:1007], sc[:1000]).tolist()
        # asynchronous halves, several slots in flight
        for k in range(3):
            c.msm_begin(k, sc[k * 8:], off=k * 8)
        got = [c.msm_end(k) for k in range(3)]
        for k in range(3):
            assert got[k].tolist() == orc.msm_affine(gs[k * 8:], sc[k * 8:]).tolist()
        with pytest.raises(hal.HaloError):
            c.msm_end(0)                                              # nothing in flight
    finally:
        c.close()


@pytest.mark.parametrize("P", [2, 4, 8])
def test_multi_ctx_2_20_equals_single_context(hal, P):
    """BASELINE config 2 through a P-shard context: equal to the plain context's point, device-resident scalars (read in
    place by every shard on this box), begin/end on all four slots, and the pcdl-level commit (the library's own MSM over
    the key fans out too)."""
    import torch
    from halo_accumulation_amd import pcdl
    n = 1 << 20
    d = torch.empty(n * 4, dtype=torch.int64, device="cuda")
    one = hal.Context(urs_n=n)
    try:
        one.rng_scalars_dev(0x48414C4F00000002, n, d.data_ptr())
        torch.cuda.synchronize()
        want = one.msm_dev(d.data_ptr(), n)
        want_off = one.msm_dev(d.data_ptr(), n - 4096, off=2048)
        want_commit = pcdl.commit_dev(one, d.data_ptr(), n, n - 1)
    finally:
        one.close()
    c = hal.Context(urs_n=n, devices=[0] * P)
    try:
        assert c.msm_dev(d.data_ptr(), n).tolist() == want.tolist()
        assert c.msm_dev(d.data_ptr(), n - 4096, off=2048).tolist() == want_off.tolist()
        for slot in range(4):
            c.msm_dev_begin(slot, d.data_ptr(), n)
        assert all(c.msm_dev_end(slot).tolist() == want.tolist() for slot in range(4))
        assert pcdl.commit_dev(c, d.data_ptr(), n, n - 1).tolist() == want_commit.tolist()
        sc_host = np.ascontiguousarray(d.cpu().numpy().view(np.uint64).reshape(n, 4))
        assert c.msm(sc_host).tolist() == want.tolist()
    finally:
        c.close()


def test_multi_ctx_host_scalars_in_stretches_per_shard(hal):
    """halo_msm on a multi-device context with shard blocks of 2^20 points: every shard's helper thread runs its block through the
    host-scalar path of a plain context (multi.hip multi_host_run -> abi.hip msm_host_run: its copy in stretches under its own
    kernels).  Equal to the plain context's point; a stretch of the key that cuts through both blocks; pcdl::commit with host
    coefficients shorter than d + 1 (zero-padded on the device)."""
    import torch
    from halo_accumulation_amd import pcdl
    n = 1 << 21
    d = torch.empty(n * 4, dtype=torch.int64, device="cuda")
    one = hal.Context(urs_n=n)
    try:
        one.rng_scalars_dev(0x48414C4F00000005, n, d.data_ptr())
        torch.cuda.synchronize()
        want = one.msm_dev(d.data_ptr(), n)
        want_mid = one.msm_dev(d.data_ptr(), n - 8192, off=4096)
        want_commit = pcdl.commit_dev(one, d.data_ptr(), n - 5, n - 1)
    finally:
        one.close()
    sc = np.ascontiguousarray(d.cpu().numpy().view(np.uint64).reshape(n, 4))
    c = hal.Context(urs_n=n, devices=[0, 0])
    try:
        assert c.msm_dev(d.data_ptr(), n).tolist() == want.tolist()     # (every shard builds its c = 20 table here)
        for _ in range(3):
            assert c.msm(sc).tolist() == want.tolist()
        assert c.msm(np.ascontiguousarray(sc[: n - 8192]), off=4096).tolist() == want_mid.tolist()
        assert pcdl.commit(c, np.ascontiguousarray(sc[: n - 5]), n - 1).tolist() == want_commit.tolist()
    finally:
        c.close()


@pytest.mark.parametrize("P,batch", [(2, 2), (4, 4), (8, 8), (3, 5)])
def test_multi_ctx_batched_launches(hal, P, batch, monkeypatch):
    """halo_msm_dev_batch_begin/_end on a multi-device context: every shard runs its stretch of all members as one batched
    launch; member b equals the plain context's MSM over scalar set b (also with the peer-copy path forced)"""
    import torch
    n = 1 << 19
    ds = [torch.empty(n * 4, dtype=torch.int64, device="cuda") for _ in range(batch)]
    one = hal.Context(urs_n=n)
    try:
        for b, d in enumerate(ds):
            one.rng_scalars_dev(0x5E7 + 97 * b, n, d.data_ptr())
        torch.cuda.synchronize()
        want = [one.msm_dev(d.data_ptr(), n).tolist() for d in ds]
        want_off = one.msm_dev(ds[-1].data_ptr() + 32 * 4096, n - 8192, off=4096).tolist()
    finally:
        one.close()
    c = hal.Context(urs_n=n, devices=[0] * P)
    try:
        for forced in (False, True):
            if forced:
                hal.dev_hook("force_peer_copy", 1)
            for slot in (0, 1):
                c.msm_dev_batch_begin(slot, [d.data_ptr() for d in ds], n)
            for slot in (0, 1):
                assert c.msm_dev_batch_end(slot, batch).tolist() == want
            c.msm_dev_batch_begin(2, [ds[-1].data_ptr() + 32 * 4096], n - 8192, off=4096)
            with pytest.raises(hal.HaloError):
                c.msm_dev_end(2)                      # a batch is collected by the batch call
            assert c.msm_dev_batch_end(2, 1)[0].tolist() == want_off
    finally:
        c.close()


def test_multi_ctx_peer_copy_path(hal, monkeypatch):
    """Device-resident scalars on a GPU other than a shard's are copied peer-to-peer into the shard's slot buffer in front of
    its launches.  One GPU here, so the copy path is forced (the development library's force_peer_copy hook): hipMemcpyPeerAsync between a
    device and itself is a device-to-device copy; buffers, ordering and offsets are the real ones."""
    import torch
    n = 1 << 18
    d = torch.empty(n * 4, dtype=torch.int64, device="cuda")
    c = hal.Context(urs_n=n, devices=[0, 0, 0, 0])
    try:
        c.rng_scalars_dev(0x9EE9, n, d.data_ptr())
        torch.cuda.synchronize()
        want = c.msm_dev(d.data_ptr(), n)
        want_off = c.msm_dev(d.data_ptr() + 32 * 1000, n - 5000, off=3000)
        hal.dev_hook("force_peer_copy", 1)
        for _ in range(2):
            assert c.msm_dev(d.data_ptr(), n).tolist() == want.tolist()
            assert c.msm_dev(d.data_ptr() + 32 * 1000, n - 5000, off=3000).tolist() == want_off.tolist()
        for slot in range(4):
            c.msm_dev_begin(slot, d.data_ptr(), n)
        assert all(c.msm_dev_end(slot).tolist() == want.tolist() for slot in range(4))
    finally:
        c.close()


def test_multi_ctx_2_24_equals_single_context(hal):
    """BASELINE config 5's size in one process: 8 shards of 2^21 points (each a fixed-base-table MSM in two pieces)."""
    import torch
    n = 1 << 24
    d = torch.empty(n * 4, dtype=torch.int64, device="cuda")
    one = hal.Context(urs_n=n)
    try:
        one.rng_scalars_dev(0x48414C4F00000005, n, d.data_ptr())
        torch.cuda.synchronize()
        want = one.msm_dev(d.data_ptr(), n)
    finally:
        one.close()
    for P in (2, 8):
        c = hal.Context(urs_n=n, devices=[0] * P)
        try:
            assert c.msm_dev(d.data_ptr(), n).tolist() == want.tolist(), P
        finally:
            c.close()


def test_multi_ctx_from_host_bases_and_full_api(hal, urs4096):
    """halo_ctx_create_multi (bases uploaded; blocks go to the shards) and the rest of the API on the same handle: an open +
    check runs on devices[0] as on a plain context"""
    from halo_accumulation_amd import pcdl
    c = hal.Context(urs4096, devices=[0, 0])
    plain = hal.Context(urs4096)
    try:
        sc, s = orc.rng_scalars(77, 4096)
        assert c.msm(sc).tolist() == plain.msm(sc).tolist()
        zw, _ = orc.rng_scalars(s, 2)
        d = 4095
        C = pcdl.commit(c, sc, d, zw[1])
        assert C.tolist() == pcdl.commit(plain, sc, d, zw[1]).tolist()
        pi = pcdl.open(c, [5], sc, C, d, zw[0], zw[1])
        assert pi.tolist() == pcdl.open(plain, [5], sc, C, d, zw[0], zw[1]).tolist()
        pcdl.check_proof(c, C, d, zw[0], c.poly_eval(sc, zw[0]), pi)
    finally:
        c.close()
        plain.close()


def test_multi_ctx_argument_errors(hal):
    with pytest.raises(hal.HaloError):
        hal.Context(urs_n=4096, devices=[0, 99])
    with pytest.raises(hal.HaloError):
        hal.Context(urs_n=4096, devices=[])


@pytest.mark.parametrize("n,P", [(10, 8), (64, 8), (1000, 3), (4097, 5)])
def test_multi_ctx_small_and_ragged_keys(hal, n, P):
    """keys smaller than 4 points per shard (some shards are empty), sizes that are no multiple of the shard count"""
    gs = orc.urs_affine(2, n)
    c = hal.Context(gs, devices=[0] * P)
    try:
        sc, _ = orc.rng_scalars(0xBEE5 + n, n)
        assert c.msm(sc).tolist() == orc.msm_affine(gs, sc).tolist()
        if n > 20:
            assert c.msm(sc[:n - 13], off=5).tolist() == orc.msm_affine(gs[5:n - 8], sc[:n - 13]).tolist()
        assert c.msm(sc[:0]).tolist() == orc.msm_affine(gs[:0], sc[:0]).tolist()   # the empty MSM: infinity
    finally:
        c.close()
