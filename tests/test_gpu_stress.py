"""Soak test: many interleaved launches on one context -- pipelined single MSMs, batched launches, window shards and whole
pcdl::open calls sharing the slots -- every result compared with a value computed once by the synchronous path.
Order-dependent state (workspace growth, launch-graph capture and replay, tuning knobs) is what this is after."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_interleaved_launches_soak():
    import torch
    import halo_accumulation_amd as h
    from halo_accumulation_amd import pcdl
    L = h._lib
    rnd = np.random.RandomState(7)
    N = 1 << 14
    ctx = L.Context(urs_n=N)
    other = L.Context(urs_n=4096)  # a second context allocating and freeing next to the first
    try:
        sizes = [1 << 14, 1 << 13, 12288, 1000]
        sets = {}
        for n in sizes:
            ds = []
            for i in range(4):
                d = torch.empty(n * 4, dtype=torch.int64, device="cuda")
                ctx.rng_scalars_dev(1234 + 17 * n + i, n, d.data_ptr())
                ds.append(d)
            sets[n] = (ds, [ctx.msm_dev(d.data_ptr(), n).tolist() for d in ds])
        # one reference proof
        co = torch.empty((N + 2) * 4, dtype=torch.int64, device="cuda")
        ctx.rng_scalars_dev(99, N + 2, co.data_ptr())
        cz = np.ascontiguousarray(co.cpu().numpy().view(np.uint64).reshape(N + 2, 4))
        coeffs, z, w = np.ascontiguousarray(cz[:N]), cz[N], cz[N + 1]
        C = pcdl.commit(ctx, coeffs, N - 1, w)
        proof = pcdl.open(ctx, [5], coeffs, C, N - 1, z, w).tolist()
        od = torch.zeros(4096 * 4, dtype=torch.int64, device="cuda")
        in_flight = {}  # slot -> (kind, n, payload)
        for step in range(2500):
            slot = int(rnd.randint(4))
            if slot in in_flight:
                kind, n, payload = in_flight.pop(slot)
                ds, want = sets[n]
                if kind == "one":
                    assert ctx.msm_dev_end(slot).tolist() == want[payload]
                elif kind == "batch":
                    got = ctx.msm_dev_batch_end(slot, len(payload))
                    assert got.tolist() == [want[i] for i in payload]
                else:  # window shards: finish this part, run the others synchronously, sum
                    parts, i = payload
                    partials = [ctx.msm_dev_end(slot)]
                    for part in range(1, parts):
                        ctx.msm_dev_begin(slot, ds[i].data_ptr(), n, part=part, parts=parts)
                        partials.append(ctx.msm_dev_end(slot))
                    assert L.point_sum(np.stack(partials)).tolist() == want[i]
                continue
            n = sizes[rnd.randint(len(sizes))]
            ds, want = sets[n]
            r = rnd.rand()
            if r < 0.5:
                i = int(rnd.randint(4))
                ctx.msm_dev_begin(slot, ds[i].data_ptr(), n)
                in_flight[slot] = ("one", n, i)
            elif r < 0.75:
                idx = [int(v) for v in rnd.permutation(4)[: rnd.randint(1, 5)]]
                ctx.msm_dev_batch_begin(slot, [ds[i].data_ptr() for i in idx], n)
                in_flight[slot] = ("batch", n, idx)
            elif r < 0.9:
                parts, i = int(rnd.randint(2, 6)), int(rnd.randint(4))
                ctx.msm_dev_begin(slot, ds[i].data_ptr(), n, part=0, parts=parts)
                in_flight[slot] = ("parts", n, (parts, i))
            elif r < 0.95 and not any(s in in_flight for s in (0, 1)):
                # a whole open uses slots 0 and 1 (and stream 2) itself
                assert pcdl.open(ctx, [5], coeffs, C, N - 1, z, w).tolist() == proof
            else:
                # the other context grows and shrinks a workspace next to us
                k = int(rnd.randint(2, 9))
                other.msm_dev_batch_begin(0, [od.data_ptr()] * k, 4096)
                other.msm_dev_batch_end(0, k)
        for slot, (kind, n, payload) in list(in_flight.items()):
            if kind == "batch":
                ctx.msm_dev_batch_end(slot, len(payload))
            else:
                ctx.msm_dev_end(slot)
    finally:
        ctx.close()
        other.close()
