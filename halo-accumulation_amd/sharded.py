"""Index-sharded MSM across the GPUs of one node (BASELINE config 5).

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI, "gloo" on CPU for
tests).  An MSM is a sum of independent terms, so rank r owns the index block
[r*n/P, (r+1)*n/P) of bases and scalars, reduces it to ONE point with the local Pippenger
pipeline, and the only exchange step is an all-gather of P x 96 bytes followed by P-1 point
additions in fixed rank order on every rank.  (RCCL has no user-defined reduction operator, so
an elliptic-curve sum cannot be an all-reduce.)  The payload is latency-, not bandwidth-bound:
one collective per MSM and nothing else crosses the fabric.
"""
from __future__ import annotations

import numpy as np


def shard_range(n: int, rank: int, world: int):
    """Block partition of [0, n): sizes differ by at most one."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class ShardedMsm:
    """partial_fn() -> (12,) uint64 Jacobian partial of this rank; sum_fn(points (P,12)) -> (12,)."""

    def __init__(self, partial_fn, sum_fn, device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.partial_fn, self.sum_fn = partial_fn, sum_fn
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.device = device if device is not None else torch.device("cpu")
        self._send = torch.zeros(12, dtype=torch.int64, device=self.device)
        self._recv = torch.zeros(self.world * 12, dtype=torch.int64, device=self.device)

    def gather_batch(self, parts):
        """All-gather k local partials (list of (12,) arrays) with ONE collective and combine each:
        fewer, larger collectives -- the exchange is latency-bound (k * 96 bytes per rank)."""
        k = len(parts)
        if k == 0:
            return []
        local = np.ascontiguousarray(np.stack(parts), dtype=np.uint64)
        if self.world == 1:
            return [local[i] for i in range(k)]
        torch = self.torch
        send = torch.from_numpy(local.view(np.int64).reshape(-1)).to(self.device)
        recv = torch.empty(self.world * k * 12, dtype=torch.int64, device=self.device)
        self.dist.all_gather_into_tensor(recv, send)
        pts = recv.cpu().numpy().view(np.uint64).reshape(self.world, k, 12)
        return [self.sum_fn(np.ascontiguousarray(pts[:, i, :])) for i in range(k)]

    def __call__(self, *args, **kw):
        part = np.ascontiguousarray(self.partial_fn(*args, **kw), dtype=np.uint64)
        if self.world == 1:
            return part
        torch = self.torch
        self._send.copy_(torch.from_numpy(part.view(np.int64)), non_blocking=False)
        self.dist.all_gather_into_tensor(self._recv, self._send)
        pts = self._recv.cpu().numpy().view(np.uint64).reshape(self.world, 12)
        return self.sum_fn(pts)
