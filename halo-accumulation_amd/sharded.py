"""Sharded MSM across the GPUs of one node (BASELINE config 5).

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI, "gloo" on CPU for
tests).  An MSM is a sum of independent terms, and it can be cut two ways; either way a rank
reduces its share to ONE point with the local Pippenger pipeline, and the only exchange step is an
all-gather of P x 96 bytes followed by P-1 point additions in fixed rank order on every rank.
(RCCL has no user-defined reduction operator, so an elliptic-curve sum cannot be an all-reduce.)
The payload is latency-, not bandwidth-bound: one collective per batch of MSMs and nothing else
crosses the fabric.

* index shards (shard_range): rank r owns the index block [r*n/P, (r+1)*n/P) of bases and scalars.
  Memory and upload traffic are 1/P per rank and the block is a fixed-base-table MSM over the rank's own
  key (from 2^17 points; below 2^20 points in batches of several MSMs per launch).  What bench.py uses
  while a rank's block has at least 2^17 points.
* window shards (window_range, halo_msm_dev_begin_part): every rank keeps the whole key and all scalars
  (160 MiB at n = 2^20 of 288 GiB) and computes the Pippenger windows [r*W/P, (r+1)*W/P) of the full-size
  MSM (general 16-window plan): exactly 1/P of that plan's bucket work.  For blocks too small for a table.
"""
from __future__ import annotations

import ctypes as C

import numpy as np


def shard_range(n: int, rank: int, world: int):
    """Block partition of [0, n): sizes differ by at most one."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def window_range(W: int, rank: int, world: int):
    """The scalar windows [w0, w1) of W that window shard `rank` of `world` computes (the library's own split)."""
    return W * rank // world, W * (rank + 1) // world


class ShardedMsm:
    """partial_fn() -> (12,) uint64 Jacobian partial of this rank; sum_fn(points (P,12)) -> (12,)."""

    def __init__(self, partial_fn, sum_fn, device=None, always_collective=False):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.partial_fn, self.sum_fn = partial_fn, sum_fn
        self.always_collective = always_collective and dist.is_initialized()  # run the all-gather even on one rank (rehearsals)
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.device = device if device is not None else torch.device("cpu")
        self._send = torch.zeros(12, dtype=torch.int64, device=self.device)
        self._recv = torch.zeros(self.world * 12, dtype=torch.int64, device=self.device)
        self._rings = {}  # batch size -> ring of (pinned host, device send, device recv) buffers for gather_start

    def gather_batch(self, parts):
        """All-gather k local partials (list of (12,) arrays) with ONE collective and combine each:
        fewer, larger collectives -- the exchange is latency-bound (k * 96 bytes per rank)."""
        return self.gather_finish(self.gather_start(parts))

    def gather_start(self, parts):
        """Issue the all-gather of k local partials without waiting for it (async_op): the caller enqueues its next launch
        and collects the result with gather_finish -- the collective runs under that launch."""
        k = len(parts)
        if k == 0:
            return (0, None, None, None)
        local = np.ascontiguousarray(np.stack(parts), dtype=np.uint64)
        if self.world == 1 and not self.always_collective:
            return (k, local, None, None)
        # staging buffers are allocated ONCE per batch size (a ring of four sets: the handle of one launch is collected
        # after the next launch has been issued) -- no pinned allocation, no device allocation per launch
        ring = self._rings.setdefault(k, {"next": 0, "sets": []})
        if len(ring["sets"]) < 4:
            torch = self.torch
            pinned = torch.empty(k * 12, dtype=torch.int64)
            if self.device.type != "cpu":
                pinned = pinned.pin_memory()
            ring["sets"].append((pinned,
                                 pinned if self.device.type == "cpu" else torch.empty(k * 12, dtype=torch.int64, device=self.device),
                                 torch.empty(self.world * k * 12, dtype=torch.int64, device=self.device)))
            pinned, send, recv = ring["sets"][-1]
        else:
            pinned, send, recv = ring["sets"][ring["next"] % 4]
        ring["next"] += 1
        pinned.numpy()[:] = local.view(np.int64).reshape(-1)
        if send is not pinned:
            send.copy_(pinned, non_blocking=True)
        work = self.dist.all_gather_into_tensor(recv, send, async_op=True)
        return (k, send, recv, work)

    def gather_finish(self, handle):
        k, send, recv, work = handle
        if k == 0:
            return []
        if work is None:
            return [send[i] for i in range(k)]
        work.wait()
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


def make_allgather(device=None):
    """allgather(arr: uint64 ndarray) -> (P, len(arr)) uint64 over torch.distributed (RCCL when `device` is the rank's GPU,
    gloo with device=None/cpu): what ShardedOpen takes.  The staging buffers of a message size are allocated once."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    dev = device if device is not None else torch.device("cpu")
    bufs = {}

    def allgather(arr):
        a = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1)
        k = a.size
        if k not in bufs:
            host = torch.empty(k, dtype=torch.int64)
            if dev.type != "cpu":
                host = host.pin_memory()
            bufs[k] = (host, host if dev.type == "cpu" else torch.empty(k, dtype=torch.int64, device=dev),
                       torch.empty(world * k, dtype=torch.int64, device=dev))
        host, send, recv = bufs[k]
        host.numpy()[:] = a.view(np.int64)
        if send is not host:
            send.copy_(host, non_blocking=True)
        dist.all_gather_into_tensor(recv, send)
        return recv.cpu().numpy().view(np.uint64).reshape(world, k)

    return allgather


# Fq Montgomery one: the X, Y of the library's normalised point at infinity (1, 1, 0)
_FQ_ONE = np.array([0x34786D38FFFFFFFD, 0x992C350BE41914AD, 0xFFFFFFFFFFFFFFFF, 0x3FFFFFFFFFFFFFFF], dtype=np.uint64)


class ShardedOpen:
    """pcdl::open (non-hiding branch, pcdl.rs:120-242) with G, c and the z-powers sharded cyclically:
    element i lives on rank i mod P.  The fold pairs j with j + m (m = n/2, n/4, ...), so both partners
    stay on one rank for every round with m >= P: a round costs one all-gather of P x 256 bytes
    (partial L, R and the two partial dot products) and NO vector exchange.  After lg(n/P) rounds each
    rank holds one element of G, c, z; they are gathered (P x 160 bytes) and the last lg P rounds run
    replicated on every rank.  Every rank derives the same challenges and returns the same proof, which is
    bit-identical to the single-GPU halo_pcdl_open.

    allgather(arr) -> (P, len(arr)) uint64 is supplied by the caller (torch.distributed over RCCL or gloo).
    """

    def __init__(self, lib, rank: int, world: int, allgather, device: int = 0, always_collective: bool = False):
        assert world & (world - 1) == 0, "world size must be a power of two"
        self.lib, self.rank, self.world, self.allgather, self.device = lib, rank, world, allgather, device
        self.coll = world > 1 or (always_collective and allgather is not None)  # one rank: run the collectives anyway (rehearsals)
        self.ctx = None

    def load_key(self, n: int, first_index: int = 2):
        """This rank's cyclic shard of the key: G_{r + jP} = hash(first_index + r + jP)  (main.rs:35-45)."""
        assert n % self.world == 0
        self.n = n
        self.ctx = self.lib.Context(urs_n=n // self.world, first_index=first_index + self.rank, stride=self.world, device=self.device)
        return self.ctx

    def _callback(self):
        """-> (object to keep alive, function pointer, user pointer) for halo_pcdl_open_sharded / _check_sharded.  An all-gather
        that brings its own C entry point (rccl.RcclGather: .fn = halo_allgather_rccl, .user = its handle) is passed as it is --
        no Python frame in the collective path; anything else is wrapped."""
        if self.allgather is None:
            return None, None, None
        if hasattr(self.allgather, "fn") and hasattr(self.allgather, "user"):
            return None, self.allgather.fn, self.allgather.user
        cb = self.lib.make_allgather_callback(self.allgather, self.world)
        return cb, C.cast(cb, C.c_void_p), None

    def open(self, coeffs_local, Cm, z, w=None, rng=None, deg=None):
        """pcdl::open over the sharded key in ONE library call (halo_pcdl_open_sharded): the round loop runs in the library,
        every collective is a call back into self.allgather.  coeffs_local: this rank's coefficients c[r::P]; hiding
        (pcdl.rs:137-164): the commitment randomness w, rng = [SplitMix64 state] (mutated, the same on every rank) and
        deg = p.degree().  -> (proof, v); open_by_rounds is the same protocol spelled out call by call."""
        from ._lib import check, ptr
        if not self.coll:  # one rank, no forced collectives
            cb, cbp, user = None, None, None
        else:
            cb, cbp, user = self._callback()
        co = np.ascontiguousarray(coeffs_local, dtype=np.uint64).reshape(-1, 4)
        lg_n = self.n.bit_length() - 1
        proof = np.zeros(self.lib.load().halo_proof_words(lg_n), dtype=np.uint64)
        v = np.zeros(4, dtype=np.uint64)
        st = C.c_uint64(rng[0] if rng is not None else 0)
        rc = self.lib.load().halo_pcdl_open_sharded(self.ctx.h, self.world, self.rank, C.byref(st), ptr(co), co.shape[0], int(deg or 0),
                                                    ptr(np.ascontiguousarray(Cm, dtype=np.uint64)), self.n - 1, ptr(np.ascontiguousarray(z, dtype=np.uint64)),
                                                    ptr(np.ascontiguousarray(w, dtype=np.uint64)) if w is not None else None, cbp, user, ptr(proof), ptr(v))
        if cb is not None and cb.error is not None:
            raise cb.error
        check(rc)
        if rng is not None and w is not None:
            rng[0] = st.value
        return proof, v

    def check(self, Cm, d, z, v, proof):
        """pcdl::check (pcdl.rs:323-342) against the sharded key in one library call (halo_pcdl_check_sharded): every rank
        runs the succinct check (host arithmetic) and commits to its own share of h's coefficients over its own points;
        one all-gather of 96 bytes per rank, the shares added in rank order, U compared (:339).  Raises HaloReject like
        pcdl.check_proof, on every rank alike."""
        from ._lib import check, ptr
        cb, cbp, user = self._callback() if self.coll else (None, None, None)
        a = lambda x: ptr(np.ascontiguousarray(x, dtype=np.uint64))
        rc = self.lib.load().halo_pcdl_check_sharded(self.ctx.h, self.world, self.rank, a(Cm), d, a(z), a(v), a(proof), cbp, user)
        if cb is not None and cb.error is not None:
            raise cb.error
        check(rc)

    def _gather(self, words, local):
        """One collective of the by-rounds protocol: local() -> this rank's record (`words` uint64).  An exception on this
        rank does not leave the peers waiting: the rank still enters the all-gather, with a status word of 1 behind a zeroed
        record (the same rule as halo_pcdl_open_sharded), and EVERY rank raises after it.  -> (P, words)"""
        rec, err = np.zeros(words + 1, dtype=np.uint64), None
        try:
            if self._pending is not None:
                raise self._pending
            rec[:words] = local()
        except Exception as e:  # noqa: BLE001 -- reported after the collective
            err = e
            rec[:] = 0
            rec[words] = 1
        self._pending = None
        got = self.allgather(rec) if self.coll else rec[None]
        bad = [r for r in range(got.shape[0]) if got[r, words] != 0]
        if err is not None:
            raise err
        if bad:
            raise self.lib.HaloError("sharded open: rank %d failed locally; every rank stops at the same collective" % bad[0])
        return np.ascontiguousarray(got[:, :words])

    def _defer(self, fn):
        """A local step between two collectives (a fold, p' = p + alpha p_bar): its failure rides into the next collective."""
        try:
            if self._pending is None:
                fn()
        except Exception as e:  # noqa: BLE001
            self._pending = e

    def _rounds(self, ipa, count, Hp, xi, Ls, Rs, world):
        for _ in range(count):
            parts = self._gather(32, ipa.round_lr_partial)
            L, R, xi, xi_inv = self.lib.open_combine(parts, Hp, xi)
            Ls.append(L)
            Rs.append(R)
            self._defer(lambda: ipa.round_fold(xi, xi_inv))
        return xi

    def open_by_rounds(self, coeffs_local, Cm, z, w=None, rng=None, deg=None):
        """The protocol of `open` spelled out call by call from Python (what halo_pcdl_open_sharded does inside, status
        word included).  coeffs_local: this rank's coefficients c[r::P] (zero-padded to n/P by the library).
        Hiding (pcdl.rs:137-164): pass the commitment randomness w, rng = [SplitMix64 state] (mutated, the
        same on every rank) and deg = p.degree().  -> (proof, v)"""
        P, nl = self.world, self.n // self.world
        lg_n = self.n.bit_length() - 1
        self._pending = None
        box = {}

        def begin():
            box["ipa"] = self.lib.Ipa(self.ctx, nl, coeffs_local, z, stride=P, offset=self.rank)
            return box["ipa"].dot_cz()  # this shard's share of p(z)

        hiding = w is not None
        Cbar = wp = None
        try:
            if hiding:
                rec = self._gather(16, lambda: np.concatenate([begin(), box["ipa"].hiding_partial(rng[0], deg, z, P, self.rank)]))
                ipa = box["ipa"]
                v_parts = np.ascontiguousarray(rec[:, :4])
                Cbar, alpha, wp, C_prime, rng[0] = self.lib.open_hiding_combine(Cm, z, v_parts, np.ascontiguousarray(rec[:, 4:]), w, rng[0], deg)
                self._defer(lambda: ipa.apply_hiding(alpha))  # p' = p + alpha p_bar
                Cm = C_prime
            else:
                v_parts = self._gather(4, begin)
                ipa = box["ipa"]
            v, xi, Hp = self.lib.open_start(Cm, z, v_parts)
            Ls, Rs = [], []
            xi = self._rounds(ipa, nl.bit_length() - 1, Hp, xi, Ls, Rs, P)
            if P > 1:
                rec = self._gather(20, lambda: np.concatenate(ipa.finish_z()))  # (P, 20): the P remaining elements in index order
                tL, tR, U, c = self.lib.open_tail(rec, Hp, xi)  # the last lg P rounds: host arithmetic, the same on every rank
                Ls.extend(tL)
                Rs.extend(tR)
            else:
                if self._pending is not None:
                    raise self._pending
                U, c, z0 = ipa.finish_z()
        finally:
            if box.get("ipa") is not None:
                box["ipa"].close()
        proof = np.zeros(self.lib.load().halo_proof_words(lg_n), dtype=np.uint64)
        proof[1] = lg_n
        for i in range(lg_n):
            proof[2 + 12 * i: 14 + 12 * i] = Ls[i]
            proof[2 + 12 * lg_n + 12 * i: 14 + 12 * lg_n + 12 * i] = Rs[i]
        o = 2 + 24 * lg_n
        proof[o: o + 12] = U
        proof[o + 12: o + 16] = c
        if hiding:
            proof[0] = 1
            proof[o + 16: o + 28] = Cbar
            proof[o + 28: o + 32] = wp
        else:
            proof[o + 16: o + 20] = _FQ_ONE  # C_bar = None: the point at infinity
            proof[o + 20: o + 24] = _FQ_ONE
        return proof, v
