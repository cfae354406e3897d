"""Mirror of the reference's `pcdl` module (code/src/pcdl.rs): commit / open / succinct_check / check.

Polynomials are coefficient arrays (len, 4); proofs are the flat EvalProof blobs described in
include/halo_accumulation.h.  `rng` is a one-element list holding a SplitMix64 state (mutated),
standing in for the reference's `rng: &mut R`.
"""
import ctypes as C

import numpy as np

from halo_accumulation_amd import _lib
from halo_accumulation_amd._lib import check, ptr


def _a(x):
    return None if x is None else np.ascontiguousarray(x, dtype=np.uint64)


def lg_of(d):
    return (d + 1).bit_length() - 1


def commit(ctx, p, d, w=None):
    """pcdl.rs:99-110"""
    p = _a(p).reshape(-1, 4)
    out = np.zeros(12, dtype=np.uint64)
    check(ctx.lib.halo_pcdl_commit(ctx.h, ptr(p), p.shape[0], d, ptr(_a(w)), ptr(out)))
    return out


def open(ctx, rng, p, Cm, d, z, w=None):
    """pcdl.rs:120-242 -> EvalProof blob"""
    p = _a(p).reshape(-1, 4)
    st = C.c_uint64(rng[0])
    proof = np.zeros(ctx.lib.halo_proof_words(max(lg_of(d), 0)), dtype=np.uint64)
    check(ctx.lib.halo_pcdl_open(ctx.h, C.byref(st), ptr(p), p.shape[0], ptr(_a(Cm)), d, ptr(_a(z)), ptr(_a(w)), ptr(proof)))
    rng[0] = st.value
    return proof


def commit_dev(ctx, dptr, length, d, w=None):
    """pcdl::commit for `length` coefficients resident in device memory"""
    out = np.zeros(12, dtype=np.uint64)
    check(ctx.lib.halo_pcdl_commit_dev(ctx.h, C.c_void_p(dptr), length, d, ptr(_a(w)), ptr(out)))
    return out


def open_dev(ctx, rng, dptr, length, Cm, d, z, w=None):
    """pcdl::open for a polynomial resident in device memory (length = degree + 1) -> EvalProof blob"""
    st = C.c_uint64(rng[0])
    proof = np.zeros(ctx.lib.halo_proof_words(max(lg_of(d), 0)), dtype=np.uint64)
    check(ctx.lib.halo_pcdl_open_dev(ctx.h, C.byref(st), C.c_void_p(dptr), length, ptr(_a(Cm)), d, ptr(_a(z)), ptr(_a(w)), ptr(proof)))
    rng[0] = st.value
    return proof


def succinct_check(ctx, Cm, d, z, v, pi):
    """pcdl.rs:252-314 -> (xis of h, U); raises HaloReject where the reference returns Err"""
    lg = max(lg_of(d), 0)
    xis = np.zeros((lg + 1, 4), dtype=np.uint64)
    U = np.zeros(12, dtype=np.uint64)
    check(ctx.lib.halo_pcdl_succinct_check(ctx.h, ptr(_a(Cm)), d, ptr(_a(z)), ptr(_a(v)), ptr(_a(pi)), ptr(xis), ptr(U)))
    return xis, U


def succinct_check_batch(ctx, d, instances):
    """m succinct checks at once -> (xis (m, lg+1, 4), Us (m, 12), status (m,)); raises HaloReject if any instance fails
    (status then still tells which: use return_status=True semantics through the status array of the exception args)"""
    qs = np.ascontiguousarray(np.concatenate(instances)) if len(instances) else np.zeros(0, dtype=np.uint64)
    m, lg = len(instances), max(lg_of(d), 0)
    xis = np.zeros((m, lg + 1, 4), dtype=np.uint64)
    Us = np.zeros((m, 12), dtype=np.uint64)
    status = (C.c_int * max(m, 1))()
    rc = ctx.lib.halo_pcdl_succinct_check_batch(ctx.h, d, ptr(qs), m, ptr(xis), ptr(Us), status)
    st = [status[i] for i in range(m)]
    if rc == _lib.HALO_E_REJECT:
        raise _lib.HaloReject(ctx.lib.halo_last_error().decode(), st)
    check(rc)
    return xis, Us, st


def check_proof(ctx, Cm, d, z, v, pi):
    """pcdl::check (pcdl.rs:323-342)"""
    check(ctx.lib.halo_pcdl_check(ctx.h, ptr(_a(Cm)), d, ptr(_a(z)), ptr(_a(v)), ptr(_a(pi))))


def check_partial(ctx, Cm, d, z, v, pi, stride, offset):
    """One rank's half of pcdl::check over a cyclically sharded key (halo_pcdl_check_partial): -> (U, this rank's share of
    CM.Commit(ck, h)).  The caller adds the shares in rank order and accepts iff the sum is U (sharded.ShardedOpen.check)."""
    U, part = np.zeros(12, dtype=np.uint64), np.zeros(12, dtype=np.uint64)
    check(ctx.lib.halo_pcdl_check_partial(ctx.h, ptr(_a(Cm)), d, ptr(_a(z)), ptr(_a(v)), ptr(_a(pi)), stride, offset, ptr(U), ptr(part)))
    return U, part


class HPoly:
    """pcdl.rs:44-92"""

    def __init__(self, ctx, xis):
        self.ctx, self.xis = ctx, _a(xis).reshape(-1, 4)

    def get_poly(self):
        return self.ctx.h_coeffs(self.xis)

    def eval(self, z):
        return self.ctx.h_eval_batch(self.xis[None], z)[0]
