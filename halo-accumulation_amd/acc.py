"""Mirror of the reference's `acc` module (code/src/acc.rs): prover / verifier / decider."""
import ctypes as C

import numpy as np

from halo_accumulation_amd._lib import check, ptr
from halo_accumulation_amd.pcdl import lg_of


def _cat(qs):
    return np.ascontiguousarray(np.concatenate(qs)) if len(qs) else np.zeros(0, dtype=np.uint64)


def instance_from_accumulator(ctx, acc, d):
    """impl From<Accumulator> for Instance (acc.rs:121-131)"""
    return np.ascontiguousarray(acc[: ctx.lib.halo_instance_words(lg_of(d))]).copy()


def random_instance(ctx, rng, d):
    """benches/acc.rs:15-29"""
    st = C.c_uint64(rng[0])
    inst = np.zeros(ctx.lib.halo_instance_words(lg_of(d)), dtype=np.uint64)
    check(ctx.lib.halo_random_instance(ctx.h, C.byref(st), d, ptr(inst)))
    rng[0] = st.value
    return inst


def prover(ctx, rng, d, qs):
    """acc.rs:190-220"""
    st = C.c_uint64(rng[0])
    acc = np.zeros(ctx.lib.halo_accumulator_words(lg_of(d)), dtype=np.uint64)
    check(ctx.lib.halo_acc_prover(ctx.h, C.byref(st), d, ptr(_cat(qs)), len(qs), ptr(acc)))
    rng[0] = st.value
    return acc


def verifier(ctx, d, qs, acc):
    """acc.rs:223-243"""
    check(ctx.lib.halo_acc_verifier(ctx.h, d, ptr(_cat(qs)), len(qs), ptr(np.ascontiguousarray(acc, dtype=np.uint64))))


def decider(ctx, acc):
    """acc.rs:245-255"""
    check(ctx.lib.halo_acc_decider(ctx.h, ptr(np.ascontiguousarray(acc, dtype=np.uint64))))
