"""Mirror of `pedersen::commit` (code/src/pedersen.rs:6-20)."""
import numpy as np

from halo_accumulation_amd import _lib


def commit(ctx, w, n_bases, ms):
    """commit(w, &GS[0..n_bases], ms): AssertionError on a length mismatch, as the reference panics."""
    ms = np.ascontiguousarray(ms, dtype=np.uint64).reshape(-1, 4)
    out = np.zeros(12, dtype=np.uint64)
    w = None if w is None else np.ascontiguousarray(w, dtype=np.uint64)
    _lib.check(ctx.lib.halo_pedersen_commit(ctx.h, _lib.ptr(w), n_bases, _lib.ptr(ms), ms.shape[0], _lib.ptr(out)))
    return out
