"""Mirror of `pedersen::commit` (code/src/pedersen.rs:6-20)."""
import numpy as np

from halo_accumulation_amd import _lib


def commit(ctx, w, Gs, ms):
    """commit(w, Gs, ms): AssertionError on a length mismatch, as the reference panics.

    Gs is either an int n (the prefix GS[0..n) of the resident key, what pcdl.rs:109,338 pass) or an (n, 8) array of
    affine generators of the caller's own (the signature takes any `&[PallasAffine]`)."""
    ms = np.ascontiguousarray(ms, dtype=np.uint64).reshape(-1, 4)
    out = np.zeros(12, dtype=np.uint64)
    w = None if w is None else np.ascontiguousarray(w, dtype=np.uint64)
    if isinstance(Gs, (int, np.integer)):
        _lib.check(ctx.lib.halo_pedersen_commit(ctx.h, _lib.ptr(w), int(Gs), _lib.ptr(ms), ms.shape[0], _lib.ptr(out)))
    else:
        Gs = np.ascontiguousarray(Gs, dtype=np.uint64).reshape(-1, 8)
        _lib.check(ctx.lib.halo_pedersen_commit_affine(ctx.h, _lib.ptr(w), _lib.ptr(Gs), Gs.shape[0], _lib.ptr(ms), ms.shape[0], _lib.ptr(out)))
    return out
