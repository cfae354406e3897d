"""Mirror of the reference's `group` module (code/src/group.rs) on top of the C ABI.

Scalars are (.., 4) uint64 Montgomery limbs, affine bases (.., 8), Jacobian points (12,) --
arkworks' in-memory form.  `ctx` is a `_lib.Context` holding the commitment key on the GPU.
"""
from halo_accumulation_amd import _lib


def scalar_dot(ctx, xs, ys):
    """group.rs:13-15"""
    return ctx.scalar_dot(xs, ys)


def point_dot(ctx, xs, Gs):
    """group.rs:18-21: Gs are Jacobian points (m, 12); zips to the shorter input."""
    return ctx.msm_points(Gs, xs)


def point_dot_affine(ctx, xs, Gs=None, off=0):
    """group.rs:24-26: over the resident key GS[off .. off + len(xs)) (Gs=None), or over the caller's own affine
    generators Gs (m, 8), uploaded for the call."""
    if Gs is not None:
        return ctx.msm_affine(Gs, xs)
    return ctx.msm(xs, off=off)


def construct_powers(ctx, z, n):
    """group.rs:29-37"""
    return ctx.powers(z, n)


public_points = _lib.public_points
