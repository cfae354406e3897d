"""MI355X-native MSM / IPA hot path of rasmus-kirk/halo-accumulation.

Python here is plumbing only: a ctypes binding of the C ABI in include/halo_accumulation.h
(`_lib`), thin mirrors of the reference's `group` / `pedersen` / `pcdl` / `acc` modules that
call it, and the torch.distributed driver for the sharded MSM.  All arithmetic runs in the
HIP kernels of csrc/ (libhalo_hip.so); importing works without a GPU, computing does not.
"""
from halo_accumulation_amd import _lib  # noqa: F401
from halo_accumulation_amd._lib import HaloError, build, load  # noqa: F401

__all__ = ["_lib", "HaloError", "build", "load"]
