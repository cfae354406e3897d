"""ctypes binding of the optional libhalo_rccl.so (include/halo_rccl.h): halo_allgather_fn over RCCL, native.

The sharded entry points (halo_pcdl_open_sharded / _check_sharded) take the caller's all-gather as a C function pointer.  With
this library the pointer is halo_allgather_rccl itself: no Python in the collective path, and a Rust host links the same symbol."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib

LIB_PATH = os.path.join(_lib.PKG, "libhalo_rccl.so")
ID_BYTES = 128
_lib_rccl = None


def available() -> bool:
    return os.path.exists(LIB_PATH)


def load():
    global _lib_rccl
    if _lib_rccl is None:
        if not available():
            raise _lib.HaloError("libhalo_rccl.so is missing (built by `make` when the image has librccl)")
        try:
            import torch  # noqa: F401 -- one HIP runtime (and one librccl) per process: torch's copies are loaded first
        except ImportError:
            pass
        lib = C.CDLL(LIB_PATH)
        lib.halo_rccl_last_error.restype = C.c_char_p
        lib.halo_rccl_unique_id.argtypes = [C.c_char_p]
        lib.halo_rccl_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        lib.halo_rccl_wrap.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        lib.halo_rccl_destroy.argtypes = [C.c_void_p]
        lib.halo_rccl_destroy.restype = None
        lib.halo_allgather_rccl.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_size_t, C.POINTER(C.c_uint64)]
        lib.halo_rccl_calls.argtypes = [C.c_void_p]
        lib.halo_rccl_calls.restype = C.c_size_t
        lib.halo_rccl_world.argtypes = [C.c_void_p]
        _lib_rccl = lib
    return _lib_rccl


def _check(rc: int) -> None:
    if rc != 0:
        raise _lib.HaloError("libhalo_rccl: " + load().halo_rccl_last_error().decode())


def unique_id() -> bytes:
    buf = C.create_string_buffer(ID_BYTES)
    _check(load().halo_rccl_unique_id(buf))
    return buf.raw


class RcclGather:
    """One rank's communicator (halo_rccl_create).  .fn / .user are what halo_pcdl_open_sharded takes as (allgather, user);
    calling the object gathers a numpy record like the Python callbacks of sharded.py do."""

    def __init__(self, uid: bytes, rank: int, world: int, device: int = 0):
        assert len(uid) == ID_BYTES
        h = C.c_void_p()
        _check(load().halo_rccl_create(uid, rank, world, device, C.byref(h)))
        self.h, self.rank, self.world = h, rank, world
        self.fn = C.cast(load().halo_allgather_rccl, C.c_void_p)
        self.user = h

    def __call__(self, arr):
        a = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1)
        out = np.zeros(self.world * a.size, dtype=np.uint64)
        _check(load().halo_allgather_rccl(self.h, a.ctypes.data_as(C.POINTER(C.c_uint64)), a.size, out.ctypes.data_as(C.POINTER(C.c_uint64))))
        return out.reshape(self.world, a.size)

    @property
    def calls(self) -> int:
        return load().halo_rccl_calls(self.h)

    def close(self):
        if getattr(self, "h", None):
            load().halo_rccl_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
