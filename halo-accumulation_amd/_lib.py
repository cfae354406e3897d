"""ctypes binding of libhalo_hip.so (include/halo_accumulation.h) and its build recipe."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB_PATH = os.path.join(PKG, "libhalo_hip.so")
DEV_LIB_PATH = os.path.join(PKG, "libhalo_hip_dev.so")

HALO_OK, HALO_E_ASSERT, HALO_E_REJECT, HALO_E_ARG, HALO_E_DEVICE = 0, -1, -2, -3, -4

u64p = C.POINTER(C.c_uint64)


class HaloError(RuntimeError):
    """Raised for HALO_E_ARG / HALO_E_DEVICE (library-level failures)."""


class HaloReject(ValueError):
    """The reference's `ensure!` -> Err (verifier-side rejection)."""


def build(force: bool = False, jobs: int = 4) -> str:
    """Compile every HIP translation unit for gfx950 into libhalo_hip.so (in-tree)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp", ".cpp"))]
    srcs.append(os.path.join(PKG, "..", "include", "halo_accumulation.h"))
    stale = force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", CSRC, "-j", str(jobs), "-s"])
    return LIB_PATH


_lib = None

_SIGS = {
    "halo_last_error": (C.c_char_p, []),
    "halo_device_count": (C.c_int, []),
    "halo_ctx_create": (C.c_int, [C.c_int, u64p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "halo_ctx_create_urs": (C.c_int, [C.c_int, C.c_uint64, C.c_size_t, C.POINTER(C.c_void_p)]),
    "halo_ctx_create_urs_strided": (C.c_int, [C.c_int, C.c_uint64, C.c_uint64, C.c_size_t, C.POINTER(C.c_void_p)]),
    "halo_ctx_create_multi": (C.c_int, [C.POINTER(C.c_int), C.c_int, u64p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "halo_ctx_create_urs_multi": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.c_uint64, C.c_size_t, C.POINTER(C.c_void_p)]),
    "halo_ctx_devices": (C.c_int, [C.c_void_p]),
    "halo_ctx_clone": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "halo_ctx_destroy": (None, [C.c_void_p]),
    "halo_ctx_size": (C.c_size_t, [C.c_void_p]),
    "halo_ctx_read_bases": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, u64p]),
    "halo_ctx_bases_dev": (C.c_void_p, [C.c_void_p]),
    "halo_ctx_stream": (C.c_void_p, [C.c_void_p]),
    "halo_public_points": (C.c_int, [u64p, u64p]),
    "halo_msm": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, u64p, C.c_int, u64p]),
    "halo_msm_begin": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, u64p, C.c_int]),
    "halo_msm_end": (C.c_int, [C.c_void_p, C.c_int, u64p]),
    "halo_msm_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_int, u64p]),
    "halo_msm_dev_begin": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_int]),
    "halo_msm_dev_end": (C.c_int, [C.c_void_p, C.c_int, u64p]),
    "halo_msm_points": (C.c_int, [C.c_void_p, u64p, u64p, C.c_size_t, u64p]),
    "halo_msm_affine": (C.c_int, [C.c_void_p, u64p, u64p, C.c_size_t, C.c_int, u64p]),
    "halo_scalar_dot": (C.c_int, [C.c_void_p, u64p, u64p, C.c_size_t, u64p]),
    "halo_powers": (C.c_int, [C.c_void_p, u64p, C.c_size_t, u64p]),
    "halo_poly_eval": (C.c_int, [C.c_void_p, u64p, C.c_size_t, u64p, u64p]),
    "halo_h_coeffs": (C.c_int, [C.c_void_p, u64p, C.c_size_t, u64p]),
    "halo_h_commit": (C.c_int, [C.c_void_p, u64p, C.c_size_t, u64p]),
    "halo_h_eval_batch": (C.c_int, [C.c_void_p, u64p, C.c_size_t, C.c_size_t, u64p, u64p]),
    "halo_h_accumulate": (C.c_int, [C.c_void_p, u64p, u64p, u64p, C.c_size_t, C.c_size_t, u64p]),
    "halo_ipa_begin": (C.c_int, [C.c_void_p, C.c_size_t, u64p, C.c_size_t, u64p, C.POINTER(C.c_void_p)]),
    "halo_ipa_begin_strided": (C.c_int, [C.c_void_p, C.c_size_t, u64p, C.c_size_t, u64p, C.c_uint64, C.c_uint64, C.POINTER(C.c_void_p)]),
    "halo_ipa_begin_vectors": (C.c_int, [C.c_void_p, C.c_size_t, u64p, u64p, C.POINTER(C.c_void_p)]),
    "halo_ipa_dot_cz": (C.c_int, [C.c_void_p, u64p]),
    "halo_ipa_round_lr_partial": (C.c_int, [C.c_void_p, u64p, u64p, u64p]),
    "halo_ipa_finish_z": (C.c_int, [C.c_void_p, u64p, u64p, u64p]),
    "halo_ipa_hiding_partial": (C.c_int, [C.c_void_p, C.c_uint64, C.c_size_t, u64p, C.c_uint64, C.c_uint64, u64p]),
    "halo_ipa_apply_hiding": (C.c_int, [C.c_void_p, u64p]),
    "halo_open_hiding_combine": (C.c_int, [u64p, u64p, u64p, u64p, C.c_size_t, u64p, C.POINTER(C.c_uint64), C.c_size_t, u64p, u64p, u64p, u64p]),
    "halo_open_start": (C.c_int, [u64p, u64p, u64p, C.c_size_t, u64p, u64p, u64p]),
    "halo_open_combine": (C.c_int, [u64p, C.c_size_t, u64p, u64p, u64p, u64p, u64p, u64p]),
    "halo_open_tail": (C.c_int, [u64p, C.c_size_t, u64p, u64p, u64p, u64p, u64p, u64p]),
    "halo_ipa_round_lr": (C.c_int, [C.c_void_p, u64p, u64p, u64p]),
    "halo_ipa_round_fold": (C.c_int, [C.c_void_p, u64p, u64p]),
    "halo_ipa_finish": (C.c_int, [C.c_void_p, u64p, u64p]),
    "halo_ipa_destroy": (None, [C.c_void_p]),
    "halo_ipa_len": (C.c_size_t, [C.c_void_p]),
    "halo_proof_words": (C.c_size_t, [C.c_size_t]),
    "halo_instance_words": (C.c_size_t, [C.c_size_t]),
    "halo_accumulator_words": (C.c_size_t, [C.c_size_t]),
    "halo_pedersen_commit": (C.c_int, [C.c_void_p, u64p, C.c_size_t, u64p, C.c_size_t, u64p]),
    "halo_pedersen_commit_affine": (C.c_int, [C.c_void_p, u64p, u64p, C.c_size_t, u64p, C.c_size_t, u64p]),
    "halo_pcdl_commit": (C.c_int, [C.c_void_p, u64p, C.c_size_t, C.c_size_t, u64p, u64p]),
    "halo_pcdl_open": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), u64p, C.c_size_t, u64p, C.c_size_t, u64p, u64p, u64p]),
    "halo_pcdl_open_dev": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p, C.c_size_t, u64p, C.c_size_t, u64p, u64p, u64p]),
    "halo_pcdl_commit_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, u64p, u64p]),
    "halo_pcdl_succinct_check": (C.c_int, [C.c_void_p, u64p, C.c_size_t, u64p, u64p, u64p, u64p, u64p]),
    "halo_pcdl_succinct_check_batch": (C.c_int, [C.c_void_p, C.c_size_t, u64p, C.c_size_t, u64p, u64p, C.POINTER(C.c_int)]),
    "halo_pcdl_check": (C.c_int, [C.c_void_p, u64p, C.c_size_t, u64p, u64p, u64p]),
    "halo_pcdl_check_partial": (C.c_int, [C.c_void_p, u64p, C.c_size_t, u64p, u64p, u64p, C.c_uint64, C.c_uint64, u64p, u64p]),
    "halo_pcdl_open_sharded": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64), u64p, C.c_size_t, C.c_size_t, u64p, C.c_size_t,
                                         u64p, u64p, C.c_void_p, C.c_void_p, u64p, u64p]),
    "halo_pcdl_check_sharded": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, u64p, C.c_size_t, u64p, u64p, u64p, C.c_void_p, C.c_void_p]),
    "halo_acc_prover": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.c_size_t, u64p, C.c_size_t, u64p]),
    "halo_acc_verifier": (C.c_int, [C.c_void_p, C.c_size_t, u64p, C.c_size_t, u64p]),
    "halo_acc_decider": (C.c_int, [C.c_void_p, u64p]),
    "halo_random_instance": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.c_size_t, u64p]),
    "halo_proof_encoded_size": (C.c_size_t, [C.c_size_t, C.c_int]),
    "halo_instance_encoded_size": (C.c_size_t, [C.c_size_t, C.c_int]),
    "halo_accumulator_encoded_size": (C.c_size_t, [C.c_size_t]),
    "halo_proof_encode": (C.c_int, [u64p, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "halo_proof_decode": (C.c_int, [C.c_char_p, C.c_size_t, u64p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "halo_instance_encode": (C.c_int, [u64p, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "halo_instance_decode": (C.c_int, [C.c_char_p, C.c_size_t, u64p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "halo_accumulator_encode": (C.c_int, [u64p, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "halo_accumulator_decode": (C.c_int, [C.c_char_p, C.c_size_t, u64p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "halo_prof_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "halo_prof_reset": (C.c_int, [C.c_void_p]),
    "halo_prof_count": (C.c_int, [C.c_void_p]),
    "halo_prof_get": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(C.c_long)]),
    "halo_msm_dev_begin_part": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "halo_msm_dev_batch_begin": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.POINTER(C.c_void_p), C.c_size_t, C.c_int, C.c_int, C.c_int]),
    "halo_msm_dev_batch_end": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, u64p]),
    "halo_set_table_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "halo_set_fold_table": (C.c_int, [C.c_void_p, C.c_int]),
    "halo_ctx_info": (C.c_size_t, [C.c_void_p, C.c_int]),
    "halo_set_memory_budget": (C.c_int, [C.c_void_p, C.c_size_t]),
    "halo_point_sum": (C.c_int, [u64p, C.c_size_t, u64p]),
    "halo_rng_scalars_dev": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.c_size_t, C.c_void_p]),
}

# libhalo_hip_dev.so (include/halo_accumulation_dev.h): experiment knobs, primitive test hooks, fault injectors.  Loaded only
# when one of these is called -- tests and tools/; the product path (pcdl.py, acc.py, sharded.py, bench.py's timed legs) never does.
_DEV_SIGS = {
    "halo_dev_hook": (C.c_int, [C.c_char_p, C.c_long]),
    "halo_dev_tuning": (C.c_long, [C.c_char_p]),
    "halo_set_batch_verify": (C.c_int, [C.c_void_p, C.c_int]),
    "halo_bench_fr_kernel": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_int]),
    "halo_set_window_bits": (C.c_int, [C.c_void_p, C.c_int]),
    "halo_set_reduce_span": (C.c_int, [C.c_void_p, C.c_int]),
    "halo_set_task_len": (C.c_int, [C.c_void_p, C.c_int]),
    "halo_set_sort_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "halo_set_small_path": (C.c_int, [C.c_void_p, C.c_int]),
    "halo_set_fold_levels": (C.c_int, [C.c_void_p, C.c_int]),
    "halo_set_fold_async": (C.c_int, [C.c_void_p, C.c_int]),
    "halo_set_ipa_switch": (C.c_int, [C.c_void_p, C.c_size_t]),
    "halo_set_graphs": (C.c_int, [C.c_void_p, C.c_int]),
    "halo_test_glv_digits": (C.c_int, [u64p, C.POINTER(C.c_uint8), C.POINTER(C.c_int)]),
    "halo_test_fold_digits": (C.c_int, [u64p, C.POINTER(C.c_int8)]),
    "halo_test_field_op": (C.c_int, [C.c_void_p, C.c_int, C.c_int, u64p, u64p, C.c_size_t, u64p]),
    "halo_test_point_op": (C.c_int, [C.c_void_p, C.c_int, u64p, u64p, C.c_size_t, u64p]),
}


def declared_symbols():
    return sorted(_SIGS)


def declared_dev_symbols():
    return sorted(_DEV_SIGS)


def load():
    """dlopen libhalo_hip.so; fails loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HaloError("libhalo_hip.so is missing: run __graft_entry__.build() (no CPU fallback exists)")
        # One HIP runtime per process: PyTorch bundles its own libamdhip64.  If this library pulled in the
        # system copy first, torch would later load a second runtime and find no device ("No HIP GPUs are
        # available").  Importing torch first makes both resolve to the copy torch loaded.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)  # (global: the development library resolves its internals against THIS copy)
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = _Lib(lib)
    return _lib


class _Lib:
    """The product library; names of the development interface resolve to libhalo_hip_dev.so, loaded at their first use."""

    def __init__(self, product):
        self._product = product
        self._dev = None

    def dev(self):
        if self._dev is None:
            if not os.path.exists(DEV_LIB_PATH):
                raise HaloError("libhalo_hip_dev.so is missing: run __graft_entry__.build()")
            dev = C.CDLL(DEV_LIB_PATH)
            for name, (res, args) in _DEV_SIGS.items():
                fn = getattr(dev, name)
                fn.restype = res
                fn.argtypes = args
            self._dev = dev
        return self._dev

    def __getattr__(self, name):
        if name in _DEV_SIGS:
            return getattr(self.dev(), name)
        return getattr(self._product, name)


def dev_hook(name: str, value: int) -> None:
    """fault injectors / forced test paths of the development library (csrc/tuning.hpp DevHooks); "reset" switches all off"""
    check(load().halo_dev_hook(name.encode(), int(value)))


def ptr(a):
    if a is None:
        return None
    assert isinstance(a, np.ndarray) and a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"], "need contiguous uint64 array"
    return a.ctypes.data_as(u64p)


def check(rc: int) -> None:
    if rc == HALO_OK:
        return
    msg = load().halo_last_error().decode()
    if rc == HALO_E_ASSERT:
        raise AssertionError(msg)
    if rc == HALO_E_REJECT:
        raise HaloReject(msg)
    raise HaloError("%s (code %d)" % (msg, rc))


class Context:
    """A commitment key resident on one GPU (consts.rs: N, GS)."""

    def __init__(self, bases=None, *, urs_n: int | None = None, first_index: int = 2, stride: int = 1, device: int = 0, devices=None):
        """devices=[d0, d1, ...]: a multi-device context (halo_ctx_create_multi): full context on d0, one index-block shard per entry"""
        lib = load()
        h = C.c_void_p()
        if devices is not None:
            assert stride == 1, "multi-device contexts hold index blocks"
            if len(devices) == 0:
                raise HaloError("ctx_create_multi: 1..64 device ids")
            devs = (C.c_int * len(devices))(*[int(d) for d in devices])
            device = int(devices[0])
            if bases is not None:
                bases = np.ascontiguousarray(bases, dtype=np.uint64).reshape(-1, 8)
                check(lib.halo_ctx_create_multi(devs, len(devices), ptr(bases), bases.shape[0], C.byref(h)))
            else:
                check(lib.halo_ctx_create_urs_multi(devs, len(devices), first_index, int(urs_n), C.byref(h)))
        elif bases is not None:
            bases = np.ascontiguousarray(bases, dtype=np.uint64).reshape(-1, 8)
            check(lib.halo_ctx_create(device, ptr(bases), bases.shape[0], C.byref(h)))
        elif stride != 1:
            check(lib.halo_ctx_create_urs_strided(device, first_index, stride, int(urs_n), C.byref(h)))
        else:
            check(lib.halo_ctx_create_urs(device, first_index, int(urs_n), C.byref(h)))
        self.h = h
        self.lib = lib
        self.device = device
        self._children = []  # weak references to the Ipa states of this context: they must go before it

    def clone(self):
        """a second context over the same resident key and tables (halo_ctx_clone): one per host thread"""
        h = C.c_void_p()
        check(self.lib.halo_ctx_clone(self.h, C.byref(h)))
        other = Context.__new__(Context)
        other.h, other.lib, other.device, other._children = h, self.lib, self.device, []
        return other

    def close(self):
        if getattr(self, "h", None):
            for ref in self._children:
                child = ref()
                if child is not None:
                    child.close()
            self._children = []
            self.lib.halo_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def size(self) -> int:
        return self.lib.halo_ctx_size(self.h)

    @property
    def n_devices(self) -> int:
        return self.lib.halo_ctx_devices(self.h)

    def read_bases(self, off=0, n=None):
        n = self.size - off if n is None else n
        out = np.zeros((n, 8), dtype=np.uint64)
        check(self.lib.halo_ctx_read_bases(self.h, off, n, ptr(out)))
        return out

    # ---- group.rs
    def msm(self, scalars, off=0, mont=True):
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
        out = np.zeros(12, dtype=np.uint64)
        check(self.lib.halo_msm(self.h, off, scalars.shape[0], ptr(scalars), int(mont), ptr(out)))
        return out

    def msm_begin(self, slot: int, scalars, off=0, mont=True):
        """host scalars, asynchronous: copy + launch on `slot`; msm_end(slot) collects"""
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
        check(self.lib.halo_msm_begin(self.h, slot, off, scalars.shape[0], ptr(scalars), int(mont)))

    def msm_end(self, slot: int):
        out = np.zeros(12, dtype=np.uint64)
        check(self.lib.halo_msm_end(self.h, slot, ptr(out)))
        return out

    def msm_dev(self, dptr: int, n: int, off=0, mont=True):
        out = np.zeros(12, dtype=np.uint64)
        check(self.lib.halo_msm_dev(self.h, off, n, C.c_void_p(dptr), int(mont), ptr(out)))
        return out

    def msm_dev_begin(self, slot: int, dptr: int, n: int, off=0, mont=True, part=0, parts=1):
        """parts > 1: only window shard `part` of `parts` (the partial results of all parts sum to the MSM)"""
        if parts == 1:
            check(self.lib.halo_msm_dev_begin(self.h, slot, off, n, C.c_void_p(dptr), int(mont)))
        else:
            check(self.lib.halo_msm_dev_begin_part(self.h, slot, off, n, C.c_void_p(dptr), int(mont), part, parts))

    def msm_dev_end(self, slot: int):
        out = np.zeros(12, dtype=np.uint64)
        check(self.lib.halo_msm_dev_end(self.h, slot, ptr(out)))
        return out

    def msm_dev_batch_begin(self, slot: int, dptrs, n: int, off=0, mont=True, part=0, parts=1):
        """len(dptrs) (1..8) MSMs over the same bases, one resident scalar array each, as one launch sequence;
        parts > 1: window shard `part` of each"""
        arr = (C.c_void_p * len(dptrs))(*[C.c_void_p(int(p)) for p in dptrs])
        check(self.lib.halo_msm_dev_batch_begin(self.h, slot, off, n, arr, len(dptrs), int(mont), part, parts))

    def msm_dev_batch_end(self, slot: int, batch: int):
        out = np.zeros((batch, 12), dtype=np.uint64)
        check(self.lib.halo_msm_dev_batch_end(self.h, slot, batch, ptr(out)))
        return out

    def msm_points(self, pts_jac, scalars):
        pts_jac = np.ascontiguousarray(pts_jac, dtype=np.uint64).reshape(-1, 12)
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
        m = min(len(pts_jac), len(scalars))  # msm_unchecked zips to the shorter input
        out = np.zeros(12, dtype=np.uint64)
        check(self.lib.halo_msm_points(self.h, ptr(pts_jac), ptr(scalars), m, ptr(out)))
        return out

    def msm_affine(self, bases_affine, scalars, mont=True):
        """point_dot_affine over bases that are not part of the context's key (uploaded for this call)"""
        bases_affine = np.ascontiguousarray(bases_affine, dtype=np.uint64).reshape(-1, 8)
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
        m = min(len(bases_affine), len(scalars))  # msm_unchecked zips to the shorter input
        out = np.zeros(12, dtype=np.uint64)
        check(self.lib.halo_msm_affine(self.h, ptr(bases_affine), ptr(scalars), m, int(mont), ptr(out)))
        return out

    def scalar_dot(self, xs, ys):
        xs = np.ascontiguousarray(xs, dtype=np.uint64).reshape(-1, 4)
        ys = np.ascontiguousarray(ys, dtype=np.uint64).reshape(-1, 4)
        out = np.zeros(4, dtype=np.uint64)
        check(self.lib.halo_scalar_dot(self.h, ptr(xs), ptr(ys), min(len(xs), len(ys)), ptr(out)))
        return out

    def powers(self, z, n):
        out = np.zeros((n, 4), dtype=np.uint64)
        check(self.lib.halo_powers(self.h, ptr(np.ascontiguousarray(z, dtype=np.uint64)), n, ptr(out)))
        return out

    def poly_eval(self, coeffs, z):
        coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(-1, 4)
        out = np.zeros(4, dtype=np.uint64)
        check(self.lib.halo_poly_eval(self.h, ptr(coeffs), coeffs.shape[0], ptr(np.ascontiguousarray(z, dtype=np.uint64)), ptr(out)))
        return out

    # ---- h(X)
    def h_coeffs(self, xis):
        xis = np.ascontiguousarray(xis, dtype=np.uint64).reshape(-1, 4)
        lg = xis.shape[0] - 1
        out = np.zeros((1 << lg, 4), dtype=np.uint64)
        check(self.lib.halo_h_coeffs(self.h, ptr(xis), lg, ptr(out)))
        return out

    def h_commit(self, xis):
        xis = np.ascontiguousarray(xis, dtype=np.uint64).reshape(-1, 4)
        out = np.zeros(12, dtype=np.uint64)
        check(self.lib.halo_h_commit(self.h, ptr(xis), xis.shape[0] - 1, ptr(out)))
        return out

    def h_eval_batch(self, xis, z):
        xis = np.ascontiguousarray(xis, dtype=np.uint64)
        m, lg1 = xis.shape[0], xis.shape[1]
        out = np.zeros((m, 4), dtype=np.uint64)
        check(self.lib.halo_h_eval_batch(self.h, ptr(xis), m, lg1 - 1, ptr(np.ascontiguousarray(z, dtype=np.uint64)), ptr(out)))
        return out

    def h_accumulate(self, h0, xis, alphas, out=None):
        """out: an (n, 4) uint64 array to write into (a caller that keeps its buffer: no fresh pages for the driver to pin)"""
        xis = np.ascontiguousarray(xis, dtype=np.uint64)
        m, lg1 = xis.shape[0], xis.shape[1]
        if out is None:
            out = np.empty((1 << (lg1 - 1), 4), dtype=np.uint64)
        assert out.shape == (1 << (lg1 - 1), 4) and out.dtype == np.uint64 and out.flags["C_CONTIGUOUS"]
        h0 = None if h0 is None else np.ascontiguousarray(h0, dtype=np.uint64)
        check(self.lib.halo_h_accumulate(self.h, ptr(h0), ptr(xis), ptr(np.ascontiguousarray(alphas, dtype=np.uint64)), m, lg1 - 1, ptr(out)))
        return out

    # ---- measurement
    def prof_enable(self, on=1):
        """1: every launch; 2: dominant kernels only; 0: off"""
        check(self.lib.halo_prof_enable(self.h, int(on)))

    def prof_reset(self):
        check(self.lib.halo_prof_reset(self.h))

    def prof(self):
        out = {}
        for i in range(self.lib.halo_prof_count(self.h)):
            name, ms, cnt = C.c_char_p(), C.c_double(), C.c_long()
            check(self.lib.halo_prof_get(self.h, i, C.byref(name), C.byref(ms), C.byref(cnt)))
            out[name.value.decode()] = (ms.value, cnt.value)
        return out

    def bench_fr_kernel(self, which: int, n: int, reps: int):
        """reps back-to-back launches of one Fr kernel (0 powers, 1 poly_eval, 2 dot, 3 dot2 of a round, 4 h_coeffs, 5 fold_scalars, 6 axpy)"""
        check(self.lib.halo_bench_fr_kernel(self.h, which, n, reps))

    def rng_scalars_dev(self, state: int, n: int, dptr: int) -> int:
        """Fill device memory with n scalars of the SplitMix64 stream; returns the advanced state."""
        st = C.c_uint64(state)
        check(self.lib.halo_rng_scalars_dev(self.h, C.byref(st), n, C.c_void_p(dptr)))
        return st.value

    def set_graphs(self, on: bool):
        check(self.lib.halo_set_graphs(self.h, int(on)))

    def set_ipa_switch(self, size):
        check(self.lib.halo_set_ipa_switch(self.h, size))

    def set_window_bits(self, c):
        check(self.lib.halo_set_window_bits(self.h, c))

    def set_reduce_span(self, span):
        check(self.lib.halo_set_reduce_span(self.h, span))

    def set_sort_mode(self, mode):
        check(self.lib.halo_set_sort_mode(self.h, mode))

    def set_batch_verify(self, on):
        check(self.lib.halo_set_batch_verify(self.h, int(on)))

    def set_fold_levels(self, levels):
        check(self.lib.halo_set_fold_levels(self.h, levels))

    def set_fold_async(self, mode):
        """-1 automatic (opens of <= 2^18 points), 0 never, 1 wherever possible"""
        check(self.lib.halo_set_fold_async(self.h, int(mode)))

    def set_fold_table(self, mode):
        check(self.lib.halo_set_fold_table(self.h, mode))

    def info(self, what: int) -> int:
        """0: MSM table bytes, 1: fold table bytes, 2: fold table build time in microseconds"""
        return self.lib.halo_ctx_info(self.h, what)

    def set_memory_budget(self, nbytes: int):
        """budget for optional table memory on this context's device, process-wide (halo_set_memory_budget)"""
        check(self.lib.halo_set_memory_budget(self.h, int(nbytes)))

    def set_table_mode(self, mode):
        check(self.lib.halo_set_table_mode(self.h, mode))

    def set_small_path(self, mode):
        check(self.lib.halo_set_small_path(self.h, mode))

    def set_task_len(self, n):
        check(self.lib.halo_set_task_len(self.h, n))

    # ---- primitive hooks
    def field_op(self, field, op, a, b=None):
        a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
        b = None if b is None else np.ascontiguousarray(b, dtype=np.uint64).reshape(-1, 4)
        out = np.zeros_like(a)
        check(self.lib.halo_test_field_op(self.h, field, op, ptr(a), ptr(b), a.shape[0], ptr(out)))
        return out

    def point_op(self, op, a, b=None):
        a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 12)
        b = None if b is None else np.ascontiguousarray(b, dtype=np.uint64)
        out = np.zeros_like(a)
        check(self.lib.halo_test_point_op(self.h, op, ptr(a), ptr(b), a.shape[0], ptr(out)))
        return out


class Ipa:
    """Device-resident state of pcdl::open's halving loop (pcdl.rs:183-231)."""

    def __init__(self, ctx: Context, n: int, coeffs, z, *, stride: int = 1, offset: int = 0, z_vec=None):
        """Default: c = coeffs zero-padded, z-vector = powers of z.  stride/offset: the cyclic shard
        z^(offset + j*stride).  z_vec: explicit vectors (coeffs and z_vec both of length n)."""
        coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(-1, 4)
        h = C.c_void_p()
        if z_vec is not None:
            z_vec = np.ascontiguousarray(z_vec, dtype=np.uint64).reshape(-1, 4)
            assert coeffs.shape[0] == n == z_vec.shape[0]
            check(ctx.lib.halo_ipa_begin_vectors(ctx.h, n, ptr(coeffs), ptr(z_vec), C.byref(h)))
        elif stride != 1 or offset != 0:
            check(ctx.lib.halo_ipa_begin_strided(ctx.h, n, ptr(coeffs), coeffs.shape[0], ptr(np.ascontiguousarray(z, dtype=np.uint64)),
                                                 stride, offset, C.byref(h)))
        else:
            check(ctx.lib.halo_ipa_begin(ctx.h, n, ptr(coeffs), coeffs.shape[0], ptr(np.ascontiguousarray(z, dtype=np.uint64)), C.byref(h)))
        self.h, self.ctx = h, ctx
        import weakref
        ctx._children.append(weakref.ref(self))

    def dot_cz(self):
        out = np.zeros(4, dtype=np.uint64)
        check(self.ctx.lib.halo_ipa_dot_cz(self.h, ptr(out)))
        return out

    def round_lr_partial(self):
        """-> one record L 12 | R 12 | dot_l 4 | dot_r 4 (no H' terms)"""
        rec = np.zeros(32, dtype=np.uint64)
        check(self.ctx.lib.halo_ipa_round_lr_partial(self.h, ptr(rec[:12]), ptr(rec[12:24]), ptr(rec[24:])))
        return rec

    def hiding_partial(self, rng_state: int, deg: int, z, stride: int, offset: int):
        out = np.zeros(12, dtype=np.uint64)
        check(self.ctx.lib.halo_ipa_hiding_partial(self.h, rng_state, deg, ptr(np.ascontiguousarray(z, dtype=np.uint64)), stride, offset, ptr(out)))
        return out

    def apply_hiding(self, alpha):
        check(self.ctx.lib.halo_ipa_apply_hiding(self.h, ptr(np.ascontiguousarray(alpha, dtype=np.uint64))))

    def finish_z(self):
        U, c, z0 = np.zeros(12, dtype=np.uint64), np.zeros(4, dtype=np.uint64), np.zeros(4, dtype=np.uint64)
        check(self.ctx.lib.halo_ipa_finish_z(self.h, ptr(U), ptr(c), ptr(z0)))
        return U, c, z0

    def round_lr(self, H_prime):
        L, R = np.zeros(12, dtype=np.uint64), np.zeros(12, dtype=np.uint64)
        check(self.ctx.lib.halo_ipa_round_lr(self.h, ptr(np.ascontiguousarray(H_prime, dtype=np.uint64)), ptr(L), ptr(R)))
        return L, R

    def round_fold(self, xi, xi_inv):
        check(self.ctx.lib.halo_ipa_round_fold(self.h, ptr(np.ascontiguousarray(xi, dtype=np.uint64)), ptr(np.ascontiguousarray(xi_inv, dtype=np.uint64))))

    def finish(self):
        U, c = np.zeros(12, dtype=np.uint64), np.zeros(4, dtype=np.uint64)
        check(self.ctx.lib.halo_ipa_finish(self.h, ptr(U), ptr(c)))
        return U, c

    def __len__(self):
        return self.ctx.lib.halo_ipa_len(self.h)

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.halo_ipa_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def open_start(Cm, z, v_parts):
    """v = sum of the shards' <c, z>; xi_0 = rho_0(C, z, v); H' = xi_0 H"""
    v_parts = np.ascontiguousarray(v_parts, dtype=np.uint64).reshape(-1, 4)
    v, xi, Hp = np.zeros(4, dtype=np.uint64), np.zeros(4, dtype=np.uint64), np.zeros(12, dtype=np.uint64)
    check(load().halo_open_start(ptr(np.ascontiguousarray(Cm, dtype=np.uint64)), ptr(np.ascontiguousarray(z, dtype=np.uint64)), ptr(v_parts),
                                 v_parts.shape[0], ptr(v), ptr(xi), ptr(Hp)))
    return v, xi, Hp


def open_hiding_combine(Cm, z, v_parts, cbar_parts, w, rng_state: int, deg: int):
    """-> C_bar, alpha, w', C', advanced rng state"""
    v_parts = np.ascontiguousarray(v_parts, dtype=np.uint64).reshape(-1, 4)
    cbar_parts = np.ascontiguousarray(cbar_parts, dtype=np.uint64).reshape(-1, 12)
    st = C.c_uint64(rng_state)
    Cbar, alpha, wp, Cp = (np.zeros(12, dtype=np.uint64), np.zeros(4, dtype=np.uint64), np.zeros(4, dtype=np.uint64),
                           np.zeros(12, dtype=np.uint64))
    check(load().halo_open_hiding_combine(ptr(np.ascontiguousarray(Cm, dtype=np.uint64)), ptr(np.ascontiguousarray(z, dtype=np.uint64)),
                                          ptr(v_parts), ptr(cbar_parts), v_parts.shape[0], ptr(np.ascontiguousarray(w, dtype=np.uint64)),
                                          C.byref(st), deg, ptr(Cbar), ptr(alpha), ptr(wp), ptr(Cp)))
    return Cbar, alpha, wp, Cp, st.value


def open_combine(parts, Hp, xi_prev):
    """parts (P, 32) in rank order -> L, R, xi_next, xi_next^-1"""
    parts = np.ascontiguousarray(parts, dtype=np.uint64).reshape(-1, 32)
    L, R = np.zeros(12, dtype=np.uint64), np.zeros(12, dtype=np.uint64)
    xi, xi_inv = np.zeros(4, dtype=np.uint64), np.zeros(4, dtype=np.uint64)
    check(load().halo_open_combine(ptr(parts), parts.shape[0], ptr(np.ascontiguousarray(Hp, dtype=np.uint64)),
                                   ptr(np.ascontiguousarray(xi_prev, dtype=np.uint64)), ptr(L), ptr(R), ptr(xi), ptr(xi_inv)))
    return L, R, xi, xi_inv


# the caller's all-gather as the library calls it (include/halo_accumulation.h halo_allgather_fn)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_size_t, C.POINTER(C.c_uint64))


def make_allgather_callback(allgather, world: int):
    """allgather(arr) -> (P, len) uint64 wrapped for halo_pcdl_open_sharded / _check_sharded; keep the returned object alive
    for the duration of the call.  An exception inside the collective is kept in cb.error and reported as a failure."""
    def _cb(_user, send, words, recv):
        try:
            a = np.ctypeslib.as_array(send, shape=(words,)).copy()
            out = np.ascontiguousarray(allgather(a), dtype=np.uint64).reshape(-1)
            assert out.size == world * words
            np.ctypeslib.as_array(recv, shape=(world * words,))[:] = out
            return 0
        except Exception as e:  # noqa: BLE001 -- must not unwind through the C frame
            cb.error = e
            return 1
    cb = ALLGATHER_FN(_cb)
    cb.error = None
    return cb


def open_tail(recs, Hp, xi_prev):
    """recs (P, 20): the gathered (G_i | c_i | z_i) in index order -> (Ls (lg P, 12), Rs (lg P, 12), U, c): the last lg P rounds on the host"""
    recs = np.ascontiguousarray(recs, dtype=np.uint64).reshape(-1, 20)
    P = recs.shape[0]
    lg = max(P.bit_length() - 1, 0)
    Ls, Rs = np.zeros((max(lg, 1), 12), dtype=np.uint64), np.zeros((max(lg, 1), 12), dtype=np.uint64)
    U, c = np.zeros(12, dtype=np.uint64), np.zeros(4, dtype=np.uint64)
    check(load().halo_open_tail(ptr(recs), P, ptr(np.ascontiguousarray(Hp, dtype=np.uint64)), ptr(np.ascontiguousarray(xi_prev, dtype=np.uint64)),
                                ptr(Ls), ptr(Rs), ptr(U), ptr(c)))
    return Ls[:lg], Rs[:lg], U, c


def _encode(fn, blob, cap):
    blob = np.ascontiguousarray(blob, dtype=np.uint64)
    buf = C.create_string_buffer(cap)
    n = C.c_size_t()
    check(fn(ptr(blob), buf, cap, C.byref(n)))
    return buf.raw[: n.value]


def _decode(fn, data, words_of_lg):
    data = bytes(data)
    out = np.zeros(words_of_lg(40), dtype=np.uint64)
    lg = C.c_size_t()
    check(fn(data, len(data), ptr(out), out.shape[0], C.byref(lg)))
    return out[: words_of_lg(lg.value)].copy()


def proof_encode(proof):
    """EvalProof blob -> bytes (ark-serialize compressed layout of pcdl.rs:22-30)"""
    lib = load()
    return _encode(lib.halo_proof_encode, proof, lib.halo_proof_encoded_size(int(proof[1]), 1))


def proof_decode(data):
    lib = load()
    return _decode(lib.halo_proof_decode, data, lib.halo_proof_words)


def instance_encode(inst):
    lib = load()
    return _encode(lib.halo_instance_encode, inst, lib.halo_instance_encoded_size(int(inst[22]), 1))


def instance_decode(data):
    lib = load()
    return _decode(lib.halo_instance_decode, data, lib.halo_instance_words)


def accumulator_encode(acc):
    lib = load()
    return _encode(lib.halo_accumulator_encode, acc, lib.halo_accumulator_encoded_size(int(acc[22])))


def accumulator_decode(data):
    lib = load()
    return _decode(lib.halo_accumulator_decode, data, lib.halo_accumulator_words)


def point_sum(pts_jac):
    """Sum of Jacobian points in index order (host): the combine step of the sharded MSM."""
    pts_jac = np.ascontiguousarray(pts_jac, dtype=np.uint64).reshape(-1, 12)
    out = np.zeros(12, dtype=np.uint64)
    check(load().halo_point_sum(ptr(pts_jac), pts_jac.shape[0], ptr(out)))
    return out


def public_points():
    S, H = np.zeros(12, dtype=np.uint64), np.zeros(12, dtype=np.uint64)
    check(load().halo_public_points(ptr(S), ptr(H)))
    return S, H
