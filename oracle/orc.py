"""ctypes binding of oracle/_build/liborc.so (CPU restatement) -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
Arrays are numpy uint64, Montgomery limbs, exactly as the product's C ABI takes them.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# ORC_NATIVE=1: the library compiled with -march=native on the host it runs on; falls back to the portable build (and says so in
# BUILD_FLAGS) if that compile is not possible.  The compiler is the Makefile's choice (the image's clang, else gcc) unless
# select_fastest() below has picked another.
NATIVE = os.environ.get("ORC_NATIVE") == "1"
LIB_PATH = os.path.join(HERE, "_build", "liborc_native.so" if NATIVE else "liborc.so")
BUILD_FLAGS = "-O3 -march=native" if NATIVE else "-O3 -march=x86-64-v2"
CLANG = "/opt/rocm/lib/llvm/bin/clang"


def _host() -> str:
    try:
        return [l for l in open("/proc/cpuinfo") if l.startswith("model name")][0].strip()
    except (OSError, IndexError):
        return "unknown"


def build(force: bool = False) -> str:
    global LIB_PATH, BUILD_FLAGS
    src = [os.path.join(HERE, f) for f in ("halo_cpu.c", "halo_cpu.h", "Makefile")]
    stamp = os.path.join(HERE, "_build", "native.host")
    host = ""
    if NATIVE:  # a native binary belongs to the CPU it was built on: rebuilt when the host's CPU model differs from the stamp
        host = _host()
        force = force or not os.path.exists(stamp) or open(stamp).read() != host
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(f) for f in src):
        try:
            if NATIVE and os.path.exists(LIB_PATH):
                os.remove(LIB_PATH)
            subprocess.check_call(["make", "-C", HERE, "-s"] + (["native"] if NATIVE else []))
            if NATIVE:
                with open(stamp, "w") as f:
                    f.write(host)
        except (subprocess.CalledProcessError, OSError):
            if not NATIVE:
                raise
            LIB_PATH, BUILD_FLAGS = os.path.join(HERE, "_build", "liborc.so"), "-O3 -march=x86-64-v2 (native build failed on this host)"
            if not os.path.exists(LIB_PATH):
                subprocess.check_call(["make", "-C", HERE, "-s"])
    return LIB_PATH


def _load(path):
    l = C.CDLL(path)
    l.orc_last_error.restype = C.c_char_p
    l.orc_rng_u64.restype = C.c_uint64
    l.orc_bench_fq_mul.restype = C.c_double
    return l


def select_fastest(log_n: int = 13):
    """bench.py's cpu_baseline leg: build the restatement with every compiler the host has (the image's clang, gcc) for the
    portable instruction set and for the host's own (-march=native), time one MSM of 2^log_n points with each and keep the
    fastest for everything that follows -- which one that is depends on compiler and CPU (gcc 11's native code was the slower
    one on an EPYC 9575F; clang's carry chains are half as long as gcc 11's).  -> {"<compiler> <flags>": seconds}"""
    global _lib, LIB_PATH, BUILD_FLAGS, NATIVE
    import shutil
    import time
    timings, libs = {}, {}
    compilers = [c for c in (CLANG if os.path.exists(CLANG) else shutil.which("clang"), shutil.which("gcc")) if c]
    for cc in compilers:
        for march in ("x86-64-v2", "native"):
            tag = "%s -O3 -march=%s" % (os.path.basename(cc), march)
            out = os.path.join("_build", "liborc_%s_%s.so" % (os.path.basename(cc), march))
            try:
                subprocess.check_call(["make", "-C", HERE, "-s", "-B", "CC=" + cc, "MARCH=" + march, "OUT=" + out],
                                      stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                l = _load(os.path.join(HERE, out))
            except Exception:  # noqa: BLE001 -- this compiler / flag pair does not build here: the others are what there is
                continue
            _lib = l
            n = 1 << log_n
            gs, (sc, _) = urs_affine(2, n), rng_scalars(7, n)
            best = None
            for _ in range(3):
                t0 = time.perf_counter()
                msm_affine(gs, sc)
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            timings[tag] = best
            libs[tag] = (l, os.path.join(HERE, out), march == "native")
    if not timings:  # no compiler on this host: the prebuilt portable library is what there is
        _lib = None
        lib()
        return {}
    BUILD_FLAGS = min(timings, key=timings.get)
    _lib, LIB_PATH, NATIVE = libs[BUILD_FLAGS]
    return timings


def ns_per_field_product(iters: int = 4_000_000) -> float:
    """Throughput of the loaded library's Fq Montgomery product on this host, ns (best of three)."""
    o = z(4)
    return min(lib().orc_bench_fq_mul(C.c_size_t(iters), ptr(o)) for _ in range(3))


u64p = C.POINTER(C.c_uint64)


class PP(C.Structure):
    _fields_ = [("S", C.c_uint64 * 12), ("H", C.c_uint64 * 12), ("GS", u64p), ("N", C.c_size_t)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = _load(LIB_PATH)
    return _lib


def ptr(a):
    if a is None:
        return None
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u64p)


def proof_words(lg):
    return 2 + 24 * lg + 32


def instance_words(lg):
    return 21 + proof_words(lg)


def acc_words(lg):
    return instance_words(lg) + 24


def last_error() -> str:
    return lib().orc_last_error().decode()


def z(n):
    return np.zeros(n, dtype=np.uint64)


# ---- thin functional wrappers -------------------------------------------------
def fr_to_mont(x: int):
    a = np.array([(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)
    o = z(4); lib().orc_fr_to_mont(ptr(a), ptr(o)); return o


def fr_from_mont(a) -> int:
    o = z(4); lib().orc_fr_from_mont(ptr(np.ascontiguousarray(a, dtype=np.uint64)), ptr(o))
    return sum(int(o[i]) << (64 * i) for i in range(4))


def fq_from_mont(a) -> int:
    o = z(4); lib().orc_fq_from_mont(ptr(np.ascontiguousarray(a, dtype=np.uint64)), ptr(o))
    return sum(int(o[i]) << (64 * i) for i in range(4))


def scalars_to_mont(xs):
    out = np.zeros((len(xs), 4), dtype=np.uint64)
    for i, x in enumerate(xs):
        out[i] = fr_to_mont(int(x))
    return out


def point_canonical(jac):
    """-> None for infinity, else (x, y) canonical ints."""
    out = (C.c_uint8 * 64)()
    inf = lib().orc_point_canonical(ptr(np.ascontiguousarray(jac, dtype=np.uint64)), out)
    if inf:
        return None
    b = bytes(out)
    return int.from_bytes(b[:32], "little"), int.from_bytes(b[32:], "little")


def affine_canonical(aff):
    j = z(12); lib().orc_affine_to_jac(ptr(np.ascontiguousarray(aff, dtype=np.uint64)), ptr(j))
    return point_canonical(j)


def urs_affine(first: int, count: int):
    out = np.zeros((count, 8), dtype=np.uint64)
    lib().orc_urs_affine(C.c_uint64(first), C.c_size_t(count), ptr(out))
    return out


def rng_scalars(seed: int, n: int):
    st = C.c_uint64(seed)
    out = np.zeros((n, 4), dtype=np.uint64)
    lib().orc_rng_scalars(C.byref(st), C.c_size_t(n), ptr(out))
    return out, st.value


def make_pp(gs):
    pp = PP()
    lib().orc_pp_init(C.byref(pp), ptr(gs), C.c_size_t(gs.shape[0]))
    pp._keep = gs
    return pp


def msm_affine(bases, scalars):
    o = z(12); lib().orc_msm_affine(ptr(bases), ptr(scalars), C.c_size_t(min(len(bases), len(scalars))), ptr(o)); return o


def msm_naive(bases, scalars):
    o = z(12); lib().orc_msm_naive(ptr(bases), ptr(scalars), C.c_size_t(min(len(bases), len(scalars))), ptr(o)); return o


def msm_jac(pts, scalars):
    o = z(12); lib().orc_msm_jac(ptr(pts), ptr(scalars), C.c_size_t(min(len(pts), len(scalars))), ptr(o)); return o


def scalar_dot(xs, ys):
    o = z(4); lib().orc_scalar_dot(ptr(xs), ptr(ys), C.c_size_t(min(len(xs), len(ys))), ptr(o)); return o


def powers(zm, n):
    o = np.zeros((n, 4), dtype=np.uint64); lib().orc_powers(ptr(zm), C.c_size_t(n), ptr(o)); return o


def h_coeffs(xis):
    lg = len(xis) - 1
    o = np.zeros((1 << lg, 4), dtype=np.uint64); lib().orc_h_coeffs(ptr(xis), C.c_size_t(lg), ptr(o)); return o


def h_eval(xis, zm):
    o = z(4); lib().orc_h_eval(ptr(xis), C.c_size_t(len(xis) - 1), ptr(zm), ptr(o)); return o


def poly_eval(coeffs, zm):
    o = z(4); lib().orc_poly_eval(ptr(coeffs), C.c_size_t(len(coeffs)), ptr(zm), ptr(o)); return o


def pcdl_commit(pp, coeffs, d, w=None):
    o = z(12)
    rc = lib().orc_pcdl_commit(C.byref(pp), ptr(coeffs), C.c_size_t(len(coeffs)), C.c_size_t(d), ptr(w), ptr(o))
    if rc:
        raise AssertionError(last_error())
    return o


def pcdl_open(pp, seed, coeffs, Cj, d, zm, w=None):
    lg = (d + 1).bit_length() - 1
    st = C.c_uint64(seed)
    pf = z(proof_words(lg))
    rc = lib().orc_pcdl_open(C.byref(pp), C.byref(st), ptr(coeffs), C.c_size_t(len(coeffs)), ptr(Cj), C.c_size_t(d),
                             ptr(zm), ptr(w), ptr(pf))
    if rc:
        raise AssertionError(last_error())
    return pf, st.value


def pcdl_succinct_check(pp, Cj, d, zm, vm, pf):
    lg = (d + 1).bit_length() - 1
    xis = np.zeros((lg + 1, 4), dtype=np.uint64); U = z(12)
    rc = lib().orc_pcdl_succinct_check(C.byref(pp), ptr(Cj), C.c_size_t(d), ptr(zm), ptr(vm), ptr(pf), ptr(xis), ptr(U))
    if rc:
        raise ValueError(last_error())
    return xis, U


def pcdl_check(pp, Cj, d, zm, vm, pf):
    rc = lib().orc_pcdl_check(C.byref(pp), ptr(Cj), C.c_size_t(d), ptr(zm), ptr(vm), ptr(pf))
    if rc:
        raise ValueError(last_error())


def random_instance(pp, seed, d):
    lg = (d + 1).bit_length() - 1
    st = C.c_uint64(seed); inst = z(instance_words(lg))
    rc = lib().orc_random_instance(C.byref(pp), C.byref(st), C.c_size_t(d), ptr(inst))
    if rc:
        raise AssertionError(last_error())
    return inst, st.value


def acc_prover(pp, seed, d, instances):
    lg = (d + 1).bit_length() - 1
    qs = np.ascontiguousarray(np.concatenate(instances)) if len(instances) else z(0)
    st = C.c_uint64(seed); acc = z(acc_words(lg))
    rc = lib().orc_acc_prover(C.byref(pp), C.byref(st), C.c_size_t(d), ptr(qs), C.c_size_t(len(instances)), ptr(acc))
    if rc:
        raise ValueError(last_error())
    return acc, st.value


def acc_verifier(pp, d, instances, acc):
    qs = np.ascontiguousarray(np.concatenate(instances)) if len(instances) else z(0)
    rc = lib().orc_acc_verifier(C.byref(pp), C.c_size_t(d), ptr(qs), C.c_size_t(len(instances)), ptr(acc))
    if rc:
        raise ValueError(last_error())


def acc_decider(pp, acc):
    rc = lib().orc_acc_decider(C.byref(pp), ptr(acc))
    if rc:
        raise ValueError(last_error())
