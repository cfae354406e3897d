"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU, exports every
symbol include/halo_accumulation.h declares, its host-only entry points are right, and every
compute entry point fails loudly (no CPU fallback) when there is no device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hal():
    import halo_accumulation_amd as h
    h.build()
    return h


def header_symbols(name="halo_accumulation.h"):
    text = open(os.path.join(ROOT, "include", name)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(halo_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(hal):
    lib = C.CDLL(hal._lib.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 45
    for s in syms:
        assert hasattr(lib, s), "missing export: " + s
    # and the Python binding covers the same set
    assert sorted(hal._lib.declared_symbols()) == syms


def test_development_library_is_separate(hal):
    """the shipped library exports no test hook, bench hook or fault injector and reads no injector from the
    environment; they live in libhalo_hip_dev.so (include/halo_accumulation_dev.h), which links against the product library.
    The environment is read in ONE translation unit (csrc/tuning.hip)."""
    import subprocess
    prod = subprocess.check_output(["nm", "-D", "--defined-only", hal._lib.LIB_PATH], text=True)
    exported = sorted(set(re.findall(r" T (halo_[a-z0-9_]+)$", prod, flags=re.M)))
    assert exported == header_symbols(), "the product library exports exactly what its header declares"
    assert not [s for s in exported if s.startswith(("halo_test", "halo_bench", "halo_dev_"))]
    dev_syms = header_symbols("halo_accumulation_dev.h")
    dev = subprocess.check_output(["nm", "-D", "--defined-only", hal._lib.DEV_LIB_PATH], text=True)
    dev_exported = sorted(set(re.findall(r" T (halo_[a-z0-9_]+)$", dev, flags=re.M)))
    assert dev_exported == dev_syms == sorted(hal._lib.declared_dev_symbols())
    assert "halo_dev_hook" in dev_syms and "halo_test_field_op" in dev_syms
    needed = subprocess.check_output(["readelf", "-d", hal._lib.DEV_LIB_PATH], text=True)
    assert "libhalo_hip.so" in needed, "the development library links against the product library (one state per process)"
    # no injector name in the product binary, no getenv outside tuning.hip
    blob = open(hal._lib.LIB_PATH, "rb").read()
    assert b"HALO_TEST_" not in blob
    csrc = os.path.join(ROOT, "halo-accumulation_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".hpp", ".cpp")) and f != "tuning.hip":
            assert "getenv" not in open(os.path.join(csrc, f)).read(), f + ": the environment is read in tuning.hip only"
    # the hooks work through the development library and are off by default (host-only call: no GPU needed)
    lib = hal.load()
    assert lib.halo_dev_hook(b"table_fail", 1) == 0 and lib.halo_dev_hook(b"reset", 0) == 0
    assert lib.halo_dev_hook(b"no_such_hook", 1) == hal._lib.HALO_E_ARG


def test_environment_is_parsed_once_and_strictly(hal):
    """csrc/tuning.hip: defaults, a valid HALO_HOST_SPLIT, invalid ones (ignored with a line on stderr, the default stays), clamps."""
    import subprocess, sys
    code = ("import sys; sys.path.insert(0, %r); import halo_accumulation_amd as h; l = h.load(); "
            "print(' '.join(str(l.halo_dev_tuning(n.encode())) for n in sys.argv[1:]))" % ROOT)
    names = ["host_pieces", "host_split0", "host_split1", "host_split2", "fold_table_after", "graph_cache", "pow_e", "memory_budget", "tagged"]

    def run(env):
        e = {k: v for k, v in os.environ.items() if not k.startswith("HALO_")}
        e.update(env)
        out = subprocess.run([sys.executable, "-c", code] + names, env=e, capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr[-500:]
        return [int(x) for x in out.stdout.split()], out.stderr

    assert run({})[0] == [2, 4, 12, 0, 8, 8, 0, -1, 1]
    assert run({"HALO_HOST_SPLIT": "2,6,8", "HALO_FOLD_TABLE_AFTER": "0", "HALO_MEMORY_BUDGET": "3G", "HALO_TAGGED": "0"})[0] == [3, 2, 6, 8, 0, 8, 0, 3072, 0]
    assert run({"HALO_HOST_SPLIT": "16"})[0][:3] == [1, 16, 0]
    for bad in ("5,5", "0,16", "4,4,4,4,4", "x", "20"):
        vals, err = run({"HALO_HOST_SPLIT": bad})
        assert vals[:3] == [2, 4, 12] and "HALO_HOST_SPLIT" in err, bad
    vals, err = run({"HALO_GRAPH_CACHE": "99", "HALO_POW_E": "12"})
    assert vals[5] == 8 and vals[6] == 0 and "HALO_POW_E" in err


def test_optional_rccl_library(hal):
    """include/halo_rccl.h / libhalo_rccl.so: halo_allgather_fn over RCCL for hosts without a collective layer of their own.
    Optional: the core library neither links nor loads a collective library."""
    import subprocess
    from halo_accumulation_amd import rccl
    core = subprocess.check_output(["ldd", hal._lib.LIB_PATH], text=True)
    assert "rccl" not in core and "nccl" not in core and "mpi" not in core
    if not rccl.available():
        pytest.skip("libhalo_rccl.so not built (no librccl in this image)")
    out = subprocess.check_output(["nm", "-D", "--defined-only", rccl.LIB_PATH], text=True)
    exported = sorted(set(re.findall(r" T (halo_[a-z0-9_]+)$", out, flags=re.M)))
    assert exported == header_symbols("halo_rccl.h")
    assert "librccl" in subprocess.check_output(["ldd", rccl.LIB_PATH], text=True)
    assert "libhalo_hip" not in subprocess.check_output(["ldd", rccl.LIB_PATH], text=True), "independent of the core library: only the typedef is shared"
    # the callback type of the core header and the exported function agree (same parameter list)
    core_h = open(os.path.join(ROOT, "include", "halo_accumulation.h")).read()
    rccl_h = open(os.path.join(ROOT, "include", "halo_rccl.h")).read()
    assert "typedef int (*halo_allgather_fn)(void *user, const uint64_t *send, size_t words, uint64_t *recv);" in core_h
    assert "int halo_allgather_rccl(void *user, const uint64_t *send, size_t words, uint64_t *recv);" in rccl_h


def test_public_points_match_consts_rs(hal, kat):
    S, H = hal._lib.public_points()
    hx = lambda p: ["%064x" % p[0], "%064x" % p[1]]
    assert hx(orc.point_canonical(S)) == kat["S"]
    assert hx(orc.point_canonical(H)) == kat["H"]


def test_point_sum_host(hal, urs4096):
    pts = np.zeros((5, 12), dtype=np.uint64)
    want = None
    for i in range(5):
        orc.lib().orc_affine_to_jac(orc.ptr(urs4096[i]), orc.ptr(pts[i]))
    pts[3, 8:] = 0  # one infinity among them
    acc = orc.z(12); acc[8:] = 0; acc[0] = 1; acc[4] = 1
    acc = np.array([1, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0], dtype=np.uint64)
    for i in range(5):
        o = orc.z(12); orc.lib().orc_point_add(orc.ptr(acc), orc.ptr(pts[i]), orc.ptr(o)); acc = o
    assert hal._lib.point_sum(pts).tolist() == acc.tolist()
    assert orc.point_canonical(hal._lib.point_sum(pts[:0])) is None


def test_layout_sizes_agree_with_oracle(hal):
    lib = hal.load()
    for lg in (0, 1, 5, 20):
        assert lib.halo_proof_words(lg) == orc.proof_words(lg)
        assert lib.halo_instance_words(lg) == orc.instance_words(lg)
        assert lib.halo_accumulator_words(lg) == orc.acc_words(lg)


def test_no_cpu_fallback(hal):
    lib = hal.load()
    if lib.halo_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(hal.HaloError) as e:
        hal._lib.Context(urs_n=64)
    assert "no CPU fallback" in str(e.value)
    with pytest.raises(hal.HaloError):
        hal._lib.Context(np.zeros((4, 8), dtype=np.uint64))
    # null handles are rejected, not dereferenced
    out = np.zeros(12, dtype=np.uint64)
    assert lib.halo_msm(None, 0, 0, None, 1, hal._lib.ptr(out)) == hal._lib.HALO_E_ARG
    assert lib.halo_ipa_finish(None, hal._lib.ptr(out), hal._lib.ptr(out)) == hal._lib.HALO_E_ARG
    # this round's entry points: a null context is an argument error, never a crash; halo_ctx_info of nothing is 0
    assert lib.halo_set_memory_budget(None, 1 << 30) != 0 and lib.halo_set_fold_async(None, 1) == hal._lib.HALO_E_ARG
    assert all(lib.halo_ctx_info(None, what) == 0 for what in range(8))
    # the sharded entry points check what every rank passes alike before any collective could be entered
    z4 = np.zeros(4, dtype=np.uint64)
    proof = np.zeros(lib.halo_proof_words(6), dtype=np.uint64)
    assert lib.halo_pcdl_check_sharded(None, 2, 0, hal._lib.ptr(out), 63, hal._lib.ptr(z4), hal._lib.ptr(z4), hal._lib.ptr(proof), None, None) != 0


def test_product_does_not_import_oracle():
    """The product path -- and the development tools -- may not import, link or call anything under oracle/
    (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do)."""
    for top in ("halo-accumulation_amd", "tools"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            if "_obj" in dirpath:
                continue
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".cpp", ".sh", "Makefile")):
                    text = open(os.path.join(dirpath, f)).read()
                    assert "import orc" not in text and "pallas_model" not in text and "liborc" not in text and "halo_cpu" not in text, f
                    assert top != "tools" or '"oracle"' not in text, f


def test_glv_digits_host(hal):
    """xi == sum_i d_i 2^i (mod r) with d_i in {0, +-1, +-lambda, +-lambda^2}, at most ~131 digits, and a non-zero
    density of about 0.6 (host_math.hpp glv_digits: the expansion the fold kernel walks)."""
    import pallas_model as pm
    lib = hal.load()
    lam = 0x6819a58283e528e511db4d81cf70f5a0fed467d47c033af2aa9d2e050aa0e4f
    assert (lam * lam + lam + 1) % pm.R_ORDER == 0
    unit = [None, 1, lam, lam * lam % pm.R_ORDER, -1, -lam, -(lam * lam) % pm.R_ORDER]
    rng = pm.SplitMix64(4)
    xs = [0, 1, 2, 3, pm.R_ORDER - 1, pm.R_ORDER - 2, lam, lam + 1, pm.R_ORDER - lam, (1 << 254) % pm.R_ORDER, lam * lam % pm.R_ORDER]
    xs += [rng.next_scalar() for _ in range(300)]
    nonzero = total = 0
    for x in xs:
        out = (C.c_uint8 * 144)()
        n = C.c_int()
        assert lib.halo_test_glv_digits(hal._lib.ptr(orc.fr_to_mont(x)), out, C.byref(n)) == 0
        assert 0 <= n.value <= 132 and all(0 <= out[i] <= 6 for i in range(144)) and all(out[i] == 0 for i in range(n.value, 144))
        assert n.value == 0 or out[n.value - 1] != 0
        assert (sum(unit[out[i]] << i for i in range(n.value) if out[i]) - x) % pm.R_ORDER == 0
        if x > 1 << 200:
            nonzero += sum(1 for i in range(n.value) if out[i]); total += n.value
    assert 0.5 < nonzero / total < 0.66



def test_fold_comb_digits_host(hal):
    """the digits the comb-table fold walks (foldtab.hip): s == k1 + k2 lambda (mod r) with k_h = sum_i d_i 64^i, 22 digits
    per half, each in [-32, 32] (the table holds the multiples 1..32)"""
    import pallas_model as pm
    lib = hal.load()
    lam = 0x6819a58283e528e511db4d81cf70f5a0fed467d47c033af2aa9d2e050aa0e4f
    rng = pm.SplitMix64(11)
    xs = [0, 1, 2, 31, 32, 33, 63, 64, pm.R_ORDER - 1, pm.R_ORDER - 32, lam, lam + 1, pm.R_ORDER - lam, (1 << 254) % pm.R_ORDER, lam * lam % pm.R_ORDER]
    xs += [rng.next_scalar() for _ in range(500)]
    for x in xs:
        out = (C.c_int8 * 44)()
        assert lib.halo_test_fold_digits(hal._lib.ptr(orc.fr_to_mont(x)), out) == 0
        assert all(-32 <= out[i] <= 32 for i in range(44))
        k1 = sum(int(out[i]) << (6 * i) for i in range(22))
        k2 = sum(int(out[22 + i]) << (6 * i) for i in range(22))
        assert abs(k1) < 1 << 129 and abs(k2) < 1 << 129
        assert (k1 + k2 * lam - x) % pm.R_ORDER == 0


def test_open_tail_reproduces_a_whole_open(hal):
    """halo_open_tail runs pcdl.rs:195-227 on the host over P gathered (G_i, c_i, z_i): fed the WHOLE state of an 8-point
    open (G = the key, c = the coefficients, z_i = z^i) it must return every L, R, U and c of the oracle's proof."""
    from halo_accumulation_amd.sharded import _FQ_ONE
    n, lg = 8, 3
    gs = orc.urs_affine(2, n)
    pp = orc.make_pp(gs)
    coeffs, s = orc.rng_scalars(0xC0FFEE, n)
    zz, _ = orc.rng_scalars(s, 1)
    z = zz[0]
    Cj = orc.pcdl_commit(pp, coeffs, n - 1)
    proof, _ = orc.pcdl_open(pp, 1, coeffs, Cj, n - 1, z)
    v = orc.poly_eval(coeffs, z)
    v2, xi0, Hp = hal._lib.open_start(Cj, z, np.asarray(v).reshape(1, 4))
    assert v2.tolist() == np.asarray(v).tolist()
    recs = np.zeros((n, 20), dtype=np.uint64)
    recs[:, :8] = np.asarray(gs).reshape(n, 8)
    recs[:, 8:12] = _FQ_ONE
    recs[:, 12:16] = coeffs
    recs[:, 16:20] = orc.powers(z, n)
    Ls, Rs, U, c = hal._lib.open_tail(recs, Hp, xi0)
    for i in range(lg):
        assert Ls[i].tolist() == proof[2 + 12 * i: 14 + 12 * i].tolist()
        assert Rs[i].tolist() == proof[2 + 12 * lg + 12 * i: 14 + 12 * lg + 12 * i].tolist()
    o = 2 + 24 * lg
    assert U.tolist() == proof[o: o + 12].tolist() and c.tolist() == proof[o + 12: o + 16].tolist()
    # one element: no round, U = G_0 and c = c_0
    Ls, Rs, U, c = hal._lib.open_tail(recs[:1], Hp, xi0)
    assert len(Ls) == 0 and U.tolist() == recs[0, :12].tolist() and c.tolist() == coeffs[0].tolist()
    # an infinity among the points (Z = 0) is folded like any other point
    inf = recs.copy(); inf[5, :12] = 0; inf[5, 0:4] = _FQ_ONE; inf[5, 4:8] = _FQ_ONE
    hal._lib.open_tail(inf, Hp, xi0)
    for bad in (3, 0, 128):
        with pytest.raises(hal._lib.HaloError):
            hal._lib.open_tail(np.zeros((bad, 20), dtype=np.uint64) if bad else np.zeros((0, 20), dtype=np.uint64), Hp, xi0)


def test_open_combine_host_against_the_big_int_model(hal, urs4096):
    """halo_open_combine (host only): L = sum L_i + (sum dot_l,i) H', R likewise, xi = rho_0(xi_prev, L, R), xi^-1 -- against
    the Python big-int model, for two different H' in a row (the entry point keeps a window table of the last H' it saw)
    and with an infinity among the partial points."""
    import pallas_model as pm
    P = 3
    jac = np.zeros((8, 12), dtype=np.uint64)
    for i in range(8):
        orc.lib().orc_affine_to_jac(orc.ptr(urs4096[i]), orc.ptr(jac[i]))
    aff = [orc.point_canonical(jac[i]) for i in range(8)]
    rng = pm.SplitMix64(99)
    for Hi in (6, 7, 6):
        dots = [[rng.next_scalar() for _ in range(2)] for _ in range(P)]
        xi_prev = rng.next_scalar()
        parts = np.zeros((P, 32), dtype=np.uint64)
        for r in range(P):
            parts[r, :12] = jac[r]
            parts[r, 12:24] = jac[3 + r]
            parts[r, 24:28] = orc.fr_to_mont(dots[r][0])
            parts[r, 28:32] = orc.fr_to_mont(dots[r][1])
        parts[1, 0:4] = parts[1, 4:8] = jac[0, 8:12]; parts[1, 8:12] = 0  # L_1 = infinity, as the library writes it: (1, 1, 0)
        L, R, xi, xi_inv = hal._lib.open_combine(parts, jac[Hi], orc.fr_to_mont(xi_prev))
        dl = sum(d[0] for d in dots) % pm.R_ORDER
        dr = sum(d[1] for d in dots) % pm.R_ORDER
        Lw = pm.add(pm.add(aff[0], aff[2]), pm.mul(aff[Hi], dl))
        Rw = pm.add(pm.add(pm.add(aff[3], aff[4]), aff[5]), pm.mul(aff[Hi], dr))
        assert orc.point_canonical(L) == Lw and orc.point_canonical(R) == Rw
        x = pm.rho_0(("s", xi_prev), ("p", Lw), ("p", Rw))
        assert orc.fr_from_mont(xi) == x and orc.fr_from_mont(xi_inv) * x % pm.R_ORDER == 1
