"""integration/harness.c: the patched pcdl::open's call sequence driven from plain C through the C ABI."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    import halo_accumulation_amd as h
    h.build()
    exe = str(tmp_path / "harness")
    lib = os.path.join(ROOT, "halo-accumulation_amd")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "integration", "harness.c"),
                           "-o", exe, "-L" + lib, "-lhalo_hip", "-Wl,-rpath," + lib])
    return exe


def test_harness_compiles_against_the_header(tmp_path):
    """CPU: the header is plain C and the library exports what the shim binds"""
    exe = _build(tmp_path)
    rust = open(os.path.join(ROOT, "integration", "ffi.rs")).read()
    hdr = open(os.path.join(ROOT, "include", "halo_accumulation.h")).read()
    import re
    for sym in re.findall(r"pub fn (halo_\w+)\(", rust):
        assert re.search(r"\b%s\(" % sym, hdr), sym
    assert os.path.exists(exe)


@pytest.mark.gpu
@pytest.mark.parametrize("lg", [3, 12, 17])
def test_harness_runs(tmp_path, lg):
    exe = _build(tmp_path)
    out = subprocess.run([exe, str(lg)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "shim loop == halo_pcdl_open: yes" in out.stdout and "wire round trip: yes" in out.stdout
    assert "acc.rs get_poly / eval through h_accumulate / h_eval_batch: yes" in out.stdout
