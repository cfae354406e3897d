"""The host arithmetic of the product (csrc/host_math.hpp) under ASan + UBSan on the CPU build (GPU sanitizers are not
available on the pool): fixed-base table == double-and-add, GLV digits bounded, no UB / out-of-bounds anywhere."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_math_is_sanitizer_clean(tmp_path):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "host_math_sanitize")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-I", os.path.join(ROOT, "halo-accumulation_amd", "csrc"), os.path.join(ROOT, "tests", "native", "host_math_sanitize.cpp"), "-o", exe]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    if b.returncode != 0 and "sanitize" in b.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("ok "), r.stdout + r.stderr[-2000:]


def test_host_worker_is_thread_sanitizer_clean(tmp_path):
    """csrc/internal.hpp HostWorker (hand-over of host jobs between the caller and the helper thread, bounded spin while IPA
    states are alive, shutdown) under TSan.  Host-only: the HIP headers are included for their types, nothing is launched."""
    if shutil.which("g++") is None or not os.path.isdir("/opt/rocm/include"):
        pytest.skip("no g++ / HIP headers")
    exe = str(tmp_path / "host_worker_tsan")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-pthread", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
           "-I", os.path.join(ROOT, "halo-accumulation_amd", "csrc"), os.path.join(ROOT, "tests", "native", "host_worker_tsan.cpp"), "-o", exe]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    if b.returncode != 0 and ("sanitize" in b.stderr or "tsan" in b.stderr):
        pytest.skip("sanitizer runtime not installed")
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("ok ") and "ThreadSanitizer" not in r.stderr, r.stdout + r.stderr[-2000:]
