import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def kat():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "urs_kat.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def urs4096():
    """First 4096 URS bases from the CPU restatement (affine Montgomery limbs)."""
    import orc
    return orc.urs_affine(2, 4096)


@pytest.fixture(autouse=True)
def _reset_dev_hooks():
    """A fault injector of the development library that a failing test left on must not leak into the next test."""
    yield
    m = sys.modules.get("halo_accumulation_amd._lib")
    lib = getattr(m, "_lib", None) if m is not None else None
    if lib is not None and getattr(lib, "_dev", None) is not None:
        lib._dev.halo_dev_hook(b"reset", 0)
