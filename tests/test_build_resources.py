"""The build gate of csrc/check_resources.py, seen from the test suite: after build() every kernel of
libhalo_hip.so reports zero scratch bytes and no dynamic stack (graph replay is only safe without scratch)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_kernel_uses_scratch():
    import halo_accumulation_amd as h
    h.build()
    path = os.path.join(ROOT, "halo-accumulation_amd", "csrc", "_obj", "kernel_resources.json")
    assert os.path.exists(path), "the Makefile writes this file before it links the library"
    res = json.load(open(path))
    names = " ".join(res)
    for k in ("k_msm_accumulate", "k_msm_reduce1", "k_fold_points", "k_msm_fine_sort"):
        assert k in names
    for name, r in res.items():
        assert r["ScratchSize [bytes/lane]"] == "0" and r["Dynamic Stack"] == "False", name
        assert int(r["VGPRs"]) <= 256, name
