"""The reference-side binding a maintainer applies (integration/*.patch) must apply to the pristine reference tree.

Runs wherever /root/reference exists (the build container); skipped on the GPU box, which never has it.  Nothing of the
reference is copied into the repository: the tree is copied to a temporary directory for the dry run and removed.
"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("HALO_REFERENCE", "/root/reference")
PATCHES = ["lib_rs.patch", "group_rs.patch", "pcdl_rs.patch"]

needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "code", "src")) or shutil.which("patch") is None,
                               reason="reference tree or patch(1) not present")


@needs_ref
def test_patches_apply_to_the_pristine_reference(tmp_path):
    shutil.copytree(os.path.join(REF, "code", "src"), tmp_path / "code" / "src")
    for name in PATCHES:
        with open(os.path.join(ROOT, "integration", name)) as f:
            r = subprocess.run(["patch", "-p1", "--dry-run"], stdin=f, cwd=tmp_path, capture_output=True, text=True)
        assert r.returncode == 0, "%s does not apply:\n%s%s" % (name, r.stdout, r.stderr)
    # and for real, all three in sequence: the result must contain the calls and none of the replaced bodies
    for name in PATCHES:
        with open(os.path.join(ROOT, "integration", name)) as f:
            subprocess.run(["patch", "-p1", "-s"], stdin=f, cwd=tmp_path, check=True)
    group = (tmp_path / "code" / "src" / "group.rs").read_text()
    pcdl = (tmp_path / "code" / "src" / "pcdl.rs").read_text()
    lib = (tmp_path / "code" / "src" / "lib.rs").read_text()
    assert "mod ffi;" in lib
    assert "msm_unchecked" not in group and group.count("crate::ffi::") == 4
    assert "ffi::KEY as GS" in pcdl and "ipa.round_lr(&H_prime)" in pcdl and "gs[j + m]" not in pcdl


@needs_ref
def test_patches_are_what_make_patches_writes(tmp_path):
    """The committed patches are the generator's output (no hand edits: that is how a hunk header went wrong before)."""
    before = {n: open(os.path.join(ROOT, "integration", n)).read() for n in PATCHES}
    out = tmp_path / "integration"
    shutil.copytree(os.path.join(ROOT, "integration"), out)
    subprocess.run(["python3", str(out / "make_patches.py")], check=True, capture_output=True)
    for n in PATCHES:
        assert (out / n).read_text() == before[n], n


def test_shim_declares_only_exported_symbols():
    """every `pub fn halo_*` of ffi.rs is declared in the header (the link step of a cargo build would fail otherwise)"""
    import re
    ffi = open(os.path.join(ROOT, "integration", "ffi.rs")).read()
    header = open(os.path.join(ROOT, "include", "halo_accumulation.h")).read()
    names = set(re.findall(r"pub fn (halo_\w+)\(", ffi))
    assert names and all(re.search(r"\b%s\(" % n, header) for n in names), sorted(n for n in names if not re.search(r"\b%s\(" % n, header))
