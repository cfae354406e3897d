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
PATCHES = ["lib_rs.patch", "group_rs.patch", "pcdl_rs.patch", "acc_rs.patch"]

needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "code", "src")) or shutil.which("patch") is None,
                               reason="reference tree or patch(1) not present")


@needs_ref
def test_patches_apply_to_the_pristine_reference(tmp_path):
    shutil.copytree(os.path.join(REF, "code", "src"), tmp_path / "code" / "src")
    for name in PATCHES:
        with open(os.path.join(ROOT, "integration", name)) as f:
            r = subprocess.run(["patch", "-p1", "--dry-run"], stdin=f, cwd=tmp_path, capture_output=True, text=True)
        assert r.returncode == 0, "%s does not apply:\n%s%s" % (name, r.stdout, r.stderr)
    # and for real, all four in sequence: the result must contain the calls and none of the replaced bodies
    for name in PATCHES:
        with open(os.path.join(ROOT, "integration", name)) as f:
            subprocess.run(["patch", "-p1", "-s"], stdin=f, cwd=tmp_path, check=True)
    group = (tmp_path / "code" / "src" / "group.rs").read_text()
    pcdl = (tmp_path / "code" / "src" / "pcdl.rs").read_text()
    lib = (tmp_path / "code" / "src" / "lib.rs").read_text()
    assert "mod ffi;" in lib
    assert "msm_unchecked" not in group and group.count("crate::ffi::") == 4
    assert "ffi::KEY as GS" in pcdl and "ipa.round_lr(&H_prime)" in pcdl and "gs[j + m]" not in pcdl
    # a13: AccumulatedHPolys::get_poly / eval (acc.rs:85-106) go to halo_h_accumulate / halo_h_eval_batch
    acc = (tmp_path / "code" / "src" / "acc.rs").read_text()
    assert "crate::ffi::h_accumulate(h_0, &xis, alphas)" in acc and "crate::ffi::h_eval_batch(&xis, z)" in acc
    assert "self.hs[i].get_poly()" not in acc and "self.hs[i].eval(z)" not in acc and "let mut h = PallasPoly::zero();" not in acc
    # every ffi:: function a patched file calls exists in ffi.rs
    import re
    ffi = open(os.path.join(ROOT, "integration", "ffi.rs")).read()
    defined = set(re.findall(r"pub fn (\w+)\s*[(<]", ffi)) | {"KEY", "Ipa"}
    for text in (group, pcdl, acc):
        for name in re.findall(r"crate::ffi::(\w+)", text) + re.findall(r"ffi::(KEY)", text):
            assert name in defined, name


@needs_ref
def test_patches_are_what_make_patches_writes(tmp_path):
    """The committed patches are the generator's output (no hand edits: that is how a hunk header went wrong before)."""
    before = {n: open(os.path.join(ROOT, "integration", n)).read() for n in PATCHES}
    out = tmp_path / "integration"
    shutil.copytree(os.path.join(ROOT, "integration"), out)
    subprocess.run(["python3", str(out / "make_patches.py")], check=True, capture_output=True)
    for n in PATCHES:
        assert (out / n).read_text() == before[n], n


def _strip_rust(src):
    """comments and string / char literals blanked (same length), so that brackets inside them do not count"""
    import re
    out = re.sub(r"//[^\n]*", lambda m: " " * len(m.group(0)), src)
    out = re.sub(r'"(?:[^"\\\n]|\\.)*"', lambda m: '"' + " " * (len(m.group(0)) - 2) + '"', out)
    out = re.sub(r"'(?:[^'\\\n]|\\.)'", lambda m: "' '".ljust(len(m.group(0))), out)
    return out


def _split_args(arglist):
    """top-level comma split of a parameter list"""
    args, depth, cur = [], 0, ""
    for ch in arglist:
        if ch in "(<[":
            depth += 1
        elif ch in ")>]":
            depth -= 1
        if ch == "," and depth == 0:
            args.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        args.append(cur.strip())
    return args


def _c_class(param):
    """coarse class of a C parameter: ptr / fn / usize / int / u64"""
    p = param.strip()
    if "halo_allgather_fn" in p:
        return "fn"
    if "*" in p or "[" in p:
        return "ptr"
    for key, cls in (("size_t", "usize"), ("uint64_t", "u64"), ("int", "int")):
        if key in p.split():
            return cls
    raise AssertionError("unclassified C parameter: %r" % param)


def _rust_class(param):
    ty = param.split(":", 1)[1].strip()
    if ty.startswith("*"):
        return "ptr"
    return {"usize": "usize", "c_int": "int", "u64": "u64", "HaloAllgatherFn": "fn"}[ty]


def _c_class_rccl(param):
    p = param.strip()
    if "*" in p or "[" in p:
        return "ptr"
    return "usize" if "size_t" in p.split() else "int"


def _header_prototypes(name="halo_accumulation.h"):
    import re
    header = open(os.path.join(ROOT, "include", name)).read()
    header = re.sub(r"/\*.*?\*/", " ", header, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b([A-Za-z_][\w \*]*?)\b(halo_\w+)\s*\(([^;{}]*?)\)\s*;", header):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if "typedef" in ret:
            continue
        params = [] if args in ("", "void") else _split_args(args)
        protos[name] = (ret, params)
    return protos


def test_ffi_rs_is_structurally_sound():
    """integration/ffi.rs is never compiled here (no Rust toolchain), and it has been wrong before: a `#[link]` attribute
    that ended up on a type alias.  This is a structural lint, not a compiler: every `#[link` sits
    directly on an `extern "C" {` block, brackets balance, every item of the extern block is a `pub fn ...;`, and every
    declaration has the arity, the parameter classes (pointer / usize / c_int / u64 / callback) and the return class of the
    prototype in include/halo_accumulation.h."""
    import re
    raw = open(os.path.join(ROOT, "integration", "ffi.rs")).read()
    src = _strip_rust(raw)
    # 1. brackets balance (and never go negative)
    pairs = {")": "(", "]": "[", "}": "{"}
    stack = []
    for i, ch in enumerate(src):
        if ch in "([{":
            stack.append((ch, i))
        elif ch in ")]}":
            assert stack and stack[-1][0] == pairs[ch], "unbalanced %r at offset %d (line %d)" % (ch, i, src.count("\n", 0, i) + 1)
            stack.pop()
    assert not stack, "unclosed %r at line %d" % (stack[-1][0], src.count("\n", 0, stack[-1][1]) + 1)
    # 2. every #[link ...] attribute is followed (attributes / blank lines aside) by `extern "C" {`, and every extern block has one
    links = [m.end() for m in re.finditer(r"#\[link\s*\([^\]]*\)\]", src)]
    assert links, "no #[link] attribute"
    for end in links:
        rest = src[end:].lstrip()
        while rest.startswith("#["):
            rest = rest[rest.index("]") + 1:].lstrip()
        assert re.match(r'extern\s+"\s*"\s*\{', rest), "#[link] does not decorate an extern block: %r" % rest[:60]  # (literals are blanked)
    blocks = list(re.finditer(r'extern\s+"[^"]*"\s*\{', src))
    assert len(blocks) == len(links) == 1
    assert re.search(r'#\[link\s*\(\s*name\s*=\s*"halo_hip"\s*\)\]\s*extern\s+"C"\s*\{', raw), "the extern block lost its #[link(name = \"halo_hip\")]"
    # 3. the extern block: only `pub fn name(args) [-> ret];` items
    start = blocks[0].end()
    depth, i = 1, start
    while depth:
        depth += {"{": 1, "}": -1}.get(src[i], 0)
        i += 1
    body = src[start:i - 1]
    items = [it.strip() for it in body.split(";")]
    assert items[-1] == "", "the last item of the extern block does not end in ';'"
    decls = {}
    for it in items[:-1]:
        m = re.fullmatch(r"pub fn (halo_\w+)\s*\((.*)\)\s*(?:->\s*(.+))?", it, flags=re.S)
        assert m, "not a `pub fn ...;` declaration: %r" % it[:80]
        assert m.group(1) not in decls, "declared twice: " + m.group(1)
        decls[m.group(1)] = (_split_args(m.group(2)), (m.group(3) or "").strip())
    # 4. against the header: arity, parameter classes, return class
    protos = _header_prototypes()
    assert len(protos) > 80
    for name, (params, ret) in decls.items():
        assert name in protos, name + " is not declared in the header"
        c_ret, c_params = protos[name]
        assert len(params) == len(c_params), "%s: %d parameters in ffi.rs, %d in the header" % (name, len(params), len(c_params))
        for k, (rp, cp) in enumerate(zip(params, c_params)):
            assert re.match(r"\w+\s*:", rp), "%s: parameter %d has no name: %r" % (name, k, rp)
            assert _rust_class(rp) == _c_class(cp), "%s: parameter %d is %r in ffi.rs and %r in the header" % (name, k, rp, cp)
        want = "ptr" if "*" in c_ret else "void" if c_ret.split()[-1] == "void" else "int" if c_ret.split()[-1] == "int" else "usize"
        got = "ptr" if ret.startswith("*") else "void" if ret == "" else {"c_int": "int", "usize": "usize"}[ret]
        assert got == want, "%s: returns %r in ffi.rs, %r in the header" % (name, ret, c_ret)
    # 5. every halo_* function the wrappers call is declared in the extern block
    called = set(re.findall(r"\b(halo_\w+)\s*\(", src[i:]))
    assert called <= set(decls), sorted(called - set(decls))
    # 6. what the patches need is there: the a13 binding and the multi-device context
    for need in ("halo_h_accumulate", "halo_h_eval_batch", "halo_ctx_create_multi"):
        assert need in decls and need in called, need
    assert "HALO_DEVICES" in raw and "dyn PrimeField" not in raw


def test_ffi_rccl_rs_matches_its_header():
    """integration/ffi_rccl.rs (the optional RCCL all-gather for a Rust host) against include/halo_rccl.h: brackets
    balance, #[link(name = "halo_rccl")] sits on the one extern block, every declaration has the header's arity, parameter
    classes and return class, and every header symbol is declared."""
    import re
    raw = open(os.path.join(ROOT, "integration", "ffi_rccl.rs")).read()
    src = _strip_rust(raw)
    pairs, stack = {")": "(", "]": "[", "}": "{"}, []
    for i, ch in enumerate(src):
        if ch in "([{":
            stack.append(ch)
        elif ch in ")]}":
            assert stack and stack.pop() == pairs[ch], "unbalanced %r at line %d" % (ch, src.count("\n", 0, i) + 1)
    assert not stack
    assert re.search(r'#\[link\s*\(\s*name\s*=\s*"halo_rccl"\s*\)\]\s*extern\s+"C"\s*\{', raw)
    blocks = list(re.finditer(r'extern\s+"[^"]*"\s*\{', src))
    assert len(blocks) == 1
    start = blocks[0].end()
    depth, i = 1, start
    while depth:
        depth += {"{": 1, "}": -1}.get(src[i], 0)
        i += 1
    items = [it.strip() for it in src[start:i - 1].split(";")]
    assert items[-1] == ""
    protos = _header_prototypes("halo_rccl.h")
    decls = {}
    for it in items[:-1]:
        m = re.fullmatch(r"pub fn (halo_\w+)\s*\((.*)\)\s*(?:->\s*(.+))?", it, flags=re.S)
        assert m, it[:80]
        decls[m.group(1)] = (_split_args(m.group(2)), (m.group(3) or "").strip())
    assert set(decls) == set(protos), (sorted(set(protos) - set(decls)), sorted(set(decls) - set(protos)))
    for name, (params, ret) in decls.items():
        c_ret, c_params = protos[name]
        assert len(params) == len(c_params), name
        for rp, cp in zip(params, c_params):
            ty = rp.split(":", 1)[1].strip()
            got = "ptr" if ty.startswith("*") else {"usize": "usize", "c_int": "int"}[ty]
            assert got == _c_class_rccl(cp), (name, rp, cp)
        want = "ptr" if "*" in c_ret else "void" if c_ret.split()[-1] == "void" else "int" if c_ret.split()[-1] == "int" else "usize"
        got = "ptr" if ret.startswith("*") else "void" if ret == "" else {"c_int": "int", "usize": "usize"}[ret]
        assert got == want, (name, ret, c_ret)
    # the callback it hands out has the type the core shim's sharded entry points take
    assert "crate::ffi::HaloAllgatherFn { Some(halo_allgather_rccl) }" in raw
    assert "pub type HaloAllgatherFn = Option<unsafe extern \"C\" fn(user: *mut c_void, send: *const u64, words: usize, recv: *mut u64) -> c_int>;" in open(os.path.join(ROOT, "integration", "ffi.rs")).read()


def test_shim_declares_only_exported_symbols():
    """every `pub fn halo_*` of ffi.rs is declared in the header (the link step of a cargo build would fail otherwise)"""
    import re
    ffi = open(os.path.join(ROOT, "integration", "ffi.rs")).read()
    header = open(os.path.join(ROOT, "include", "halo_accumulation.h")).read()
    names = set(re.findall(r"pub fn (halo_\w+)\(", ffi))
    assert names and all(re.search(r"\b%s\(" % n, header) for n in names), sorted(n for n in names if not re.search(r"\b%s\(" % n, header))


def test_integration_md_quotes_ffi_rs_verbatim():
    """the declarations INTEGRATION.md section 1 shows are the lines of integration/ffi.rs (no second, drifting copy)"""
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    ffi = open(os.path.join(ROOT, "integration", "ffi.rs")).read()
    quoted = [l for l in md.splitlines() if l.strip().startswith("pub fn halo_")]
    assert len(quoted) >= 20
    for l in quoted:
        assert l.strip() in ffi, l
    assert '#[link(name = "halo_hip")]\nextern "C" {' in md
