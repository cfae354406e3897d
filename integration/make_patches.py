#!/usr/bin/env python3
"""Regenerate integration/*.patch from the pristine reference tree (needs /root/reference; run in the build container).

The patches are real `diff -u` output of (pristine file, edited copy), so `patch -p1 --dry-run` on a pristine tree is
guaranteed to succeed -- tests/test_integration_patches.py checks exactly that whenever the reference is present.
Each edit below is (anchor text that must occur exactly once in the reference file, replacement); the replacement text
is this repository's own (the calls into ffi.rs), the anchors are the lines it replaces.
"""
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("HALO_REFERENCE", "/root/reference")


def lines(path, a, b):
    """lines a..b (1-based, inclusive) of a file as one string"""
    with open(path) as f:
        ls = f.readlines()
    return "".join(ls[a - 1:b])


def edit_group(src):
    s = open(src).read()
    edits = [
        (lines(src, 14, 14), "    crate::ffi::scalar_dot(xs, ys)\n"),
        (lines(src, 19, 20), "    crate::ffi::point_dot(xs, &Gs)\n"),
        (lines(src, 25, 25),
         "    // a slice of the key `ffi::KEY` (pcdl.rs:109) is named by (offset, length); any other generators\n"
         "    // (pedersen::commit is public API over any &[PallasAffine]) are uploaded for the call: halo_msm_affine\n"
         "    crate::ffi::point_dot_affine(xs, Gs)\n"),
        (lines(src, 30, 36), "    crate::ffi::construct_powers(z, n)\n"),
    ]
    return apply(s, edits, src)


def edit_pcdl(src):
    s = open(src).read()
    edits = [
        # the key is read through the static copy in ffi.rs, so that `&GS[0..n]` has ONE address the shim can recognise
        # (a `const` array is re-materialised at every use site)
        (lines(src, 15, 18),
         "    consts::{D, H, S},\n"
         "    ffi::KEY as GS,\n"
         "    group::{rho_0, PallasPoint, PallasPoly, PallasScalar},\n"),
        (lines(src, 135, 135), "    let v = crate::ffi::poly_eval(&p.coeffs, z);\n"),
        (lines(src, 183, 187),
         "    // G = GS[0..n), c = coefficients zero-padded to n, z-powers: device resident (state of the loop below)\n"
         "    let mut ipa = crate::ffi::Ipa::begin(n, &p_prime.coeffs, z);\n"
         "\n"),
        (lines(src, 191, 192), ""),
        (lines(src, 199, 209),
         "        let (L, R) = ipa.round_lr(&H_prime); // two MSMs and two dot products on the device, H' terms included\n"
         "        Ls.push(L);\n"
         "        Rs.push(R);\n"),
        (lines(src, 216, 226),
         "        // 4./5. the folds of G, c and z with the challenge just hashed (xi depends on L and R: order is mandatory)\n"
         "        ipa.round_fold(&xi_next, &xi_next_inv);\n"),
        (lines(src, 230, 231), "    let (U, c) = ipa.finish();\n"),
        (lines(src, 338, 338),
         "    let comm = crate::ffi::h_commit(&h.xis); // h.get_poly() expanded on the device, fused with the MSM\n"),
    ]
    return apply(s, edits, src)


def edit_acc(src):
    """AccumulatedHPolys::get_poly / eval (acc.rs:85-106): m x lg n dense-polynomial products on one core -> one O(n)
    expansion per h_i on the device (halo_h_accumulate), and the m evaluations in one launch (halo_h_eval_batch)"""
    s = open(src).read()
    edits = [
        (lines(src, 86, 93),
         "        if self.hs.is_empty() {\n"
         "            return self.h_0.clone().unwrap_or_else(PallasPoly::zero);\n"
         "        }\n"
         "        // h_0 + sum_i alpha^(i+1) h_i(X): every h_i expanded from its challenges on the device, O(n) each\n"
         "        let h_0: &[PallasScalar] = match &self.h_0 {\n"
         "            Some(h_0) => h_0.coeffs.as_slice(),\n"
         "            None => &[],\n"
         "        };\n"
         "        let xis: Vec<&[PallasScalar]> = self.hs.iter().map(|h| h.xis.as_slice()).collect();\n"
         "        let alphas = &self.alphas[1..=self.hs.len()];\n"
         "        PallasPoly::from_coefficients_vec(crate::ffi::h_accumulate(h_0, &xis, alphas))\n"),
        (lines(src, 102, 104),
         "        let xis: Vec<&[PallasScalar]> = self.hs.iter().map(|h| h.xis.as_slice()).collect();\n"
         "        for (i, h_i_z) in crate::ffi::h_eval_batch(&xis, z).iter().enumerate() {\n"
         "            v += *h_i_z * self.alphas[i + 1];\n"
         "        }\n"),
    ]
    return apply(s, edits, src)


def edit_lib(src):
    s = open(src).read()
    return apply(s, [(lines(src, 2, 3), "mod consts;\nmod ffi;\npub mod group;\n")], src)


def apply(s, edits, src):
    for old, new in edits:
        if s.count(old) != 1:
            sys.exit("anchor not unique in %s:\n%s" % (src, old))
        s = s.replace(old, new, 1)
    return s


def make(rel, editor, out_name):
    src = os.path.join(REF, rel)
    with tempfile.TemporaryDirectory() as tmp:
        for side in ("a", "b"):
            os.makedirs(os.path.join(tmp, side, os.path.dirname(rel)))
        shutil.copy(src, os.path.join(tmp, "a", rel))
        with open(os.path.join(tmp, "b", rel), "w") as f:
            f.write(editor(src))
        r = subprocess.run(["diff", "-u", "--label", "a/" + rel, "--label", "b/" + rel, os.path.join("a", rel), os.path.join("b", rel)],
                           cwd=tmp, capture_output=True, text=True)
        assert r.returncode == 1, r.stderr  # 1 = differences found
    with open(os.path.join(HERE, out_name), "w") as f:
        f.write(r.stdout)


if __name__ == "__main__":
    make("code/src/group.rs", edit_group, "group_rs.patch")
    make("code/src/pcdl.rs", edit_pcdl, "pcdl_rs.patch")
    make("code/src/acc.rs", edit_acc, "acc_rs.patch")
    make("code/src/lib.rs", edit_lib, "lib_rs.patch")
    print("wrote group_rs.patch, pcdl_rs.patch, acc_rs.patch, lib_rs.patch")
