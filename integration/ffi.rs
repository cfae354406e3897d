//! `code/src/ffi.rs` -- binding of libhalo_hip.so (include/halo_accumulation.h) for rasmus-kirk/halo-accumulation.
//!
//! Drop this file into `code/src/` and apply `lib_rs.patch` (`mod ffi;`), `group_rs.patch`, `pcdl_rs.patch` and `acc_rs.patch`.
//! `build.rs` next to it tells cargo where the library is.  NOT COMPILED in the build image of this repository
//! (no Rust toolchain there): the C side is exercised through the identical C ABI by `integration/harness.c`
//! and by the ctypes binding of the parity tests.
#![allow(non_snake_case, dead_code)]

use std::os::raw::{c_char, c_int, c_void};
use std::sync::OnceLock;

use ark_ff::BigInt;

use crate::consts::{GS, N};
use crate::group::{PallasAffine, PallasPoint, PallasScalar};

/// The commitment key at ONE address.  `consts::GS` is a `const`: every `&GS[a..b]` in the crate borrows a fresh
/// (promoted or stack) copy, so a slice of it cannot be recognised by address.  `pcdl_rs.patch` makes `pcdl.rs` read the
/// key through this static (`use crate::ffi::KEY as GS`); `point_dot_affine` below recognises slices of it and sends
/// everything else -- any other `&[PallasAffine]` a caller of `pedersen::commit` may pass -- through `halo_msm_affine`.
pub static KEY: [PallasAffine; N] = GS;

#[repr(C)] pub struct HaloCtx { _p: [u8; 0] }
#[repr(C)] pub struct HaloIpa { _p: [u8; 0] }

pub const HALO_OK: c_int = 0;
pub const HALO_E_ASSERT: c_int = -1; // -> panic!   (reference: assert!)
pub const HALO_E_REJECT: c_int = -2; // -> bail!    (reference: ensure!)
pub const HALO_E_ARG: c_int = -3;
pub const HALO_E_DEVICE: c_int = -4;

/// every rank contributes `words` u64, `recv` receives P x words in rank order (e.g. a wrapper over `MPI_Allgather` or
/// `ncclAllGather` + stream sync); 0 = success
pub type HaloAllgatherFn = Option<unsafe extern "C" fn(user: *mut c_void, send: *const u64, words: usize, recv: *mut u64) -> c_int>;

#[link(name = "halo_hip")]
extern "C" {
    pub fn halo_last_error() -> *const c_char;
    pub fn halo_ctx_create(device: c_int, bases_affine: *const u64, n: usize, out: *mut *mut HaloCtx) -> c_int;
    pub fn halo_ctx_create_multi(devices: *const c_int, n_dev: c_int, bases_affine: *const u64, n: usize, out: *mut *mut HaloCtx) -> c_int;
    pub fn halo_ctx_destroy(ctx: *mut HaloCtx);
    pub fn halo_msm(ctx: *mut HaloCtx, off: usize, n: usize, scalars: *const u64, mont: c_int, out_jac: *mut u64) -> c_int;
    pub fn halo_msm_points(ctx: *mut HaloCtx, pts_jac: *const u64, scalars: *const u64, m: usize, out_jac: *mut u64) -> c_int;
    pub fn halo_msm_affine(ctx: *mut HaloCtx, bases_affine: *const u64, scalars: *const u64, m: usize, mont: c_int, out_jac: *mut u64) -> c_int;
    pub fn halo_scalar_dot(ctx: *mut HaloCtx, xs: *const u64, ys: *const u64, m: usize, out: *mut u64) -> c_int;
    pub fn halo_powers(ctx: *mut HaloCtx, z: *const u64, n: usize, out: *mut u64) -> c_int;
    pub fn halo_poly_eval(ctx: *mut HaloCtx, coeffs: *const u64, len: usize, z: *const u64, out: *mut u64) -> c_int;
    pub fn halo_h_coeffs(ctx: *mut HaloCtx, xis: *const u64, lg_n: usize, out: *mut u64) -> c_int;
    pub fn halo_h_commit(ctx: *mut HaloCtx, xis: *const u64, lg_n: usize, out_jac: *mut u64) -> c_int;
    pub fn halo_h_eval_batch(ctx: *mut HaloCtx, xis: *const u64, m: usize, lg_n: usize, z: *const u64, out: *mut u64) -> c_int;
    pub fn halo_h_accumulate(ctx: *mut HaloCtx, h0: *const u64, xis: *const u64, alphas: *const u64, m: usize, lg_n: usize, out: *mut u64) -> c_int;
    pub fn halo_ipa_begin(ctx: *mut HaloCtx, n: usize, coeffs: *const u64, len: usize, z: *const u64, out: *mut *mut HaloIpa) -> c_int;
    pub fn halo_ipa_round_lr(st: *mut HaloIpa, h_prime: *const u64, l: *mut u64, r: *mut u64) -> c_int;
    pub fn halo_ipa_round_fold(st: *mut HaloIpa, xi: *const u64, xi_inv: *const u64) -> c_int;
    pub fn halo_ipa_finish(st: *mut HaloIpa, u: *mut u64, c: *mut u64) -> c_int;
    pub fn halo_ipa_destroy(st: *mut HaloIpa);
    // one process per GPU (MPI / RCCL ranks): the key placed cyclically, the collectives supplied by the caller
    pub fn halo_ctx_create_urs_strided(device: c_int, first_index: u64, stride: u64, n: usize, out: *mut *mut HaloCtx) -> c_int;
    pub fn halo_pcdl_open_sharded(ctx: *mut HaloCtx, stride: u64, offset: u64, rng_state: *mut u64, coeffs_local: *const u64, len_local: usize,
                                  deg: usize, c: *const u64, d: usize, z: *const u64, w: *const u64, allgather: HaloAllgatherFn,
                                  user: *mut c_void, proof_out: *mut u64, v_out: *mut u64) -> c_int;
    pub fn halo_pcdl_check_sharded(ctx: *mut HaloCtx, stride: u64, offset: u64, c: *const u64, d: usize, z: *const u64, v: *const u64,
                                   proof: *const u64, allgather: HaloAllgatherFn, user: *mut c_void) -> c_int;
}

// ---- limbs: exactly what `main.rs:47-53` prints (`x.0 .0` is the `[u64; 4]` Montgomery representation).
// Never transmute `&[Affine]` / `&[Projective]`: Rust struct layout is unspecified and `Affine` carries `infinity: bool`.
pub fn fr_limbs(xs: &[PallasScalar]) -> Vec<u64> { xs.iter().flat_map(|x| x.0 .0).collect() }
pub fn aff_limbs(gs: &[PallasAffine]) -> Vec<u64> {
    gs.iter()
        .flat_map(|g| {
            if g.infinity { [0u64; 8] } else {
                let (x, y) = (g.x.0 .0, g.y.0 .0);
                [x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]]
            }
        })
        .collect()
}
pub fn jac_limbs(ps: &[PallasPoint]) -> Vec<u64> {
    ps.iter()
        .flat_map(|p| {
            let (x, y, z) = (p.x.0 .0, p.y.0 .0, p.z.0 .0);
            [x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3], z[0], z[1], z[2], z[3]]
        })
        .collect()
}
/// library output is (x, y, 1) or (1, 1, 0): the constructor `consts.rs:13-21` uses
pub fn point_from(w: [u64; 12]) -> PallasPoint {
    let f = |i: usize| ark_pallas::Fq::new_unchecked(BigInt::new([w[i], w[i + 1], w[i + 2], w[i + 3]]));
    PallasPoint::new_unchecked(f(0), f(4), f(8))
}
pub fn scalar_from(w: [u64; 4]) -> PallasScalar { PallasScalar::new_unchecked(BigInt::new(w)) }

pub fn last_error() -> String {
    unsafe { std::ffi::CStr::from_ptr(halo_last_error()) }.to_string_lossy().into_owned()
}
/// prover side: the reference `assert!`s -> panic on any failure
pub fn ck(rc: c_int) { if rc != HALO_OK { panic!("{}", last_error()); } }
/// verifier side: `ensure!` -> Err, everything else is a library or device failure
pub fn ensure_ok(rc: c_int) -> anyhow::Result<()> {
    match rc {
        HALO_OK => Ok(()),
        HALO_E_REJECT => anyhow::bail!(last_error()),
        _ => panic!("{}", last_error()),
    }
}

/// One context per process for the constant key (`consts.rs:68` is a process-global constant).
/// `HALO_DEVICES=0,1,2,3` (comma-separated HIP device ids, in the environment of the process): ONE process drives several
/// GPUs of the node through a multi-device context (`halo_ctx_create_multi`: a full context on the first device plus one
/// index-block shard per listed device) -- the MSMs behind `point_dot_affine` and the commits inside `pcdl` fan out, nothing
/// else in this file changes.  Unset, or one id: a plain context on that device (default 0).
pub fn ctx() -> *mut HaloCtx {
    static CTX: OnceLock<usize> = OnceLock::new();
    *CTX.get_or_init(|| {
        let mut c = std::ptr::null_mut();
        let limbs = aff_limbs(&KEY);
        let devices: Vec<c_int> = std::env::var("HALO_DEVICES")
            .map(|s| s.split(',').filter_map(|t| t.trim().parse().ok()).collect())
            .unwrap_or_default();
        if devices.len() > 1 {
            ck(unsafe { halo_ctx_create_multi(devices.as_ptr(), devices.len() as c_int, limbs.as_ptr(), KEY.len(), &mut c) });
        } else {
            ck(unsafe { halo_ctx_create(devices.first().copied().unwrap_or(0), limbs.as_ptr(), KEY.len(), &mut c) });
        }
        c as usize
    }) as *mut HaloCtx
}

// ---- the group.rs function set (group.rs:13-37) ------------------------------------------------------------------
/// `point_dot_affine` (group.rs:24-26).  A slice that lies inside the static `KEY` IS that stretch of the key (the static is
/// immutable), so it is named by (offset, length) and runs over the resident bases -- fixed-base tables and all.  Any
/// other slice is sent along with the call.  Both are sound for every `Gs`; only the speed differs.
pub fn point_dot_affine(xs: &[PallasScalar], gs: &[PallasAffine]) -> PallasPoint {
    let n = gs.len().min(xs.len()); // msm_unchecked zips to the shorter input
    let sz = std::mem::size_of::<PallasAffine>();
    let (lo, hi) = (KEY.as_ptr() as usize, KEY.as_ptr() as usize + KEY.len() * sz);
    let p = gs.as_ptr() as usize;
    let mut out = [0u64; 12];
    if p >= lo && p + gs.len() * sz <= hi && (p - lo) % sz == 0 {
        ck(unsafe { halo_msm(ctx(), (p - lo) / sz, n, fr_limbs(&xs[..n]).as_ptr(), 1, out.as_mut_ptr()) });
    } else {
        ck(unsafe { halo_msm_affine(ctx(), aff_limbs(&gs[..n]).as_ptr(), fr_limbs(&xs[..n]).as_ptr(), n, 1, out.as_mut_ptr()) });
    }
    point_from(out)
}
pub fn point_dot(xs: &[PallasScalar], gs: &[PallasPoint]) -> PallasPoint {
    let m = xs.len().min(gs.len());
    let mut out = [0u64; 12];
    ck(unsafe { halo_msm_points(ctx(), jac_limbs(&gs[..m]).as_ptr(), fr_limbs(&xs[..m]).as_ptr(), m, out.as_mut_ptr()) });
    point_from(out)
}
pub fn scalar_dot(xs: &[PallasScalar], ys: &[PallasScalar]) -> PallasScalar {
    let m = xs.len().min(ys.len());
    let mut out = [0u64; 4];
    ck(unsafe { halo_scalar_dot(ctx(), fr_limbs(&xs[..m]).as_ptr(), fr_limbs(&ys[..m]).as_ptr(), m, out.as_mut_ptr()) });
    scalar_from(out)
}
pub fn construct_powers(z: &PallasScalar, n: usize) -> Vec<PallasScalar> {
    let mut out = vec![0u64; 4 * n];
    ck(unsafe { halo_powers(ctx(), z.0 .0.as_ptr(), n, out.as_mut_ptr()) });
    out.chunks_exact(4).map(|w| scalar_from([w[0], w[1], w[2], w[3]])).collect()
}
pub fn poly_eval(coeffs: &[PallasScalar], z: &PallasScalar) -> PallasScalar {
    let mut out = [0u64; 4];
    ck(unsafe { halo_poly_eval(ctx(), fr_limbs(coeffs).as_ptr(), coeffs.len(), z.0 .0.as_ptr(), out.as_mut_ptr()) });
    scalar_from(out)
}
/// `pedersen::commit(None, &GS[0..d+1], &h.get_poly().coeffs)` of pcdl.rs:338, fused on the device
pub fn h_commit(xis: &[PallasScalar]) -> PallasPoint {
    let mut out = [0u64; 12];
    ck(unsafe { halo_h_commit(ctx(), fr_limbs(xis).as_ptr(), xis.len() - 1, out.as_mut_ptr()) });
    point_from(out)
}

// ---- AccumulatedHPolys (acc.rs:69-107) ---------------------------------------------------------------------------
/// the challenges of m polynomials h_i, (lg n + 1) scalars each, back to back
fn xis_limbs(xis: &[&[PallasScalar]]) -> (Vec<u64>, usize) {
    let per = xis.first().map_or(1, |x| x.len());
    assert!(per >= 1 && xis.iter().all(|x| x.len() == per), "the accumulated h polynomials have one degree bound (acc.rs:158-166)");
    (xis.iter().flat_map(|x| fr_limbs(x)).collect(), per - 1)
}
/// `AccumulatedHPolys::get_poly` (acc.rs:85-94): the coefficients of h_0 + sum_i alphas[i] h_i(X), every h_i expanded from
/// its challenges on the device in O(n) (the reference multiplies lg n dense polynomials per h_i).  `h0` = the coefficients
/// of the linear polynomial h_0 (0, 1 or 2 of them: DensePolynomial drops leading zeros); `alphas[i]` belongs to `xis[i]`.
pub fn h_accumulate(h0: &[PallasScalar], xis: &[&[PallasScalar]], alphas: &[PallasScalar]) -> Vec<PallasScalar> {
    assert!(h0.len() <= 2 && alphas.len() == xis.len() && !xis.is_empty());
    let (xl, lg_n) = xis_limbs(xis);
    let mut h0l = fr_limbs(h0);
    h0l.resize(8, 0); // zero in Montgomery form is all-zero limbs
    let mut out = vec![0u64; 4 << lg_n];
    ck(unsafe { halo_h_accumulate(ctx(), h0l.as_ptr(), xl.as_ptr(), fr_limbs(alphas).as_ptr(), xis.len(), lg_n, out.as_mut_ptr()) });
    out.chunks_exact(4).map(|w| scalar_from([w[0], w[1], w[2], w[3]])).collect()
}
/// `HPoly::eval` (pcdl.rs:79-91) of m polynomials at one point, as `AccumulatedHPolys::eval` needs them (acc.rs:102-104)
pub fn h_eval_batch(xis: &[&[PallasScalar]], z: &PallasScalar) -> Vec<PallasScalar> {
    if xis.is_empty() { return Vec::new(); }
    let (xl, lg_n) = xis_limbs(xis);
    let mut out = vec![0u64; 4 * xis.len()];
    ck(unsafe { halo_h_eval_batch(ctx(), xl.as_ptr(), xis.len(), lg_n, z.0 .0.as_ptr(), out.as_mut_ptr()) });
    out.chunks_exact(4).map(|w| scalar_from([w[0], w[1], w[2], w[3]])).collect()
}

// ---- the halving loop of pcdl::open (pcdl.rs:183-231) --------------------------------------------------------------
pub struct Ipa(*mut HaloIpa);
impl Ipa {
    pub fn begin(n: usize, coeffs: &[PallasScalar], z: &PallasScalar) -> Self {
        let mut st = std::ptr::null_mut();
        ck(unsafe { halo_ipa_begin(ctx(), n, fr_limbs(coeffs).as_ptr(), coeffs.len(), z.0 .0.as_ptr(), &mut st) });
        Ipa(st)
    }
    /// L, R of pcdl.rs:203-208 (with the H' terms)
    pub fn round_lr(&mut self, h_prime: &PallasPoint) -> (PallasPoint, PallasPoint) {
        let hp = jac_limbs(&[*h_prime]);
        let (mut l, mut r) = ([0u64; 12], [0u64; 12]);
        ck(unsafe { halo_ipa_round_lr(self.0, hp.as_ptr(), l.as_mut_ptr(), r.as_mut_ptr()) });
        (point_from(l), point_from(r))
    }
    /// the folds of pcdl.rs:216-224 with the challenge the caller hashed from (xi_prev, L, R)
    pub fn round_fold(&mut self, xi: &PallasScalar, xi_inv: &PallasScalar) {
        ck(unsafe { halo_ipa_round_fold(self.0, xi.0 .0.as_ptr(), xi_inv.0 .0.as_ptr()) });
    }
    /// U = G[0], c = c[0] (pcdl.rs:230-231)
    pub fn finish(self) -> (PallasPoint, PallasScalar) {
        let (mut u, mut c) = ([0u64; 12], [0u64; 4]);
        ck(unsafe { halo_ipa_finish(self.0, u.as_mut_ptr(), c.as_mut_ptr()) });
        (point_from(u), scalar_from(c))
    }
}
impl Drop for Ipa { fn drop(&mut self) { unsafe { halo_ipa_destroy(self.0) } } }

