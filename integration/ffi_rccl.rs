// code/src/ffi_rccl.rs -- optional: the sharded entry points' all-gather over RCCL, native (include/halo_rccl.h,
// libhalo_rccl.so).  Uncompiled source, like ffi.rs (no Rust toolchain in the build image); tests/test_integration_patches.py
// lints it structurally against the header.  A multi-GPU host adds `pub mod ffi_rccl;` next to `pub mod ffi;` and passes
// `RcclGather::callback()` / `.user()` to `ffi::halo_pcdl_open_sharded` / `halo_pcdl_check_sharded`; one rank per GPU.
use std::os::raw::{c_char, c_int, c_void};

pub const HALO_RCCL_ID_BYTES: usize = 128;

#[link(name = "halo_rccl")]
extern "C" {
    pub fn halo_rccl_unique_id(id: *mut u8) -> c_int;
    pub fn halo_rccl_create(id: *const u8, rank: c_int, world: c_int, device: c_int, out: *mut *mut c_void) -> c_int;
    pub fn halo_rccl_wrap(nccl_comm: *mut c_void, hip_stream: *mut c_void, world: c_int, device: c_int, out: *mut *mut c_void) -> c_int;
    pub fn halo_rccl_destroy(g: *mut c_void);
    pub fn halo_allgather_rccl(user: *mut c_void, send: *const u64, words: usize, recv: *mut u64) -> c_int;
    pub fn halo_rccl_calls(g: *const c_void) -> usize;
    pub fn halo_rccl_world(g: *const c_void) -> c_int;
    pub fn halo_rccl_last_error() -> *const c_char;
}

fn last_error() -> String {
    unsafe { std::ffi::CStr::from_ptr(halo_rccl_last_error()).to_string_lossy().into_owned() }
}

/// Rank 0 makes the communicator's id; the launcher hands the 128 bytes to the other ranks (environment, file, its store).
pub fn unique_id() -> [u8; HALO_RCCL_ID_BYTES] {
    let mut id = [0u8; HALO_RCCL_ID_BYTES];
    let rc = unsafe { halo_rccl_unique_id(id.as_mut_ptr()) };
    if rc != 0 { panic!("halo_rccl_unique_id: {}", last_error()); }
    id
}

/// One rank's communicator, stream and staging buffers.  Collective: every rank of `world` creates it with the same id.
pub struct RcclGather { handle: *mut c_void }

impl RcclGather {
    pub fn new(id: &[u8; HALO_RCCL_ID_BYTES], rank: usize, world: usize, device: usize) -> RcclGather {
        let mut handle: *mut c_void = std::ptr::null_mut();
        let rc = unsafe { halo_rccl_create(id.as_ptr(), rank as c_int, world as c_int, device as c_int, &mut handle) };
        if rc != 0 { panic!("halo_rccl_create: {}", last_error()); }
        RcclGather { handle }
    }
    /// what `ffi::halo_pcdl_open_sharded(.., allgather, user, ..)` takes: the C function itself, no Rust frame in the collective path
    pub fn callback(&self) -> crate::ffi::HaloAllgatherFn { Some(halo_allgather_rccl) }
    pub fn user(&self) -> *mut c_void { self.handle }
    pub fn calls(&self) -> usize { unsafe { halo_rccl_calls(self.handle) } }
}

impl Drop for RcclGather {
    fn drop(&mut self) { unsafe { halo_rccl_destroy(self.handle) } }
}
