// code/build.rs -- point cargo at libhalo_hip.so (built by `python __graft_entry__.py` in the halo-accumulation_amd tree)
fn main() {
    let dir = std::env::var("HALO_HIP_DIR").expect("set HALO_HIP_DIR to the directory that holds libhalo_hip.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=halo_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=HALO_HIP_DIR");
}
