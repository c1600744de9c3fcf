// build.rs -- links the reference crate against librtx_hip.so (the C ABI of include/rtx.h).
// RTX_HIP_LIB_DIR: directory that holds librtx_hip.so (default: rust-raytracer_amd/ next to the crate).
use std::env;

fn main() {
    let dir = env::var("RTX_HIP_LIB_DIR").unwrap_or_else(|_| "rust-raytracer_amd".to_string());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=rtx_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    println!("cargo:rerun-if-env-changed=RTX_HIP_LIB_DIR");
}
