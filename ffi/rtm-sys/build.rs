// Finds libraytracer_rs_amd.so: RTM_LIB_DIR, or the in-tree build next to this crate (make -C raytracer-rs_amd/csrc).
use std::{env, path::PathBuf};

fn main() {
    let dir = env::var("RTM_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../raytracer-rs_amd/lib")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=raytracer_rs_amd");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=RTM_LIB_DIR");
}
