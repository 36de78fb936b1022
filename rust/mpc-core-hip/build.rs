// Links libcogroth16_hip.so (built by `make -C collaborative-circom_amd/csrc`).  COGROTH16_HIP_LIB_DIR points at the directory that holds
// it; the default is the in-tree location relative to this crate.
use std::{env, path::PathBuf};

fn main() {
    let dir = env::var("COGROTH16_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../collaborative-circom_amd")
    });
    let dir = dir.canonicalize().unwrap_or(dir);
    if !dir.join("libcogroth16_hip.so").exists() {
        panic!("libcogroth16_hip.so not found in {} (set COGROTH16_HIP_LIB_DIR or run `make -C collaborative-circom_amd/csrc`)", dir.display());
    }
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=cogroth16_hip");
    println!("cargo:rustc-link-lib=dylib=cogroth16_host");   // session.rs (the host mirror's prove entry points; `make -C collaborative-circom_amd/host`)
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=COGROTH16_HIP_LIB_DIR");
    println!("cargo:rerun-if-changed=../../include/cogroth16_hip.h");
}
