// Links the prebuilt shared library.  RWKV_HIP_LIB_DIR = the directory that holds librwkv_hip.so
// (`python -m ai00_server_amd.build` writes it to <repo>/ai00_server_amd/).
use std::{env, path::PathBuf};

fn main() {
    let dir = env::var("RWKV_HIP_LIB_DIR")
        .map(PathBuf::from)
        .unwrap_or_else(|_| PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../ai00_server_amd"));
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=rwkv_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=RWKV_HIP_LIB_DIR");
}
