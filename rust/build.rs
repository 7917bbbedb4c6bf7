// Links libblaze_hip.so.  BLAZE_HIP_LIB_DIR overrides the default (../blaze_amd/lib, where
// `make -C blaze_amd/csrc` leaves it); the HIP runtime comes in through the library's own DT_NEEDED.
use std::{env, path::PathBuf};

fn main() {
    let dir = env::var("BLAZE_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("..").join("blaze_amd").join("lib")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=blaze_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=BLAZE_HIP_LIB_DIR");
}
