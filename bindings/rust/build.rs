// Links libkmx.so (built by `python -m kmers_amd.build` at the repository root; hipcc, gfx950 only).
// KMX_LIB_DIR overrides the directory that holds it (default: <repo>/kmers_amd).
use std::env;
use std::path::PathBuf;

fn main() {
    let manifest = PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap());
    let default_dir = manifest.join("..").join("..").join("kmers_amd");
    let dir = env::var("KMX_LIB_DIR").map(PathBuf::from).unwrap_or(default_dir);
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=kmx");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=KMX_LIB_DIR");
    println!("cargo:rerun-if-changed=../../include/kmx.h");
}
