// Links libbz2_mi355x.so (the C ABI of include/bz2_mi355x.h).
//   BZ2_MI355X_LIB_DIR  directory holding libbz2_mi355x.so (default: ../ -- the package directory
//                       rust-compression_amd/ where `python -m rust-compression_amd._build` puts it)
use std::env;
use std::path::PathBuf;

fn main() {
    println!("cargo:rerun-if-env-changed=BZ2_MI355X_LIB_DIR");
    if env::var_os("CARGO_FEATURE_MI355X").is_none() {
        return;
    }
    let dir = env::var_os("BZ2_MI355X_LIB_DIR").map(PathBuf::from).unwrap_or_else(|| {
        PathBuf::from(env::var_os("CARGO_MANIFEST_DIR").expect("cargo sets CARGO_MANIFEST_DIR")).join("..")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=bz2_mi355x");
    // the library is found at run time next to where it was linked from
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
}
