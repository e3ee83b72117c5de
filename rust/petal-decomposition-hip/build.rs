// Links libpetal_hip.so.  PETAL_HIP_LIB_DIR points at the directory holding it (the repository's
// petal-decomposition_amd/ after `python __graft_entry__.py build`).
fn main() {
    if let Ok(dir) = std::env::var("PETAL_HIP_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    }
    println!("cargo:rustc-link-lib=dylib=petal_hip");
    println!("cargo:rerun-if-env-changed=PETAL_HIP_LIB_DIR");
}
