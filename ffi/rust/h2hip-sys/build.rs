// Links libh2hip.so.  H2HIP_LIB_DIR points at <repo>/halo2-lib_amd/csrc (where __graft_entry__.build() leaves it).
fn main() {
    let dir = std::env::var("H2HIP_LIB_DIR").unwrap_or_else(|_| "../../../halo2-lib_amd/csrc".to_string());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=h2hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=H2HIP_LIB_DIR");
}
