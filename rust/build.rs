// Links the in-tree libaether_hip.so (built by `make -C aether_primitives_amd/csrc`).
fn main() {
    let dir = std::env::var("AETHER_HIP_LIB_DIR").unwrap_or_else(|_| "../aether_primitives_amd/lib".into());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=aether_hip");
    println!("cargo:rerun-if-env-changed=AETHER_HIP_LIB_DIR");
}
