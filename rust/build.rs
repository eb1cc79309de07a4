fn main() {
    // libbroadphase_amd.so is built by `make -C broadphase_rs_amd/csrc` (hipcc, gfx950)
    let dir = std::env::var("BROADPHASE_AMD_LIB_DIR").unwrap_or_else(|_| "../broadphase_rs_amd/lib".into());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=broadphase_amd");
}
