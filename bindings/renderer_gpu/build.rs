// Link against librt_hip.so (built by `make -C raytracing_amd/csrc` in the MI355X repository).
fn main() {
    if let Ok(dir) = std::env::var("RT_HIP_LIB_DIR") {
        println!("cargo:rustc-link-search=native={}", dir);
        println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    }
    println!("cargo:rustc-link-lib=dylib=rt_hip");
    println!("cargo:rerun-if-env-changed=RT_HIP_LIB_DIR");
}
