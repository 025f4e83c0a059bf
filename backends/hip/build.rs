fn main() {
    // librfw_hip.so is built by `make -C rfw-rs_amd/csrc` (hipcc, --offload-arch=gfx950)
    let dir = std::env::var("RFW_HIP_LIB_DIR").expect("set RFW_HIP_LIB_DIR to the directory holding librfw_hip.so");
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=rfw_hip");
}
