// build.rs — links the reference crate against the MI355X renderer's C ABI (include/rtc_hip.h).
//
// RTC_HIP_LIB_DIR names the directory that holds librtc_hip.so (ray-tracer-challenge-rust_amd/csrc after
// `python ray-tracer-challenge-rust_amd/build.py`).  The library depends on the ROCm HIP runtime only; RCCL is
// loaded lazily by rtc_multi_render and is not a link-time dependency.
//
// Not compiled in the build image (no cargo / rustc there); tests/test_rust_binding.py checks the struct layouts
// and the extern block of src/gpu.rs against include/rtc_hip.h.
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("RTC_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../ray-tracer-challenge-rust_amd/csrc")
    });
    println!("cargo:rerun-if-env-changed=RTC_HIP_LIB_DIR");
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=rtc_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
}
