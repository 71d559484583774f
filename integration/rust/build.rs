// Additions to dbgphmm's build.rs (the existing one embeds the git hash, build.rs:5-12): link the HIP library.
// PHMM_AMD_LIB_DIR = the directory holding libphmm_amd.so (built by `make -C dbgphmm_amd/csrc`).
fn main() {
    if let Ok(dir) = std::env::var("PHMM_AMD_LIB_DIR") {
        println!("cargo:rustc-link-search=native={}", dir);
        println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    }
    println!("cargo:rustc-link-lib=dylib=phmm_amd");
    println!("cargo:rerun-if-env-changed=PHMM_AMD_LIB_DIR");
}
