// Links the C-ABI product library.  FASTLLM_MI355X_LIB_DIR points at the directory that holds
// libfastllm_mi355x.so (in this repository: fastllm_amd/lib, built by `python -c "import __graft_entry__ as g; g.build()"`).
use std::env;
use std::path::PathBuf;

fn main() {
    println!("cargo:rerun-if-env-changed=FASTLLM_MI355X_LIB_DIR");
    println!("cargo:rerun-if-changed=build.rs");
    let dir = env::var("FASTLLM_MI355X_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        // default: the in-tree build output, two levels above this crate
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../fastllm_amd/lib")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=fastllm_mi355x");
    // the library's own dependencies (HIP runtime, RCCL) are resolved through its DT_RUNPATH (/opt/rocm/lib)
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
}
