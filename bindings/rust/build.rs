// build.rs -- link against libmp3rgain_amd.so (built by `make -C mp3rgain_amd/csrc`, hipcc, gfx950).
//
// MP3RGAIN_AMD_LIB_DIR names the directory that holds the shared library; default: the in-tree build output,
// ../../mp3rgain_amd relative to this crate.  The library has no CPU path: at run time `rg_create` answers
// RG_ERR_NO_DEVICE without a gfx950 device, which `Context::new` turns into an error.
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("MP3RGAIN_AMD_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../mp3rgain_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=mp3rgain_amd");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=MP3RGAIN_AMD_LIB_DIR");
    println!("cargo:rerun-if-changed=../../include/mp3rgain_amd.h");
    println!("cargo:rerun-if-changed=../../include/mp3rgain_amd_node.h");
}
