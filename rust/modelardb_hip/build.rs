// Links libmdb_hip.so. MDB_HIP_LIB_DIR points at the directory that holds it
// (modelardb-rs_amd/csrc of the HIP repository after `make -C modelardb-rs_amd/csrc`).
fn main() {
    println!("cargo:rerun-if-env-changed=MDB_HIP_LIB_DIR");
    if let Ok(directory) = std::env::var("MDB_HIP_LIB_DIR") {
        println!("cargo:rustc-link-search=native={directory}");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{directory}");
    }
    println!("cargo:rustc-link-lib=dylib=mdb_hip");
}
