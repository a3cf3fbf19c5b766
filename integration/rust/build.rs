// Links libmilzma.so.  MILZMA_LIB_DIR = the directory that holds it (lzma_rs_amd/ in this repository).
fn main() {
    if let Ok(dir) = std::env::var("MILZMA_LIB_DIR") {
        println!("cargo:rustc-link-search=native={}", dir);
        println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    }
    println!("cargo:rustc-link-lib=dylib=milzma");
    println!("cargo:rerun-if-env-changed=MILZMA_LIB_DIR");
}
