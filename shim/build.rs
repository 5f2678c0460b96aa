fn main() {
    let dir = std::env::var("EXON_HIP_LIB_DIR").unwrap_or_else(|_| "../exon_amd/lib".into());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=exon_hip");
}
