// Links libsnapmi.so (built by `python -c 'import __graft_entry__ as g; g.build()'`
// into rust-snappy_amd/).  SNAPMI_LIB_DIR overrides the search path.
fn main() {
    let dir = std::env::var("SNAPMI_LIB_DIR")
        .unwrap_or_else(|_| format!("{}/../rust-snappy_amd", env!("CARGO_MANIFEST_DIR")));
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=snapmi");
    println!("cargo:rerun-if-env-changed=SNAPMI_LIB_DIR");
}
