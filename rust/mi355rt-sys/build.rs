// Links the two libraries `make` leaves in raytracer_amd/lib (MI355RT_LIB_DIR overrides the place).
fn main() {
    let dir = std::env::var("MI355RT_LIB_DIR").unwrap_or_else(|_| {
        format!("{}/../../raytracer_amd/lib", std::env::var("CARGO_MANIFEST_DIR").unwrap())
    });
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=mi355rt");
    println!("cargo:rustc-link-lib=dylib=mi355rt_host");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=MI355RT_LIB_DIR");
}
