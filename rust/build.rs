// Links libjpeg_hip.so (and libjpeg_host.so for the BMP reader and the JFIF writer).
//
//   JPEG_HIP_DIR=<dir with the two libraries>   link prebuilt libraries (python jpeg-encoder_amd/build.py makes them)
//   otherwise                                   compile csrc/*.hip for gfx950 with hipcc into OUT_DIR and link that
//
// UNVERIFIED: written without a Rust toolchain at hand.
use std::env;
use std::path::{Path, PathBuf};
use std::process::Command;

const HIP_SOURCES: [&str; 7] = ["front.hip", "coded.hip", "staged.hip", "entropy.hip", "api.hip", "static.hip", "chain.hip"];
const HOST_SOURCES: [&str; 3] = ["host/bmp.cpp", "host/jfif.cpp", "host/stripes.cpp"];
// the flags of jpeg-encoder_amd/build.py: the reference's f32 arithmetic is never fused, so contraction stays off
const HIP_FLAGS: [&str; 8] = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
                              "-fno-slp-vectorize", "-fhip-fp32-correctly-rounded-divide-sqrt"];

fn run(cmd: &mut Command) {
    let status = cmd.status().unwrap_or_else(|e| panic!("cannot start {:?}: {}", cmd, e));
    assert!(status.success(), "{:?} failed", cmd);
}

fn build_from_source(root: &Path, out: &Path) {
    let hipcc = env::var("HIPCC").unwrap_or_else(|_| "/opt/rocm/bin/hipcc".to_string());
    let csrc = root.join("jpeg-encoder_amd").join("csrc");
    let include = root.join("include");
    let mut objs = Vec::new();
    for src in HIP_SOURCES {
        let obj = out.join(src.replace(".hip", ".o"));
        run(Command::new(&hipcc).args(HIP_FLAGS).arg("-I").arg(&include).arg("-I").arg(&csrc).arg("-c")
            .arg(csrc.join(src)).arg("-o").arg(&obj));
        objs.push(obj);
        println!("cargo:rerun-if-changed={}", csrc.join(src).display());
    }
    run(Command::new(&hipcc).args(["--offload-arch=gfx950", "-shared", "-fPIC", "-o"]).arg(out.join("libjpeg_hip.so")).args(&objs));
    let cxx = env::var("CXX").unwrap_or_else(|_| "g++".to_string());
    let mut host = Command::new(cxx);
    host.args(["-O2", "-std=c++17", "-fPIC", "-shared", "-I"]).arg(&include).arg("-I").arg(&csrc).arg("-o")
        .arg(out.join("libjpeg_host.so"));
    for src in HOST_SOURCES {
        host.arg(csrc.join(src));
    }
    run(&mut host);
}

fn main() {
    let dir = match env::var("JPEG_HIP_DIR") {
        Ok(d) => PathBuf::from(d),
        Err(_) => {
            let root = PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("..");
            let out = PathBuf::from(env::var("OUT_DIR").unwrap());
            build_from_source(&root, &out);
            out
        }
    };
    println!("cargo:rerun-if-env-changed=JPEG_HIP_DIR");
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=jpeg_hip");
    println!("cargo:rustc-link-lib=dylib=jpeg_host");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
}
