//! The reference's `JpegImage` (src/jpeg/jpeg_image.rs:8-21) over the C ABI of `libjpeg_hip.so`
//! (include/jpeg_hip.h) and `libjpeg_host.so` (include/jpeg_host.h).
//!
//! `main.rs` of the reference calls five methods in order (src/main.rs:17,35,48,61,66); `JpegImage` below has
//! the same five with the same meaning, each one call into the library.  Errors keep the reference's behaviour:
//! a panic with the reference's message (the library returns the message through `jpg_strerror`).
//!
//! UNVERIFIED: no Rust toolchain exists in the image this repository was built in; the crate has never been
//! compiled.  The C ABI it binds is exercised by the C++ command-line front end and the Python binding.
pub mod hip;
pub mod jpeg_image;

pub use jpeg_image::{DctAlgorithm, JpegImage};
