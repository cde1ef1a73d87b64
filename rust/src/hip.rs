//! `extern "C"` declarations of the entry points the five `JpegImage` methods need, and of the fused encode.
use std::os::raw::{c_char, c_int};

#[repr(C)]
pub struct JpgCtx {
    _private: [u8; 0],
}

/// `jph_bmp` of include/jpeg_host.h
#[repr(C)]
pub struct JphBmp {
    pub width: i32,
    pub height: i32,
    pub data_offset: u32,
    pub row_stride: u32,
    pub pixels: *mut u8,
}

#[link(name = "jpeg_hip")]
extern "C" {
    pub fn jpg_create(out: *mut *mut JpgCtx, device: c_int, dct_algo: c_int, a: c_int, b: c_int, c: c_int) -> c_int;
    pub fn jpg_destroy(ctx: *mut JpgCtx);
    pub fn jpg_load_bgr(ctx: *mut JpgCtx, pixels: *const u8, w: i32, h: i32, stride: u32, bottom_up: c_int) -> c_int;
    pub fn jpg_downsample(ctx: *mut JpgCtx) -> c_int;
    pub fn jpg_dct_quant(ctx: *mut JpgCtx) -> c_int;
    pub fn jpg_entropy(ctx: *mut JpgCtx, nbits: *mut u64) -> c_int;
    pub fn jpg_stuff(ctx: *mut JpgCtx, len: *mut usize) -> c_int;
    pub fn jpg_get_scan(ctx: *mut JpgCtx, out: *mut u8, cap: usize) -> c_int;
    pub fn jpg_get_plane(ctx: *mut JpgCtx, comp: c_int, out: *mut u8, cap: usize) -> c_int;
    pub fn jpg_get_coeffs(ctx: *mut JpgCtx, comp: c_int, out: *mut i16, cap: usize) -> c_int;
    pub fn jpg_plane_dims(ctx: *mut JpgCtx, comp: c_int, w: *mut i32, h: *mut i32) -> c_int;
    pub fn jpg_coeff_dims(ctx: *mut JpgCtx, comp: c_int, w: *mut i32, h: *mut i32) -> c_int;
    pub fn jpg_encode_scan(ctx: *mut JpgCtx, pixels: *const u8, w: i32, h: i32, stride: u32, bottom_up: c_int,
                           stuffed: *mut *mut u8, len: *mut usize) -> c_int;
    pub fn jpg_free(p: *mut u8);
    pub fn jpg_strerror(status: c_int) -> *const c_char;
}

#[link(name = "jpeg_host")]
extern "C" {
    pub fn jph_bmp_read(path: *const c_char, out: *mut JphBmp) -> c_int;
    pub fn jph_bmp_free(bmp: *mut JphBmp);
    pub fn jph_jpeg_write(path: *const c_char, w: i32, h: i32, hs: c_int, vs: c_int, scan: *const u8, len: usize) -> c_int;
}

/// The reference panics where the library returns a status (src/jpeg/sampling.rs:19,
/// src/jpeg/entropy_coding.rs:154,190,212,221): same message, same exit code 101.
pub fn check(status: c_int) {
    if status != 0 {
        let msg = unsafe { std::ffi::CStr::from_ptr(jpg_strerror(status)) }.to_string_lossy().into_owned();
        panic!("{}", msg);
    }
}
