//! `JpegImage` with the reference's method names; all compute happens in libjpeg_hip.so.
use crate::hip::*;
use std::ffi::CString;
use std::os::raw::c_int;

/// src/jpeg/dct_quant.rs:8-12; the discriminants are the library's `jpg_dct` values.
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum DctAlgorithm {
    RealDct = 0,
    BinDct = 1,
}

pub struct JpegImage {
    ctx: *mut JpgCtx,
    path: String,
    pub width: i32,
    pub height: i32,
    pub chrominance_downsampling_ratio: (u8, u8, u8),
    pub dct_algorithm: DctAlgorithm,
}

/// (4,4,4) -> (1,1), (4,2,0) -> (2,2), (4,2,2) -> (2,1) (src/jpeg/sampling.rs:6-22)
fn factors(ratio: (u8, u8, u8)) -> (c_int, c_int) {
    match ratio {
        (4, 4, 4) => (1, 1),
        (4, 2, 0) => (2, 2),
        (4, 2, 2) => (2, 1),
        _ => panic!("Invalid chrominance downsampling ratio!"),
    }
}

impl JpegImage {
    /// src/jpeg/jpeg_image.rs:104-137: reads the BMP (one buffered read instead of a read per pixel) and converts
    /// its pixels to the three zero-padded YCbCr planes on the device.
    pub fn from_bmp(bmp_path: &String, jpeg_path: &String, ratio: (u8, u8, u8), algo: DctAlgorithm) -> JpegImage {
        let mut bmp = JphBmp { width: 0, height: 0, data_offset: 0, row_stride: 0, pixels: std::ptr::null_mut() };
        let cpath = CString::new(bmp_path.as_str()).expect("path contains a NUL byte");
        if unsafe { jph_bmp_read(cpath.as_ptr(), &mut bmp) } != 0 {
            panic!("Could not open bitmap image file!\n"); // src/bmp/bmp_image.rs:26
        }
        let mut ctx: *mut JpgCtx = std::ptr::null_mut();
        check(unsafe { jpg_create(&mut ctx, 0, algo as c_int, ratio.0 as c_int, ratio.1 as c_int, ratio.2 as c_int) });
        check(unsafe { jpg_load_bgr(ctx, bmp.pixels, bmp.width, bmp.height, bmp.row_stride, 1) });
        let (width, height) = (bmp.width, bmp.height);
        unsafe { jph_bmp_free(&mut bmp) };
        JpegImage { ctx, path: jpeg_path.clone(), width, height, chrominance_downsampling_ratio: ratio, dct_algorithm: algo }
    }

    /// src/jpeg/sampling.rs:46-102
    pub fn chrominance_downsampling(&mut self) {
        check(unsafe { jpg_downsample(self.ctx) });
    }

    /// src/jpeg/dct_quant.rs:15-61
    pub fn dct_and_quantization(&mut self) {
        check(unsafe { jpg_dct_quant(self.ctx) });
    }

    /// src/jpeg/entropy_coding.rs:16-125
    pub fn generate_entropy_encoded_data(&mut self) {
        let mut nbits = 0u64;
        check(unsafe { jpg_entropy(self.ctx, &mut nbits) });
    }

    /// src/jpeg/file.rs:105-137: the stuffing loop of write_image_data runs on the device, the markers are written
    /// by the host library around the stuffed bytes.
    pub fn generate_file(&mut self) -> std::io::Result<()> {
        let mut len = 0usize;
        check(unsafe { jpg_stuff(self.ctx, &mut len) });
        let mut scan = vec![0u8; len];
        check(unsafe { jpg_get_scan(self.ctx, scan.as_mut_ptr(), len) });
        let (h, v) = factors(self.chrominance_downsampling_ratio);
        let cpath = CString::new(self.path.as_str()).expect("path contains a NUL byte");
        match unsafe { jph_jpeg_write(cpath.as_ptr(), self.width, self.height, h, v, scan.as_ptr(), len) } {
            0 => Ok(()),
            e => Err(std::io::Error::new(std::io::ErrorKind::Other, format!("jph_jpeg_write: status {}", e))),
        }
    }

    /// The public field `y_channel` / `cb_channel` / `cr_channel` (src/jpeg/jpeg_image.rs:14-16) as (plane, width, height).
    pub fn channel(&mut self, comp: usize) -> (Vec<u8>, i32, i32) {
        let (mut w, mut h) = (0i32, 0i32);
        check(unsafe { jpg_plane_dims(self.ctx, comp as c_int, &mut w, &mut h) });
        let mut out = vec![0u8; (w as usize) * (h as usize)];
        check(unsafe { jpg_get_plane(self.ctx, comp as c_int, out.as_mut_ptr(), out.len()) });
        (out, w, h)
    }

    /// The public field `y_dct_coeffs` / `cb_dct_coeffs` / `cr_dct_coeffs` (src/jpeg/jpeg_image.rs:17-19).
    pub fn dct_coeffs(&mut self, comp: usize) -> (Vec<i16>, i32, i32) {
        let (mut w, mut h) = (0i32, 0i32);
        check(unsafe { jpg_coeff_dims(self.ctx, comp as c_int, &mut w, &mut h) });
        let mut out = vec![0i16; (w as usize) * (h as usize)];
        check(unsafe { jpg_get_coeffs(self.ctx, comp as c_int, out.as_mut_ptr(), out.len() * 2) });
        (out, w, h)
    }
}

impl Drop for JpegImage {
    fn drop(&mut self) {
        unsafe { jpg_destroy(self.ctx) }
    }
}
