/*
 * jpeg_host.h -- host-side helpers that stay on the CPU: the BMP reader and
 * the JFIF marker writer of the reference, plus the synthetic-input generator
 * used by tests and benchmarks.  No GPU dependency (libjpeg_host.so).
 */
#ifndef JPEG_HOST_H
#define JPEG_HOST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JPH_OK 0
#define JPH_ERR_OPEN -1     /* "Could not open bitmap image file!\n"  src/bmp/bmp_image.rs:26 */
#define JPH_ERR_FORMAT -2   /* width or height not positive (a negative height is a top-down file), bits per pixel
                               (u16 at offset 28) != 24, or compression (u32 at offset 30) != 0: the checks the
                               reference leaves as TODOs (src/bmp/bmp_image.rs:22-24) */
#define JPH_ERR_IO -3
#define JPH_ERR_NOMEM -4
#define JPH_ERR_ARG -5

typedef struct jph_bmp {
    int32_t width, height;  /* i32 at file offsets 18 and 22 (src/bmp/bmp_image.rs:8-9,37-43) */
    uint32_t data_offset;   /* u32 at file offset 10 (src/bmp/bmp_image.rs:7,33-35) */
    uint32_t row_stride;    /* ceil(3*width/4)*4 (src/bmp/bmp_image.rs:68) */
    uint8_t *pixels;        /* height rows of row_stride bytes, file order (bottom-up, B,G,R) */
} jph_bmp;

/* BmpImage::new + load_pixels (src/bmp/bmp_image.rs:21-90) with one buffered
 * read instead of one syscall per pixel.  Like the reference, only the three
 * header fields above are interpreted; unlike it, files that are not uncompressed
 * 24-bit bottom-up bitmaps are refused (JPH_ERR_FORMAT) instead of misread.  If the file ends early the missing
 * pixels repeat the last 3-byte buffer, as the reference's ignored short
 * reads do.  Release with jph_bmp_free. */
int jph_bmp_read(const char *path, jph_bmp *out);
void jph_bmp_free(jph_bmp *bmp);
/* The same with the pixel array placed in memory from `alloc` (for example page-locked
 * memory from jpg_host_alloc, which the device can read without a staging copy).
 * `release` is only used if reading fails; afterwards the caller releases out->pixels
 * itself, not with jph_bmp_free. */
int jph_bmp_read_with(const char *path, jph_bmp *out, void *(*alloc)(size_t), void (*release)(void *));
/* writes a 24-bit bottom-up BMP (54-byte header) from rows in file order */
int jph_bmp_write(const char *path, int32_t width, int32_t height, const uint8_t *rows, uint32_t row_stride);

/* Everything generate_file writes before the scan data (src/jpeg/file.rs:
 * 108-128): SOI, APP0, 2 x DQT, SOF0, 4 x DHT, SOS.  Returns the byte count
 * (623); writes at most cap bytes. */
size_t jph_jfif_header(uint8_t *out, size_t cap, int32_t width, int32_t height, int h, int v);
/* generate_file (src/jpeg/file.rs:105-137): header, the already stuffed scan
 * bytes, EOI, with buffered writes. */
int jph_jpeg_write(const char *path, int32_t width, int32_t height, int h, int v, const uint8_t *scan, size_t scan_len);
/* same into memory; returns the total size (call with out == NULL to size) */
size_t jph_jpeg_assemble(uint8_t *out, size_t cap, int32_t width, int32_t height, int h, int v, const uint8_t *scan,
                         size_t scan_len);

/* ---- stripes: one image cut between MCU rows, one stripe per GPU (no counterpart in the
 * reference, which codes serially; see jpg_entropy_stripe / jpg_stitch in jpeg_hip.h) ---- */

/* Image rows [y0[k], y1[k]) (top-down) of stripe k of n: whole MCU rows (8*v pixel rows), as even
 * as possible; stripes may be empty when there are fewer MCU rows than stripes. */
int jph_plan_stripes(int32_t height, int v, int n, int32_t *y0, int32_t *y1);
/* nbits[k]: bits of stripe k; tails[k]: its last min(32, nbits[k]) bits, right-aligned
 * (jpg_stream_tail).  Out: offsets[k] = bit offset of stripe k in the whole scan; heads[k] =
 * the byte its first byte is OR-ed with (the offsets[k] % 8 bits before it, top-aligned). */
int jph_stitch_plan(int n, const uint64_t *nbits, const uint32_t *tails, uint64_t *offsets, uint8_t *heads);

/* Deterministic synthetic BMP rows (file order: bottom-up, B,G,R):
 * photo-like sinusoid + ramp + noise, or uniform noise when noise_only != 0. */
void jph_synth_bgr(uint8_t *rows, int32_t width, int32_t height, uint32_t row_stride, uint64_t seed, int noise_only);

#ifdef __cplusplus
}
#endif
#endif
