/*
 * refcpu.h -- CPU restatement of the uriGrif/jpeg-encoder encode path.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is the parity checker for the HIP
 * path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load it.  Nothing under jpeg-encoder_amd/ links, imports or calls it.
 *
 * Parity status: the reference is Rust and cannot be compiled in this image
 * (no cargo/rustc, crates not vendored), and its own tests only pin the block
 * walk order (block_iterator.rs:172-191) and a loose mean-|error|<=1 check on
 * one 8x8 block (dct_quant.rs:242-322).  Those pins are enforced in
 * tests/test_oracle.py.  Everything else (colour conversion, chroma
 * subsampling, run-length/Huffman coding, the zero-bit tail padding that comes
 * from the third-party crate bitvec 1.0.1, byte stuffing, JFIF header bytes)
 * is "parity unpinned": it follows the reference source text only.
 */
#ifndef REFCPU_H
#define REFCPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* error codes == the reference's panics */
#define REF_OK 0
#define REF_ERR_RATIO -1    /* "Invalid chrominance downsampling ratio!"  sampling.rs:19 */
#define REF_ERR_DC_RANGE -2 /* "DC coefficient bit length greater than 11!" entropy_coding.rs:154 */
#define REF_ERR_AC_RANGE -3 /* "AC coefficient bit length greater than 10!" entropy_coding.rs:190 */
#define REF_ERR_HUFF_DC -4  /* "DC Huffman Code Not Found!" entropy_coding.rs:212 */
#define REF_ERR_HUFF_AC -5  /* "AC Huffman Code Not Found!" entropy_coding.rs:221 */
#define REF_ERR_ARG -6
#define REF_ERR_IO -7       /* "Could not open bitmap image file!" bmp_image.rs:26 */

#define REF_DCT_REAL 0
#define REF_DCT_BIN 1

/* how forward_real_dct_and_quant obtains its cosines */
#define REF_COS_CALL 0  /* cosf() per term, as dct_quant.rs:222-223 does */
#define REF_COS_TABLE 1 /* the same 64 cosf() values, evaluated once */

typedef struct ref_image ref_image;

/* ---- unit functions (known-answer tests) ---- */
void ref_rgb_to_ycbcr(uint8_t r, uint8_t g, uint8_t b, uint8_t out[3]);
void ref_real_dct_block(const uint8_t in[64], const uint8_t q[64], int16_t out[64], int cos_mode);
void ref_bin_dct_block(const uint8_t in[64], const uint8_t q[64], int16_t out[64]);
void ref_bin_dct_block_raw(const uint8_t in[64], int32_t out[64]); /* lifting output before quantisation */
void ref_cos_table(float out[64]);                                /* out[u*8+x] */
const uint8_t *ref_quant_table(int chroma);                       /* natural order, 64 entries */
const int *ref_zigzag(void);
/* table: 0 YDC 1 CHDC 2 YAC 3 CHAC; returns 1 when the symbol has a code */
int ref_huffman_code(int table, uint8_t symbol, uint32_t *code, uint8_t *len);
int ref_downsampling_factor(int a, int b, int c, int *h, int *v);
/* block walk used by the reference's own unit test (block_iterator.rs:172-191) */
size_t ref_block_walk_max_i32(const int32_t *px, int w, int h, int bw, int bh, int pad, int32_t *out_max);

/* ---- pipeline, one call per JpegImage method ---- */
/* from_bmp minus file parsing: pixels are BMP rows (BGR, `stride` bytes apart,
 * bottom-up when bottom_up != 0). */
ref_image *ref_from_bgr(const uint8_t *pixels, int32_t width, int32_t height, uint32_t stride,
                        int bottom_up, int ra, int rb, int rc, int algo, int *err);
/* from_bmp including the reference's file reader (bmp_image.rs:21-90) */
ref_image *ref_from_bmp_file(const char *path, int ra, int rb, int rc, int algo, int *err);
int ref_chrominance_downsampling(ref_image *im, int threads);
int ref_dct_and_quantization(ref_image *im, int cos_mode, int threads);
int ref_generate_entropy_encoded_data(ref_image *im);
/* whole file image in memory; returns the byte count (call with out==NULL to size) */
size_t ref_generate_file(const ref_image *im, uint8_t *out, size_t cap);
/* only the stuffed scan bytes (file.rs:92-103) */
size_t ref_stuffed_scan(const ref_image *im, uint8_t *out, size_t cap);
void ref_free(ref_image *im);

/* ---- accessors ---- */
int ref_width(const ref_image *im);
int ref_height(const ref_image *im);
/* comp: 0 Y, 1 Cb, 2 Cr.  w/h are the *declared* dims; len is the Vec length */
const uint8_t *ref_plane(const ref_image *im, int comp, int *w, int *h, size_t *len);
const int16_t *ref_coeffs(const ref_image *im, int comp, int *w, int *h);
const uint8_t *ref_bits(const ref_image *im, uint64_t *nbits, size_t *nbytes);
/* test aid (no counterpart in the reference): replace one component's quantised coefficients, w x h as
 * ref_coeffs reports them, so that the entropy coder can be run on coefficients computed elsewhere */
int ref_set_coeffs(ref_image *im, int comp, const int16_t *src, int w, int h);

/* BMP reader alone: returns top-down RGB triples (malloc'd, caller frees with ref_free_buf) */
uint8_t *ref_bmp_load_rgb(const char *path, int32_t *width, int32_t *height, int *err);
void ref_free_buf(void *p);

#ifdef __cplusplus
}
#endif
#endif
