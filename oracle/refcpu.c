/*
 * refcpu.c -- CPU restatement of the uriGrif/jpeg-encoder encode path.
 *
 * TEST INFRASTRUCTURE ONLY (see refcpu.h).  Each function cites the reference
 * lines (paths under /root/reference/src) whose behaviour it restates.  The
 * arithmetic is kept in the reference's number types and evaluation order:
 * f32, left-associated, never contracted into FMA (build with
 * -ffp-contract=off), `as u8`/`as i16` casts saturating and truncating toward
 * zero, integer `/` truncating toward zero, `>>` arithmetic.
 */
#include "refcpu.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ tables */

/* quant_tables.rs:2-23, natural (row-major) order */
static const uint8_t QT_Y[64] = {
    16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,
    14, 13, 16, 24, 40,  57,  69,  56,  14, 17, 22, 29, 51,  87,  80,  62,
    18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
    49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
static const uint8_t QT_C[64] = {
    17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99,
    24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};

/* huffman_tables.rs:97-101 */
static const int ZIGZAG[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
                               12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                               35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
                               58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

/* huffman_tables.rs:45-95: offsets[i]..offsets[i+1] index the symbols whose
 * code is i+1 bits long. */
typedef struct {
    uint8_t offsets[17];
    const uint8_t *symbols;
    uint32_t codes[176];
} huff_table;

static const uint8_t SYM_DC[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static const uint8_t SYM_YAC[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07,
    0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0,
    0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28,
    0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49,
    0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69,
    0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89,
    0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7,
    0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5,
    0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
    0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8,
    0xf9, 0xfa};
static const uint8_t SYM_CAC[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71,
    0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0,
    0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26,
    0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48,
    0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68,
    0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87,
    0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5,
    0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
    0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8,
    0xf9, 0xfa};

static huff_table HT[4] = {
    {{0, 0, 1, 6, 7, 8, 9, 10, 11, 12, 12, 12, 12, 12, 12, 12, 12}, SYM_DC, {0}},
    {{0, 0, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 12, 12, 12, 12, 12}, SYM_DC, {0}},
    {{0, 0, 2, 3, 6, 9, 11, 15, 18, 23, 28, 32, 36, 36, 36, 37, 162}, SYM_YAC, {0}},
    {{0, 0, 2, 3, 5, 9, 13, 16, 20, 27, 32, 36, 40, 40, 41, 43, 162}, SYM_CAC, {0}},
};
static pthread_once_t ht_once = PTHREAD_ONCE_INIT;

/* huffman_tables.rs:18-29 canonical code assignment */
static void ht_generate(void) {
    for (int t = 0; t < 4; t++) {
        uint32_t code = 0;
        for (int i = 0; i < 16; i++) {
            for (int j = HT[t].offsets[i]; j < HT[t].offsets[i + 1]; j++) HT[t].codes[j] = code++;
            code <<= 1;
        }
    }
}

/* huffman_tables.rs:31-42 */
int ref_huffman_code(int table, uint8_t symbol, uint32_t *code, uint8_t *len) {
    pthread_once(&ht_once, ht_generate);
    const huff_table *t = &HT[table];
    for (int i = 0; i < 16; i++)
        for (int j = t->offsets[i]; j < t->offsets[i + 1]; j++)
            if (t->symbols[j] == symbol) {
                *code = t->codes[j];
                *len = (uint8_t)(i + 1);
                return 1;
            }
    return 0;
}

const uint8_t *ref_quant_table(int chroma) { return chroma ? QT_C : QT_Y; }
const int *ref_zigzag(void) { return ZIGZAG; }

/* ------------------------------------------------------------- containers */

typedef struct {
    int w, h;    /* declared dimensions (PixelMatrix.width/height) */
    size_t len;  /* Vec length; may exceed w*h after chrominance_downsampling */
    uint8_t *px;
} plane_u8;

typedef struct {
    int w, h;
    int16_t *px;
} plane_i16;

typedef struct {
    uint8_t *buf;
    size_t cap;
    uint64_t nbits;
} bitvec;

struct ref_image {
    int width, height;
    int ra, rb, rc, h, v;
    int algo;
    plane_u8 ch[3];
    plane_i16 co[3];
    bitvec bits;
};

static int div_ceil(int a, int b) { return (a + b - 1) / b; }

/* Rust `as u8` on f32: truncate toward zero, saturate, NaN -> 0 */
static uint8_t f32_as_u8(float f) {
    if (!(f > 0.0f)) return 0;
    if (f >= 255.0f) return 255;
    return (uint8_t)f;
}
/* Rust `as i16` on f32 */
static int16_t f32_as_i16(float f) {
    if (f != f) return 0;
    if (f <= -32768.0f) return INT16_MIN;
    if (f >= 32767.0f) return INT16_MAX;
    return (int16_t)f;
}

/* sampling.rs:6-22 */
int ref_downsampling_factor(int a, int b, int c, int *h, int *v) {
    if (a == 4 && b == 4 && c == 4) { *h = 1; *v = 1; return REF_OK; }
    if (a == 4 && b == 2 && c == 0) { *h = 2; *v = 2; return REF_OK; }
    if (a == 4 && b == 2 && c == 2) { *h = 2; *v = 1; return REF_OK; }
    return REF_ERR_RATIO;
}

/* ------------------------------------------------------ colour conversion */

/* colorspace.rs:5-15 */
void ref_rgb_to_ycbcr(uint8_t r8, uint8_t g8, uint8_t b8, uint8_t out[3]) {
    float r = (float)r8, g = (float)g8, b = (float)b8;
    float y = 0.299f * r + 0.587f * g;
    y = y + 0.114f * b;
    float cb = 128.0f - 0.168736f * r;
    cb = cb - 0.331264f * g;
    cb = cb + 0.5f * b;
    float cr = 128.0f + 0.5f * r;
    cr = cr - 0.418688f * g;
    cr = cr - 0.081312f * b;
    out[0] = f32_as_u8(y);
    out[1] = f32_as_u8(cb);
    out[2] = f32_as_u8(cr);
}

/* ----------------------------------------------------------- block walker */

/* block_iterator.rs:100-120: gather block `idx` row-major; cells outside the
 * declared w x h are skipped, or pushed as 0 when pad != 0.  Returns the count. */
static int gather_u8(const plane_u8 *p, int bw, int bh, int idx, int pad, uint8_t *out) {
    int bpr = div_ceil(p->w, bw);
    int si = (idx / bpr) * bh, sj = (idx % bpr) * bw, n = 0;
    for (int i = 0; i < bh; i++)
        for (int j = 0; j < bw; j++) {
            int r = si + i, c = sj + j;
            if (r < p->h && c < p->w) out[n++] = p->px[(size_t)r * p->w + c];
            else if (pad) out[n++] = 0;
        }
    return n;
}
static int gather_i16(const plane_i16 *p, int bw, int bh, int idx, int pad, int16_t *out) {
    int bpr = div_ceil(p->w, bw);
    int si = (idx / bpr) * bh, sj = (idx % bpr) * bw, n = 0;
    for (int i = 0; i < bh; i++)
        for (int j = 0; j < bw; j++) {
            int r = si + i, c = sj + j;
            if (r < p->h && c < p->w) out[n++] = p->px[(size_t)r * p->w + c];
            else if (pad) out[n++] = 0;
        }
    return n;
}

/* The reference's own test of the walker (block_iterator.rs:172-191) takes the
 * maximum of every block of an i32 matrix. */
size_t ref_block_walk_max_i32(const int32_t *px, int w, int h, int bw, int bh, int pad, int32_t *out_max) {
    int bpr = div_ceil(w, bw), bpc = div_ceil(h, bh);
    size_t k = 0;
    for (int idx = 0; idx < bpr * bpc; idx++) {
        int si = (idx / bpr) * bh, sj = (idx % bpr) * bw, have = 0;
        int32_t m = 0;
        for (int i = 0; i < bh; i++)
            for (int j = 0; j < bw; j++) {
                int r = si + i, c = sj + j;
                int32_t val;
                if (r < h && c < w) val = px[(size_t)r * w + c];
                else if (pad) val = 0;
                else continue;
                if (!have || val > m) m = val;
                have = 1;
            }
        out_max[k++] = m;
    }
    return k;
}

/* ------------------------------------------------------------ construction */

static void free_image(ref_image *im) {
    if (!im) return;
    for (int c = 0; c < 3; c++) { free(im->ch[c].px); free(im->co[c].px); }
    free(im->bits.buf);
    free(im);
}
void ref_free(ref_image *im) { free_image(im); }
void ref_free_buf(void *p) { free(p); }

static int round_up(int x, int m) { return x % m == 0 ? x : x + m - x % m; }

/* jpeg_image.rs:24-102 */
static ref_image *image_new(int width, int height, int ra, int rb, int rc, int algo, int *err) {
    int h, v;
    int e = ref_downsampling_factor(ra, rb, rc, &h, &v);
    if (e) { if (err) *err = e; return NULL; }
    if (width <= 0 || height <= 0 || (algo != REF_DCT_REAL && algo != REF_DCT_BIN)) {
        if (err) *err = REF_ERR_ARG;
        return NULL;
    }
    ref_image *im = (ref_image *)calloc(1, sizeof *im);
    im->width = width; im->height = height;
    im->ra = ra; im->rb = rb; im->rc = rc; im->h = h; im->v = v; im->algo = algo;
    int pw = round_up(width, 8 * h), ph = round_up(height, 8 * v);
    /* sampling.rs:24-44 */
    int dw = round_up(width / h, 8), dh = round_up(height / v, 8);
    for (int c = 0; c < 3; c++) {
        im->ch[c].w = pw; im->ch[c].h = ph; im->ch[c].len = (size_t)pw * ph;
        im->ch[c].px = (uint8_t *)calloc(im->ch[c].len, 1);
    }
    im->co[0].w = pw; im->co[0].h = ph;
    im->co[1].w = im->co[2].w = dw; im->co[1].h = im->co[2].h = dh;
    for (int c = 0; c < 3; c++)
        im->co[c].px = (int16_t *)calloc((size_t)im->co[c].w * im->co[c].h, sizeof(int16_t));
    if (err) *err = REF_OK;
    return im;
}

/* jpeg_image.rs:104-137 with bmp_image.rs:57-90's row order / BGR swap folded
 * in: image row i (top-down) is BMP row height-1-i when the rows are bottom-up. */
ref_image *ref_from_bgr(const uint8_t *pixels, int32_t width, int32_t height, uint32_t stride,
                        int bottom_up, int ra, int rb, int rc, int algo, int *err) {
    ref_image *im = image_new(width, height, ra, rb, rc, algo, err);
    if (!im) return NULL;
    int pw = im->ch[0].w;
    for (int i = 0; i < height; i++) {
        const uint8_t *row = pixels + (size_t)(bottom_up ? height - 1 - i : i) * stride;
        for (int j = 0; j < width; j++) {
            uint8_t o[3];
            ref_rgb_to_ycbcr(row[3 * j + 2], row[3 * j + 1], row[3 * j + 0], o);
            for (int c = 0; c < 3; c++) im->ch[c].px[(size_t)i * pw + j] = o[c];
        }
    }
    return im;
}

/* bmp_image.rs:21-90.  Only the three header fields the reference reads are
 * used.  Reads are 3 bytes per pixel; a short read leaves the rest of the
 * 3-byte buffer as it was (the reference ignores read() results). */
uint8_t *ref_bmp_load_rgb(const char *path, int32_t *width, int32_t *height, int *err) {
    FILE *f = fopen(path, "rb");
    if (!f) { if (err) *err = REF_ERR_IO; return NULL; }
    uint8_t m[4];
    uint32_t off = 0; int32_t w = 0, h = 0;
    memset(m, 0, 4); fseek(f, 10, SEEK_SET); if (fread(m, 1, 4, f)) {}
    off = (uint32_t)m[0] | (uint32_t)m[1] << 8 | (uint32_t)m[2] << 16 | (uint32_t)m[3] << 24;
    fseek(f, 18, SEEK_SET); if (fread(m, 1, 4, f)) {}
    w = (int32_t)((uint32_t)m[0] | (uint32_t)m[1] << 8 | (uint32_t)m[2] << 16 | (uint32_t)m[3] << 24);
    fseek(f, 22, SEEK_SET); if (fread(m, 1, 4, f)) {}
    h = (int32_t)((uint32_t)m[0] | (uint32_t)m[1] << 8 | (uint32_t)m[2] << 16 | (uint32_t)m[3] << 24);
    if (w <= 0 || h <= 0) { fclose(f); if (err) *err = REF_ERR_ARG; return NULL; }
    uint8_t *rgb = (uint8_t *)calloc((size_t)w * h, 3);
    fseek(f, (long)off, SEEK_SET);
    uint32_t skip = (((uint32_t)w * 3 + 3) / 4) * 4 - (uint32_t)w * 3;
    uint8_t px[3] = {0, 0, 0};
    for (int row = h - 1; row >= 0; row--)
        for (int col = 0; col < w; col++) {
            if (fread(px, 1, 3, f)) {}
            uint8_t *d = rgb + ((size_t)row * w + col) * 3;
            d[0] = px[2]; d[1] = px[1]; d[2] = px[0];
            if (col == w - 1)
                for (uint32_t k = 0; k < skip; k++) { uint8_t a; if (fread(&a, 1, 1, f)) {} }
        }
    fclose(f);
    *width = w; *height = h;
    if (err) *err = REF_OK;
    return rgb;
}

ref_image *ref_from_bmp_file(const char *path, int ra, int rb, int rc, int algo, int *err) {
    int32_t w, h;
    int e = 0;
    uint8_t *rgb = ref_bmp_load_rgb(path, &w, &h, &e);
    if (!rgb) { if (err) *err = e; return NULL; }
    ref_image *im = image_new(w, h, ra, rb, rc, algo, err);
    if (im) {
        int pw = im->ch[0].w;
        for (int i = 0; i < h; i++)
            for (int j = 0; j < w; j++) {
                const uint8_t *s = rgb + ((size_t)i * w + j) * 3;
                uint8_t o[3];
                ref_rgb_to_ycbcr(s[0], s[1], s[2], o);
                for (int c = 0; c < 3; c++) im->ch[c].px[(size_t)i * pw + j] = o[c];
            }
    }
    free(rgb);
    return im;
}

/* ------------------------------------------------------ chroma subsampling */

typedef struct { ref_image *im; int comp; } down_job;

/* sampling.rs:46-102: one value floor(sum/len) per h x v block of the padded
 * plane, pushed in block order; the new plane *declares* the downsampled dims. */
static void *down_one(void *arg) {
    down_job *j = (down_job *)arg;
    ref_image *im = j->im;
    plane_u8 *src = &im->ch[j->comp];
    int h = im->h, v = im->v;
    int nblk = div_ceil(src->w, h) * div_ceil(src->h, v);
    plane_u8 dst;
    dst.w = round_up(im->width / h, 8);
    dst.h = round_up(im->height / v, 8);
    dst.len = (size_t)nblk;
    dst.px = (uint8_t *)malloc(dst.len ? dst.len : 1);
    uint8_t buf[4];
    for (int b = 0; b < nblk; b++) {
        int n = gather_u8(src, h, v, b, 0, buf);
        size_t sum = 0;
        for (int k = 0; k < n; k++) sum += buf[k];
        dst.px[b] = (uint8_t)(sum / (size_t)n);
    }
    free(src->px);
    *src = dst;
    return NULL;
}

int ref_chrominance_downsampling(ref_image *im, int threads) {
    if (im->h == 1 && im->v == 1) return REF_OK; /* sampling.rs:52-54 */
    down_job jobs[2] = {{im, 1}, {im, 2}};
    if (threads > 1) {
        pthread_t t[2];
        for (int k = 0; k < 2; k++) pthread_create(&t[k], NULL, down_one, &jobs[k]);
        for (int k = 0; k < 2; k++) pthread_join(t[k], NULL);
    } else {
        for (int k = 0; k < 2; k++) down_one(&jobs[k]);
    }
    return REF_OK;
}

/* ------------------------------------------------------------ transforms */

#define PI_F32 3.14159274101257324f   /* std::f32::consts::PI */
#define SQRT2_F32 1.41421356237309505f /* std::f32::consts::SQRT_2 rounds to 0x3FB504F3 */

void ref_cos_table(float out[64]) {
    for (int u = 0; u < 8; u++)
        for (int x = 0; x < 8; x++)
            out[u * 8 + x] = cosf(((float)((2 * x + 1) * u) * PI_F32) / 16.0f);
}

/* dct_quant.rs:63-65 */
static int8_t shift_range(uint8_t n) { return (int8_t)(n <= 127 ? (n | 128u) : (n & 127u)); }

/* dct_quant.rs:189-234 */
void ref_real_dct_block(const uint8_t in[64], const uint8_t q[64], int16_t out[64], int cos_mode) {
    const float inv_sqrt2 = 1.0f / (float)SQRT2_F32;
    float ct[64];
    if (cos_mode == REF_COS_TABLE) ref_cos_table(ct);
    int qi = 0;
    for (int u = 0; u < 8; u++) {
        float au = u == 0 ? inv_sqrt2 : 1.0f;
        for (int v = 0; v < 8; v++) {
            float av = v == 0 ? inv_sqrt2 : 1.0f;
            float sum = 0.0f;
            for (int x = 0; x < 8; x++)
                for (int y = 0; y < 8; y++) {
                    float e = (float)shift_range(in[x * 8 + y]);
                    float cu, cv;
                    if (cos_mode == REF_COS_TABLE) {
                        cu = ct[u * 8 + x];
                        cv = ct[v * 8 + y];
                    } else {
                        cu = cosf(((float)((2 * x + 1) * u) * PI_F32) / 16.0f);
                        cv = cosf(((float)((2 * y + 1) * v) * PI_F32) / 16.0f);
                    }
                    float t = e * cu;
                    t = t * cv;
                    sum = sum + t;
                }
            float s = 0.25f * au;
            s = s * av;
            s = s * sum;
            out[qi] = f32_as_i16(s / (float)q[qi]);
            qi++;
        }
    }
}

/* dct_quant.rs:93-129: one 8-point all-lifting binDCT-C pass on elements
 * p[0], p[st], ..., p[7*st], written back in the reference's output order. */
static void bin_pass(int32_t *p, int st) {
    int32_t x0 = p[0], x1 = p[st], x2 = p[2 * st], x3 = p[3 * st];
    int32_t x4 = p[4 * st], x5 = p[5 * st], x6 = p[6 * st], x7 = p[7 * st];
    int32_t x7_1 = x0 - x7, x0_1 = x0 - (x7_1 >> 1);
    int32_t x6_1 = x1 - x6, x1_1 = x1 - (x6_1 >> 1);
    int32_t x5_1 = x2 - x5, x2_1 = x2 - (x5_1 >> 1);
    int32_t x4_1 = x3 - x4, x3_1 = x3 - (x4_1 >> 1);
    x6_1 = ((x5_1 * 3) >> 3) + x6_1;
    x5_1 = ((x6_1 * 5) >> 3) - x5_1;
    int32_t x0_2 = x0_1 + x3_1, x3_2 = x0_1 - x3_1;
    int32_t x1_2 = x1_1 + x2_1, x2_2 = x1_1 - x2_1;
    int32_t x4_2 = x4_1 + x5_1, x5_2 = x4_1 - x5_1;
    int32_t x6_2 = x7_1 - x6_1, x7_2 = x7_1 + x6_1;
    x4_2 = x4_2 - (x7_2 >> 3);
    x0_2 = x0_2 + x1_2;
    x1_2 = -x1_2 + (x0_2 >> 1);
    x2_2 = x2_2 - ((x3_2 * 3) >> 3);
    x3_2 = x3_2 + ((x2_2 * 3) >> 3);
    x5_2 = x5_2 + ((x6_2 * 7) >> 3);
    x6_2 = x6_2 - (x5_2 >> 1);
    p[0] = x0_2; p[4 * st] = x1_2; p[6 * st] = x2_2; p[2 * st] = x3_2;
    p[7 * st] = x4_2; p[5 * st] = x5_2; p[3 * st] = x6_2; p[1 * st] = x7_2;
}

void ref_bin_dct_block_raw(const uint8_t in[64], int32_t a[64]) {
    for (int i = 0; i < 64; i++) a[i] = (int32_t)shift_range(in[i]);
    for (int i = 0; i < 8; i++) bin_pass(a + i * 8, 1); /* rows,    dct_quant.rs:83-130 */
    for (int i = 0; i < 8; i++) bin_pass(a + i, 8);     /* columns, dct_quant.rs:133-180 */
}

/* dct_quant.rs:67-187 */
void ref_bin_dct_block(const uint8_t in[64], const uint8_t q[64], int16_t out[64]) {
    int32_t a[64];
    ref_bin_dct_block_raw(in, a);
    for (int i = 0; i < 64; i++) out[i] = (int16_t)(a[i] / (int32_t)q[i]);
}

typedef struct { ref_image *im; int comp; int cos_mode; } dct_job;

/* dct_quant.rs:15-61 + block_iterator.rs:78-90: blocks walked row-major with
 * zero padding; the 64 results of block k land in place in the coefficient
 * plane at the k-th 8x8 cell of *that* plane's own walk. */
static void *dct_one(void *arg) {
    dct_job *j = (dct_job *)arg;
    ref_image *im = j->im;
    const plane_u8 *src = &im->ch[j->comp];
    plane_i16 *dst = &im->co[j->comp];
    const uint8_t *q = j->comp == 0 ? QT_Y : QT_C;
    int nblk = div_ceil(src->w, 8) * div_ceil(src->h, 8);
    int dbpr = div_ceil(dst->w, 8), dnblk = dbpr * div_ceil(dst->h, 8);
    float ct[64];
    ref_cos_table(ct);
    int didx = 0;
    uint8_t blk[64];
    int16_t res[64];
    for (int b = 0; b < nblk; b++) {
        gather_u8(src, 8, 8, b, 1, blk);
        if (im->algo == REF_DCT_BIN) ref_bin_dct_block(blk, q, res);
        else if (j->cos_mode == REF_COS_CALL) ref_real_dct_block(blk, q, res, REF_COS_CALL);
        else {
            /* same arithmetic as ref_real_dct_block with the table hoisted */
            const float inv_sqrt2 = 1.0f / (float)SQRT2_F32;
            float e[64];
            for (int i = 0; i < 64; i++) e[i] = (float)shift_range(blk[i]);
            int qi = 0;
            for (int u = 0; u < 8; u++) {
                float au = u == 0 ? inv_sqrt2 : 1.0f;
                float t[64];
                for (int x = 0; x < 8; x++)
                    for (int y = 0; y < 8; y++) t[x * 8 + y] = e[x * 8 + y] * ct[u * 8 + x];
                for (int v = 0; v < 8; v++) {
                    float av = v == 0 ? inv_sqrt2 : 1.0f;
                    float sum = 0.0f;
                    for (int x = 0; x < 8; x++)
                        for (int y = 0; y < 8; y++) {
                            float p = t[x * 8 + y] * ct[v * 8 + y];
                            sum = sum + p;
                        }
                    float s = 0.25f * au;
                    s = s * av;
                    s = s * sum;
                    res[qi] = f32_as_i16(s / (float)q[qi]);
                    qi++;
                }
            }
        }
        int si = (didx / dbpr) * 8, sj = (didx % dbpr) * 8;
        for (int r = 0; r < 8; r++)
            for (int c = 0; c < 8; c++) dst->px[(size_t)(si + r) * dst->w + sj + c] = res[r * 8 + c];
        didx = didx == dnblk - 1 ? 0 : didx + 1;
    }
    return NULL;
}

int ref_dct_and_quantization(ref_image *im, int cos_mode, int threads) {
    dct_job jobs[3] = {{im, 0, cos_mode}, {im, 1, cos_mode}, {im, 2, cos_mode}};
    if (threads > 1) {
        pthread_t t[3];
        for (int k = 0; k < 3; k++) pthread_create(&t[k], NULL, dct_one, &jobs[k]);
        for (int k = 0; k < 3; k++) pthread_join(t[k], NULL);
    } else {
        for (int k = 0; k < 3; k++) dct_one(&jobs[k]);
    }
    return REF_OK;
}

/* --------------------------------------------------------- entropy coding */

/* bitvec_utils.rs:3-8: MSB-first, one bit at a time; unwritten tail bits of
 * the last byte stay 0 (bitvec 1.0.1 grows with zeroed elements). */
static void write_bits(bitvec *bv, uint32_t value, uint8_t length) {
    for (int i = (int)length - 1; i >= 0; i--) {
        size_t byte = (size_t)(bv->nbits >> 3);
        if (byte >= bv->cap) {
            size_t ncap = bv->cap ? bv->cap * 2 : 4096;
            bv->buf = (uint8_t *)realloc(bv->buf, ncap);
            memset(bv->buf + bv->cap, 0, ncap - bv->cap);
            bv->cap = ncap;
        }
        if ((value >> i) & 1u) bv->buf[byte] |= (uint8_t)(0x80u >> (bv->nbits & 7));
        bv->nbits++;
    }
}

/* entropy_coding.rs:127-134 */
static uint8_t bit_length(int16_t value) {
    uint8_t n = 0;
    while (value > 0) { value >>= 1; n++; }
    return n;
}

typedef struct { uint8_t symbol; int16_t amplitude; } run_length;

/* entropy_coding.rs:140-142 */
static int16_t to_amplitude(int16_t value, uint8_t len) {
    return value < 0 ? (int16_t)(value + (1 << len) - 1) : value;
}

/* entropy_coding.rs:144-199 */
static int runlength_encode(int16_t *prev_dc, const int16_t *co, run_length *out, int *n_out) {
    int n = 0;
    int16_t dc = (int16_t)(co[0] - *prev_dc);
    *prev_dc = co[0];
    uint8_t dl = bit_length((int16_t)(dc < 0 ? -dc : dc));
    if (dl > 11) return REF_ERR_DC_RANGE;
    out[n].symbol = dl; out[n].amplitude = to_amplitude(dc, dl); n++;
    uint8_t zeros = 0;
    int i = 1;
    while (i < 64) {
        while (i < 64 && co[ZIGZAG[i]] == 0) { zeros++; i++; }
        if (i == 64) { out[n].symbol = 0x00; out[n].amplitude = 0; n++; break; }
        while (zeros >= 16) { out[n].symbol = 0xf0; out[n].amplitude = 0; n++; zeros -= 16; }
        int16_t ac = co[ZIGZAG[i]];
        uint8_t al = bit_length((int16_t)(ac < 0 ? -ac : ac));
        if (al > 10) return REF_ERR_AC_RANGE;
        out[n].symbol = (uint8_t)((zeros << 4) | al); out[n].amplitude = to_amplitude(ac, al); n++;
        zeros = 0;
        i++;
    }
    *n_out = n;
    return REF_OK;
}

/* entropy_coding.rs:201-227 */
static int huffman_encode(const run_length *rl, int n, bitvec *bv, int dc_table, int ac_table) {
    for (int i = 0; i < n; i++) {
        uint32_t code; uint8_t len;
        if (i == 0) {
            if (!ref_huffman_code(dc_table, rl[i].symbol, &code, &len)) return REF_ERR_HUFF_DC;
            write_bits(bv, code, len);
            write_bits(bv, (uint32_t)(int32_t)rl[i].amplitude, rl[i].symbol & 0x0f);
            continue;
        }
        if (!ref_huffman_code(ac_table, rl[i].symbol, &code, &len)) return REF_ERR_HUFF_AC;
        write_bits(bv, code, len);
        if (rl[i].symbol != 0xf0 && rl[i].symbol != 0x00)
            write_bits(bv, (uint32_t)(int32_t)rl[i].amplitude, rl[i].symbol & 0x0f);
    }
    return REF_OK;
}

static int encode_block(const int16_t *co, int16_t *prev_dc, bitvec *bv, int dc_table, int ac_table) {
    run_length rl[80];
    int n = 0;
    int e = runlength_encode(prev_dc, co, rl, &n);
    if (e) return e;
    return huffman_encode(rl, n, bv, dc_table, ac_table);
}

/* entropy_coding.rs:16-125: one trip per Cb block; trip i takes Y region i
 * (8h x 8v cells of the Y plane, its 8x8 sub-blocks row-major), Cb block i,
 * Cr block i.  Three DC predictors, never reset. */
int ref_generate_entropy_encoded_data(ref_image *im) {
    pthread_once(&ht_once, ht_generate);
    int h = im->h, v = im->v;
    const plane_i16 *yc = &im->co[0], *cbc = &im->co[1], *crc = &im->co[2];
    int y_n = div_ceil(yc->w, 8 * h) * div_ceil(yc->h, 8 * v);
    int cb_n = div_ceil(cbc->w, 8) * div_ceil(cbc->h, 8);
    int cr_n = div_ceil(crc->w, 8) * div_ceil(crc->h, 8);
    int yi = 0, cbi = 0, cri = 0;
    int16_t py = 0, pcb = 0, pcr = 0;
    int16_t region[256], blk[64];
    free(im->bits.buf);
    memset(&im->bits, 0, sizeof im->bits);
    for (int i = 0; i < cb_n; i++) {
        if (i != 0) {
            yi = yi == y_n - 1 ? 0 : yi + 1;
            cbi = cbi == cb_n - 1 ? 0 : cbi + 1;
            cri = cri == cr_n - 1 ? 0 : cri + 1;
        }
        int n = gather_i16(yc, 8 * h, 8 * v, yi, 0, region);
        /* new_from_pixels(8h, 8v, region).get_block_iterator(8,8,false) */
        plane_i16 sub = {8 * h, 8 * v, region};
        (void)n;
        int sub_n = div_ceil(sub.w, 8) * div_ceil(sub.h, 8);
        for (int s = 0; s < sub_n; s++) {
            gather_i16(&sub, 8, 8, s, 0, blk);
            int e = encode_block(blk, &py, &im->bits, 0, 2);
            if (e) return e;
        }
        gather_i16(cbc, 8, 8, cbi, 0, blk);
        int e = encode_block(blk, &pcb, &im->bits, 1, 3);
        if (e) return e;
        gather_i16(crc, 8, 8, cri, 0, blk);
        e = encode_block(blk, &pcr, &im->bits, 1, 3);
        if (e) return e;
    }
    return REF_OK;
}

/* ------------------------------------------------------------- JFIF writer */

typedef struct { uint8_t *out; size_t cap, n; } sink;
static void put(sink *s, uint8_t b) { if (s->out && s->n < s->cap) s->out[s->n] = b; s->n++; }
static void put16(sink *s, unsigned v) { put(s, (uint8_t)(v >> 8)); put(s, (uint8_t)v); }

/* file.rs:92-103 */
static void put_scan(sink *s, const ref_image *im) {
    size_t nbytes = (size_t)((im->bits.nbits + 7) >> 3);
    for (size_t i = 0; i < nbytes; i++) {
        put(s, im->bits.buf[i]);
        if (im->bits.buf[i] == 0xff) put(s, 0);
    }
}

size_t ref_stuffed_scan(const ref_image *im, uint8_t *out, size_t cap) {
    sink s = {out, cap, 0};
    put_scan(&s, im);
    return s.n;
}

/* file.rs:105-137 */
size_t ref_generate_file(const ref_image *im, uint8_t *out, size_t cap) {
    pthread_once(&ht_once, ht_generate);
    sink s = {out, cap, 0};
    put(&s, 0xff); put(&s, 0xd8);                       /* file.rs:10-12 */
    put(&s, 0xff); put(&s, 0xe0); put16(&s, 16);        /* file.rs:14-23 */
    put(&s, 'J'); put(&s, 'F'); put(&s, 'I'); put(&s, 'F'); put(&s, 0);
    put(&s, 1); put(&s, 1); put(&s, 1); put16(&s, 72); put16(&s, 72); put(&s, 0); put(&s, 0);
    for (int t = 0; t < 2; t++) {                        /* file.rs:25-37 */
        const uint8_t *q = t ? QT_C : QT_Y;
        put(&s, 0xff); put(&s, 0xdb); put16(&s, 67); put(&s, (uint8_t)t);
        for (int i = 0; i < 64; i++) put(&s, q[ZIGZAG[i]]);
    }
    put(&s, 0xff); put(&s, 0xc0); put16(&s, 17); put(&s, 8); /* file.rs:39-54 */
    put16(&s, (unsigned)(uint16_t)im->height); put16(&s, (unsigned)(uint16_t)im->width);
    put(&s, 3);
    for (int i = 1; i < 4; i++) {
        put(&s, (uint8_t)i);
        put(&s, i == 1 ? (uint8_t)((im->h << 4) | im->v) : 0x11);
        put(&s, i == 1 ? 0 : 1);
    }
    static const int order[4][3] = {{0, 0, 0}, {0, 1, 1}, {1, 0, 2}, {1, 1, 3}}; /* file.rs:122-125 */
    for (int k = 0; k < 4; k++) {                        /* file.rs:56-75 */
        const huff_table *t = &HT[order[k][2]];
        put(&s, 0xff); put(&s, 0xc4); put16(&s, 19u + t->offsets[16]);
        put(&s, (uint8_t)((order[k][0] << 4) | order[k][1]));
        for (int i = 0; i < 16; i++) put(&s, (uint8_t)(t->offsets[i + 1] - t->offsets[i]));
        for (int j = 0; j < t->offsets[16]; j++) put(&s, t->symbols[j]);
    }
    put(&s, 0xff); put(&s, 0xda); put16(&s, 12); put(&s, 3); /* file.rs:77-90 */
    for (int i = 1; i < 4; i++) { put(&s, (uint8_t)i); put(&s, i == 1 ? 0 : 0x11); }
    put(&s, 0); put(&s, 63); put(&s, 0);
    put_scan(&s, im);
    put(&s, 0xff); put(&s, 0xd9);                        /* file.rs:134 */
    return s.n;
}

/* --------------------------------------------------------------- accessors */

int ref_width(const ref_image *im) { return im->width; }
int ref_height(const ref_image *im) { return im->height; }
const uint8_t *ref_plane(const ref_image *im, int comp, int *w, int *h, size_t *len) {
    if (w) *w = im->ch[comp].w;
    if (h) *h = im->ch[comp].h;
    if (len) *len = im->ch[comp].len;
    return im->ch[comp].px;
}
const int16_t *ref_coeffs(const ref_image *im, int comp, int *w, int *h) {
    if (w) *w = im->co[comp].w;
    if (h) *h = im->co[comp].h;
    return im->co[comp].px;
}
/* Test aid without a counterpart in the reference: overwrite the quantised coefficients of one component (same
 * dimensions, the reference's row-major plane layout), so that generate_entropy_encoded_data can be run on
 * coefficients that came from elsewhere. */
int ref_set_coeffs(ref_image *im, int comp, const int16_t *src, int w, int h) {
    if (!im || !src || comp < 0 || comp > 2 || w != im->co[comp].w || h != im->co[comp].h) return REF_ERR_ARG;
    memcpy(im->co[comp].px, src, (size_t)w * (size_t)h * sizeof(int16_t));
    return REF_OK;
}
const uint8_t *ref_bits(const ref_image *im, uint64_t *nbits, size_t *nbytes) {
    if (nbits) *nbits = im->bits.nbits;
    if (nbytes) *nbytes = (size_t)((im->bits.nbits + 7) >> 3);
    return im->bits.buf;
}
