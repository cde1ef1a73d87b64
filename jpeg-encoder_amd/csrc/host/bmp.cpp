// BMP input and synthetic-image generation (host side).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "jpeg_host.h"

static uint32_t rd_u32(const uint8_t *p) {
    return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24;
}

static int bmp_read(const char *path, jph_bmp *out, void *(*alloc)(size_t), void (*release)(void *));

extern "C" int jph_bmp_read(const char *path, jph_bmp *out) { return bmp_read(path, out, malloc, free); }

extern "C" int jph_bmp_read_with(const char *path, jph_bmp *out, void *(*alloc)(size_t), void (*release)(void *)) {
    if (!alloc || !release) return JPH_ERR_ARG;
    return bmp_read(path, out, alloc, release);
}

static int bmp_read(const char *path, jph_bmp *out, void *(*alloc)(size_t), void (*release)(void *)) {
    if (!path || !out) return JPH_ERR_ARG;
    memset(out, 0, sizeof *out);
    FILE *f = fopen(path, "rb");
    if (!f) return JPH_ERR_OPEN;
    uint8_t hdr[34];
    memset(hdr, 0, sizeof hdr);
    const size_t got = fread(hdr, 1, sizeof hdr, f);
    (void)got;  // short headers read as zeros, like the reference's ignored read results
    out->data_offset = rd_u32(hdr + 10);
    out->width = (int32_t)rd_u32(hdr + 18);
    out->height = (int32_t)rd_u32(hdr + 22);
    // The reference interprets nothing else and would read any other pixel format as 24-bit rows
    // (the checks are TODOs at src/bmp/bmp_image.rs:22-24).  Refused here instead of encoding garbage:
    // top-down files (negative height), empty images, anything but uncompressed 24 bits per pixel.
    const unsigned bpp = (unsigned)hdr[28] | (unsigned)hdr[29] << 8;
    const uint32_t compression = rd_u32(hdr + 30);
    if (out->width <= 0 || out->height <= 0 || bpp != 24 || compression != 0) {
        fclose(f);
        return JPH_ERR_FORMAT;
    }
    const uint32_t w3 = (uint32_t)out->width * 3u;
    out->row_stride = (w3 + 3u) / 4u * 4u;
    const size_t need = (size_t)out->row_stride * (size_t)out->height;
    out->pixels = (uint8_t *)alloc(need + 16);
    if (!out->pixels) {
        fclose(f);
        return JPH_ERR_NOMEM;
    }
    if (fseek(f, (long)out->data_offset, SEEK_SET) != 0) {
        fclose(f);
        release(out->pixels);
        out->pixels = nullptr;
        return JPH_ERR_IO;
    }
    const size_t have = fread(out->pixels, 1, need, f);
    fclose(f);
    if (have < need) {
        // The reference reads 3 bytes per pixel and 1 byte per padding byte and
        // ignores short reads, so past the end of the file every pixel repeats
        // the last 3-byte buffer (partly overwritten by a final short read).
        uint8_t buf[3] = {0, 0, 0};
        size_t pos = 0;  // bytes of the pixel array consumed so far
        for (int32_t r = 0; r < out->height; r++) {
            uint8_t *row = out->pixels + (size_t)r * out->row_stride;
            for (int32_t c = 0; c < out->width; c++) {
                for (int k = 0; k < 3; k++)
                    if (pos + k < have) buf[k] = row[3 * c + k];
                pos += (pos + 3 <= have) ? 3 : (pos < have ? have - pos : 0);
                row[3 * c + 0] = buf[0];
                row[3 * c + 1] = buf[1];
                row[3 * c + 2] = buf[2];
            }
            for (uint32_t k = w3; k < out->row_stride; k++)
                if (pos < have) pos++;
        }
    }
    return JPH_OK;
}

extern "C" void jph_bmp_free(jph_bmp *bmp) {
    if (bmp && bmp->pixels) {
        free(bmp->pixels);
        bmp->pixels = nullptr;
    }
}

static void wr_u32(uint8_t *p, uint32_t v) {
    p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24);
}

extern "C" int jph_bmp_write(const char *path, int32_t width, int32_t height, const uint8_t *rows, uint32_t row_stride) {
    if (!path || !rows || width <= 0 || height <= 0) return JPH_ERR_ARG;
    const uint32_t fs = ((uint32_t)width * 3u + 3u) / 4u * 4u;
    uint8_t hdr[54];
    memset(hdr, 0, sizeof hdr);
    hdr[0] = 'B'; hdr[1] = 'M';
    wr_u32(hdr + 2, 54u + fs * (uint32_t)height);
    wr_u32(hdr + 10, 54);
    wr_u32(hdr + 14, 40);
    wr_u32(hdr + 18, (uint32_t)width);
    wr_u32(hdr + 22, (uint32_t)height);
    hdr[26] = 1; hdr[28] = 24;
    wr_u32(hdr + 34, fs * (uint32_t)height);
    wr_u32(hdr + 38, 2835); wr_u32(hdr + 42, 2835);
    FILE *f = fopen(path, "wb");
    if (!f) return JPH_ERR_OPEN;
    bool ok = fwrite(hdr, 1, sizeof hdr, f) == sizeof hdr;
    std::vector<uint8_t> line(fs, 0);
    for (int32_t r = 0; ok && r < height; r++) {
        memcpy(line.data(), rows + (size_t)r * row_stride, (size_t)width * 3);
        ok = fwrite(line.data(), 1, fs, f) == fs;
    }
    ok = (fclose(f) == 0) && ok;
    return ok ? JPH_OK : JPH_ERR_IO;
}

static inline uint64_t splitmix64(uint64_t &s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// pixel(x, y, c) = clamp(128 + 96 sin(2 pi (x f_c + y g_c) / 1024) + (((x + y + 37 c) >> 3) & 31) + noise),
// noise uniform in [-8, 8]; y counts from the top of the image.
extern "C" void jph_synth_bgr(uint8_t *rows, int32_t width, int32_t height, uint32_t row_stride, uint64_t seed,
                              int noise_only) {
    static const int F[3] = {3, 5, 7}, G[3] = {11, 2, 5};
    std::vector<float> sx[3], sy[3], cx[3], cy[3];
    const double k = 2.0 * 3.14159265358979323846 / 1024.0;
    for (int c = 0; c < 3; c++) {
        sx[c].resize(width); cx[c].resize(width); sy[c].resize(height); cy[c].resize(height);
        for (int x = 0; x < width; x++) { sx[c][x] = (float)sin(k * x * F[c]); cx[c][x] = (float)cos(k * x * F[c]); }
        for (int y = 0; y < height; y++) { sy[c][y] = (float)sin(k * y * G[c]); cy[c][y] = (float)cos(k * y * G[c]); }
    }
    for (int32_t fr = 0; fr < height; fr++) {
        const int32_t y = height - 1 - fr;  // file rows are bottom-up
        uint64_t s = seed ^ (0xD1B54A32D192ED03ull * (uint64_t)(y + 1));
        uint8_t *row = rows + (size_t)fr * row_stride;
        for (int32_t x = 0; x < width; x++) {
            const uint64_t r = splitmix64(s);
            for (int c = 0; c < 3; c++) {  // c: 0 R, 1 G, 2 B; stored B,G,R
                int v;
                if (noise_only) {
                    v = (int)((r >> (8 * c)) & 0xff);
                } else {
                    const float wave = sx[c][x] * cy[c][y] + cx[c][x] * sy[c][y];  // sin(a + b)
                    const int noise = (int)((r >> (8 * c + 24)) % 17) - 8;
                    v = 128 + (int)lrintf(96.0f * wave) + (((x + y + 37 * c) >> 3) & 31) + noise;
                    v = v < 0 ? 0 : (v > 255 ? 255 : v);
                }
                row[3 * x + (2 - c)] = (uint8_t)v;
            }
        }
        for (uint32_t p = (uint32_t)width * 3; p < row_stride; p++) row[p] = 0;
    }
}
