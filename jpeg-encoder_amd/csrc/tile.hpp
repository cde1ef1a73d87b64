// One tile of the fused front end, shared by k_front (front.hip: coefficients to
// memory) and k_coded (coded.hip: coefficients straight into the entropy coder).
// A tile is 512 pixels (1024 at 4:2:2) x one MCU row (8*v image rows):
//   tile_convert    every thread converts groups of 4 pixels x v rows read straight
//                   from global memory (12 contiguous bytes per row and lane, so a
//                   wave reads 768 contiguous bytes) and leaves Y and the averaged
//                   Cb/Cr as bytes in LDS (src/jpeg/jpeg_image.rs:121-134,
//                   src/jpeg/sampling.rs:46-102);
//   tile_transform  one thread per 8x8 block (wave-uniform component class, so the
//                   quantisation table is a compile-time constant) transforms it in
//                   registers and parks the 128-byte zigzag block in LDS
//                   (src/jpeg/dct_quant.rs:15-61).
#pragma once
#include "internal.hpp"
#include "transform.hpp"

namespace jpg {

// Pixels per tile row.  At 4:2:2 an MCU row is only 8 pixel rows high: the wider tile gives the workgroup
// as many pixels (8192) and blocks (256, one per thread) as it takes to keep all four waves busy.
__host__ __device__ constexpr int tile_width(int hs, int vs) { return hs == 2 && vs == 1 ? 1024 : 512; }
// Pixel rows are device memory whatever the pointer's static type says (a pointer read from a struct or an
// array is "generic" to the compiler, which then emits flat loads that also count as LDS operations).
typedef const __attribute__((address_space(1))) uint32_t *global_u32_ptr;
typedef const __attribute__((address_space(1))) uint8_t *global_u8_ptr;
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
constexpr int OUT_STRIDE = 144; // bytes between parked blocks in LDS (128 + 16: conflict-free b128 access)
constexpr int THREADS = 256;

template <int HS, int VS>
struct TileCfg {
    static constexpr int TW = tile_width(HS, VS);             // pixels per tile row
    static constexpr int MCUS = TW / (8 * HS);                // MCUs per tile
    static constexpr int NBLK = HS * VS + 2;                  // blocks per MCU
    static constexpr int TILE_H = 8 * VS;                     // image rows per tile
    static constexpr int ROW_WAVES = TW / 512;                // waves per row of luma blocks (64 blocks each)
    static constexpr int YWAVES = VS * ROW_WAVES;
    static constexpr int CWAVES = (2 * MCUS) / 64;            // chroma blocks: MCUS per component
    static constexpr int CW = TW / HS;                        // chroma samples per tile row
    static constexpr int BLOCKS = MCUS * NBLK;
    static constexpr int ROW_ITEMS = TW / 4;                  // groups of 4 pixels per row
    static constexpr int ITEMS = ROW_ITEMS * 8;               // groups of 4 pixels x VS rows
    static constexpr int ROUND_ROWS = THREADS / ROW_ITEMS;    // item rows the threads cover per round
    static constexpr int ROUNDS = ITEMS / THREADS;            // items per thread in a full tile
    static constexpr int PLANE_BYTES = TILE_H * TW + 2 * 8 * CW;
    static constexpr int OUT_BYTES = BLOCKS * OUT_STRIDE;
    static constexpr int LDS_BYTES = PLANE_BYTES > OUT_BYTES ? PLANE_BYTES : OUT_BYTES;
    // the planes of one more MCU (k_coded: the MCU before the tile)
    static constexpr int HALO_Y = 64 * HS * VS;               // bytes of its luma samples
    static constexpr int HALO_BYTES = HALO_Y + 2 * 64;
    static constexpr int SPARE_WAVE = THREADS / 64 - 1;       // the wave that takes the extra work: it has no block of
                                                              // the tile to transform, except at 4:2:2
    static_assert(YWAVES + CWAVES <= THREADS / 64 && BLOCKS <= THREADS, "one thread per block");
    static_assert(ITEMS % THREADS == 0, "whole rounds");
};

// 4 pixels of one row: 12 bytes B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3 -> packed Y bytes, Cb and Cr as f32
__device__ __forceinline__ void convert4_f(uint32_t d0, uint32_t d1, uint32_t d2, uint32_t &y4, float (&fb)[4], float (&fr)[4]) {
    float fy[4];
    rgb_to_ycbcr_f((float)((d0 >> 16) & 0xff), (float)((d0 >> 8) & 0xff), (float)(d0 & 0xff), fy[0], fb[0], fr[0]);
    rgb_to_ycbcr_f((float)((d1 >> 8) & 0xff), (float)(d1 & 0xff), (float)(d0 >> 24), fy[1], fb[1], fr[1]);
    rgb_to_ycbcr_f((float)(d2 & 0xff), (float)(d1 >> 24), (float)((d1 >> 16) & 0xff), fy[2], fb[2], fr[2]);
    rgb_to_ycbcr_f((float)(d2 >> 24), (float)((d2 >> 16) & 0xff), (float)((d2 >> 8) & 0xff), fy[3], fb[3], fr[3]);
    // `as u8` (truncation; nothing saturates): the four Y bytes are converted into place
    cvt_pack8<0>(y4, fy[0]);
    cvt_pack8<1>(y4, fy[1]);
    cvt_pack8<2>(y4, fy[2]);
    cvt_pack8<3>(y4, fy[3]);
}

// the same with Cb and Cr truncated to integers
__device__ __forceinline__ void convert4(uint32_t d0, uint32_t d1, uint32_t d2, uint32_t &y4, uint32_t (&cb)[4],
                                         uint32_t (&cr)[4]) {
    float fb[4], fr[4];
    convert4_f(d0, d1, d2, y4, fb, fr);
#pragma unroll
    for (int q = 0; q < 4; q++) {
        cb[q] = (uint32_t)fb[q];
        cr[q] = (uint32_t)fr[q];
    }
}

// 4 pixels of each of N rows at once, written operation by operation across all 4N pixels: every instruction's
// operands were produced at least 4N instructions earlier.  (Pixel by pixel, as the compiler orders the code when
// left alone, almost every instruction waits for the one before it; a wave then issues an instruction every 7
// cycles instead of every 4.)  The scheduling barriers keep the order.  Same operations on the same values as
// rgb_to_ycbcr_f (src/utils/colorspace.rs:5-15).
#define JPG_EACH(expr)                              \
    _Pragma("unroll") for (int r = 0; r < N; r++) { \
        _Pragma("unroll") for (int q = 0; q < 4; q++) { expr; } \
    }                                               \
    __builtin_amdgcn_sched_barrier(0)
template <int N>
__device__ __forceinline__ void convert_rows(const uint32_t (&d)[N][3], uint32_t (&y4)[N], float (&fb)[N][4], float (&fr)[N][4]) {
    float R[N][4], G[N][4], B[N][4], fy[N][4], t[N][4], u[N][4], w[N][4];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < N; r++) {
        const uint32_t d0 = d[r][0], d1 = d[r][1], d2 = d[r][2];
        B[r][0] = (float)(d0 & 0xff); G[r][0] = (float)((d0 >> 8) & 0xff); R[r][0] = (float)((d0 >> 16) & 0xff);
        B[r][1] = (float)(d0 >> 24); G[r][1] = (float)(d1 & 0xff); R[r][1] = (float)((d1 >> 8) & 0xff);
        B[r][2] = (float)((d1 >> 16) & 0xff); G[r][2] = (float)(d1 >> 24); R[r][2] = (float)(d2 & 0xff);
        B[r][3] = (float)((d2 >> 8) & 0xff); G[r][3] = (float)((d2 >> 16) & 0xff); R[r][3] = (float)(d2 >> 24);
    }
    __builtin_amdgcn_sched_barrier(0);
    JPG_EACH(t[r][q] = 0.299f * R[r][q]);
    JPG_EACH(u[r][q] = 0.168736f * R[r][q]);
    JPG_EACH(fr[r][q] = fmaf(0.5f, R[r][q], 128.0f));        // exact: k/2 + 128 is representable
    JPG_EACH(fy[r][q] = 0.587f * G[r][q]);
    JPG_EACH(fb[r][q] = 128.0f - u[r][q]);
    JPG_EACH(w[r][q] = 0.418688f * G[r][q]);
    JPG_EACH(fy[r][q] = t[r][q] + fy[r][q]);
    JPG_EACH(u[r][q] = 0.331264f * G[r][q]);
    JPG_EACH(fr[r][q] = fr[r][q] - w[r][q]);
    JPG_EACH(t[r][q] = 0.114f * B[r][q]);
    JPG_EACH(fb[r][q] = fb[r][q] - u[r][q]);
    JPG_EACH(w[r][q] = 0.081312f * B[r][q]);
    JPG_EACH(fy[r][q] = fy[r][q] + t[r][q]);
    JPG_EACH(fb[r][q] = fmaf(0.5f, B[r][q], fb[r][q]));      // 0.5*b is exact, so one rounding either way
    JPG_EACH(fr[r][q] = fr[r][q] - w[r][q]);
    // `as u8` (truncation; nothing saturates): the Y bytes are converted into place, row by row in turn
#pragma unroll
    for (int r = 0; r < N; r++) cvt_pack8<0>(y4[r], fy[r][0]);
#pragma unroll
    for (int r = 0; r < N; r++) cvt_pack8<1>(y4[r], fy[r][1]);
#pragma unroll
    for (int r = 0; r < N; r++) cvt_pack8<2>(y4[r], fy[r][2]);
#pragma unroll
    for (int r = 0; r < N; r++) cvt_pack8<3>(y4[r], fy[r][3]);
}
#undef JPG_EACH

// Truncation of 16 values in [0, 2^22) at the price of additions: with the f32 rounding mode set to "toward
// zero" x + 2^23 is exactly 2^23 + floor(x), whose bit pattern is 0x4B000000 + floor(x).  An addition issues in
// half the cycles of a conversion.  One instruction sequence, so that nothing else runs under the changed mode;
// only the two bits of the f32 rounding mode are written, and put back to "nearest even".
__device__ __forceinline__ void trunc_magic16(float (&x)[16]) {
    asm("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"
        "s_nop 1\n\t"
        "v_add_f32 %0, 0x4b000000, %0\n\t"
        "v_add_f32 %1, 0x4b000000, %1\n\t"
        "v_add_f32 %2, 0x4b000000, %2\n\t"
        "v_add_f32 %3, 0x4b000000, %3\n\t"
        "v_add_f32 %4, 0x4b000000, %4\n\t"
        "v_add_f32 %5, 0x4b000000, %5\n\t"
        "v_add_f32 %6, 0x4b000000, %6\n\t"
        "v_add_f32 %7, 0x4b000000, %7\n\t"
        "v_add_f32 %8, 0x4b000000, %8\n\t"
        "v_add_f32 %9, 0x4b000000, %9\n\t"
        "v_add_f32 %10, 0x4b000000, %10\n\t"
        "v_add_f32 %11, 0x4b000000, %11\n\t"
        "v_add_f32 %12, 0x4b000000, %12\n\t"
        "v_add_f32 %13, 0x4b000000, %13\n\t"
        "v_add_f32 %14, 0x4b000000, %14\n\t"
        "v_add_f32 %15, 0x4b000000, %15\n\t"
        "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0\n\t"
        "s_nop 1"
        : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]),
          "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]));
}

// Chroma samples of row iy of a 4:2:0 image from the Cb / Cr values of 4 pixels x 2 rows given as f32 (x[8r + 0..3]
// Cb, x[8r + 4..7] Cr of row r): truncation as above, then the sums are formed on the bit patterns.  The multiples
// of 0x4B000000 only reach bits the average does not use.  (Measured on 7680x4320: 0.8 % on the whole kernel; with
// one row per sample, 4:2:2, the eight additions do not pay for the two mode switches.)
template <int CW>
__device__ __forceinline__ void store_chroma_f(uint8_t *sCb, uint8_t *sCr, int iy, int ix, float (&x)[16]) {
    trunc_magic16(x);
    uint32_t s[8];
#pragma unroll
    for (int q = 0; q < 8; q++) s[q] = __float_as_uint(x[q]) + __float_as_uint(x[8 + q]);
    // floor(sum / 4) (src/jpeg/sampling.rs:66-73)
    const uint32_t b0 = (s[0] + s[1]) >> 2, b1 = (s[2] + s[3]) >> 2;
    const uint32_t r0 = (s[4] + s[5]) >> 2, r1 = (s[6] + s[7]) >> 2;
    // low bytes of the two averages side by side
    *reinterpret_cast<uint16_t *>(&sCb[iy * CW + 2 * ix]) = (uint16_t)__builtin_amdgcn_perm(b1, b0, 0x0c0c0400u);
    *reinterpret_cast<uint16_t *>(&sCr[iy * CW + 2 * ix]) = (uint16_t)__builtin_amdgcn_perm(r1, r0, 0x0c0c0400u);
}

// the sums of v rows of 4 pixels -> chroma samples of row iy, CW samples per row
template <int HS, int VS, int CW>
__device__ __forceinline__ void store_chroma(uint8_t *sCb, uint8_t *sCr, int iy, int ix, const uint32_t (&cbs)[4],
                                             const uint32_t (&crs)[4]) {
    // box average, floor(sum / (h*v)) (src/jpeg/sampling.rs:66-73)
    if constexpr (HS == 2) {
        const uint32_t b0 = (cbs[0] + cbs[1]) / (2 * VS), b1 = (cbs[2] + cbs[3]) / (2 * VS);
        const uint32_t r0 = (crs[0] + crs[1]) / (2 * VS), r1 = (crs[2] + crs[3]) / (2 * VS);
        *reinterpret_cast<uint16_t *>(&sCb[iy * CW + 2 * ix]) = (uint16_t)(b0 | (b1 << 8));
        *reinterpret_cast<uint16_t *>(&sCr[iy * CW + 2 * ix]) = (uint16_t)(r0 | (r1 << 8));
    } else {
        *reinterpret_cast<uint32_t *>(&sCb[iy * CW + 4 * ix]) = cbs[0] | (cbs[1] << 8) | (cbs[2] << 16) | (cbs[3] << 24);
        *reinterpret_cast<uint32_t *>(&sCr[iy * CW + 4 * ix]) = crs[0] | (crs[1] << 8) | (crs[2] << 16) | (crs[3] << 24);
    }
}

// 4 pixels x VS rows at (x, ytop), any of which may lie outside the image: samples outside W x H
// stay 0 in all three planes (zero-filled planes of src/jpeg/jpeg_image.rs:59-70)
template <int VS>
__device__ __forceinline__ void convert_item_edge(const uint8_t *src, const Geom &g, int x, int ytop, uint32_t (&yw)[VS],
                                                  uint32_t (&cbs)[4], uint32_t (&crs)[4]) {
#pragma unroll
    for (int q = 0; q < 4; q++) cbs[q] = crs[q] = 0;
#pragma unroll
    for (int r = 0; r < VS; r++) {
        const int y = ytop + r;
        uint32_t yy[4] = {0, 0, 0, 0};
        if (y < g.H) {
            const global_u8_ptr p = (global_u8_ptr)(src + (size_t)(g.bottom_up ? g.H - 1 - y : y) * g.stride);
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (x + q < g.W) {
                    const global_u8_ptr c = p + (size_t)3 * (x + q);
                    uint32_t cb, cr;
                    rgb_to_ycbcr((float)c[2], (float)c[1], (float)c[0], yy[q], cb, cr);
                    cbs[q] += cb;
                    crs[q] += cr;
                }
        }
        yw[r] = yy[0] | (yy[1] << 8) | (yy[2] << 16) | (yy[3] << 24);
    }
}

// Colour conversion + chroma average of the tile at (x0, y0) with `mcus` MCUs into sY (TW
// bytes per row), sCb, sCr (CW bytes per row).
struct NoHook {
    __device__ __forceinline__ void operator()(uint32_t, uint32_t) const {}
};

// `loaded` is called with the first and the last dword of pixels between the loads and the arithmetic (instrumented builds)
template <int HS, int VS, class Hook = NoHook>
__device__ __forceinline__ void tile_convert(const uint8_t *__restrict__ src, const Geom &g, int x0, int y0, int mcus,
                                             uint8_t *sY, uint8_t *sCb, uint8_t *sCr, int tid, Hook loaded = Hook()) {
    using C = TileCfg<HS, VS>;
    const int items_x = mcus * 2 * HS;                       // groups of 4 pixels per row that exist
    if (x0 + 4 * items_x <= g.W && y0 + C::TILE_H <= g.H) {
        // every MCU of the tile lies inside the image: every thread owns ROUNDS items (those beyond a
        // partial tile's last MCU are skipped); all loads are issued before the arithmetic
        uint32_t d[C::ROUNDS][VS][3];
        const int ix = tid % C::ROW_ITEMS;
        const bool present = ix < items_x;
        // Buffer loads: the address is a descriptor with the tile's lowest row as its base (scalar registers, set up
        // once), a scalar offset per row and a 32-bit offset per lane, so that no vector instruction and only two
        // scalar ones per row go into addresses.  (Written with pointers the compiler forms a 64-bit address per row
        // and lane: two 4-cycle vector and eight scalar instructions per row, a twentieth of the kernel.)
        // A lane's offset: the 12 bytes per item and the item row of this thread within a round; in a bottom-up file
        // the later row lies lower, so there the rows of a tile are counted from its last one.
        const int sub = tid / C::ROW_ITEMS;                     // 0 .. ROUND_ROWS-1
        const uint32_t lane_off = 12u * (uint32_t)ix +
                                  (uint32_t)(g.bottom_up ? C::ROUND_ROWS - 1 - sub : sub) * (uint32_t)VS * g.stride;
        const size_t low_row = g.bottom_up ? (size_t)(g.H - y0 - C::TILE_H) : (size_t)y0;
        const uint8_t *tile_base = src + low_row * g.stride + (size_t)3 * x0;
        // (descriptor word 3 = 0x00020000 is the gfx9-family layout: 32-bit raw data format, no swizzle; make_geom keeps
        // TILE_H * stride + the widest lane offset below the 2^31 bytes the descriptor covers)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "buffer descriptor and instruction selection of this file are written for gfx950"
#endif
        const __amdgpu_buffer_rsrc_t rows = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(tile_base), 0, 0x7fffffff, 0x00020000);
        // row c of the tile (c = VS * ROUND_ROWS * k + r for the first item row of round k) lies c rows above the base
        // in a top-down file and TILE_H - 1 - VS * (ROUND_ROWS - 1) - c rows in a bottom-up one
        const uint32_t row_step = g.bottom_up ? 0u - g.stride : g.stride;
        const uint32_t row_first = g.bottom_up ? (uint32_t)(C::TILE_H - 1 - VS * (C::ROUND_ROWS - 1)) * g.stride : 0u;
#pragma unroll
        for (int k = 0; k < C::ROUNDS; k++) {
#pragma unroll
            for (int r = 0; r < VS; r++) {
                const uint32_t row_off = row_first + (uint32_t)(VS * C::ROUND_ROWS * k + r) * row_step;
                if (present) {
                    const u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(rows, (int)lane_off, (int)row_off, 0);
                    d[k][r][0] = v.x; d[k][r][1] = v.y; d[k][r][2] = v.z;
                }
            }
        }
        loaded(d[0][0][0], d[C::ROUNDS - 1][VS - 1][2]);
        if (present) {
#pragma unroll
            for (int k = 0; k < C::ROUNDS; k++) {
                const int iy = C::ROUND_ROWS * k + sub;
#if JPG_ABLATE >= 3
                *reinterpret_cast<uint32_t *>(&sY[(iy * VS) * C::TW + 4 * ix]) = d[k][0][0] ^ d[k][0][1] ^ d[k][0][2] ^ d[k][VS - 1][0] ^ d[k][VS - 1][1] ^ d[k][VS - 1][2];
                continue;
#endif
                uint32_t y4[VS];
                float fb[VS][4], fr[VS][4];
                convert_rows<VS>(d[k], y4, fb, fr);
#pragma unroll
                for (int r = 0; r < VS; r++) *reinterpret_cast<uint32_t *>(&sY[(iy * VS + r) * C::TW + 4 * ix]) = y4[r];
                if constexpr (HS == 2 && VS == 2) {
                    float x[16];
#pragma unroll
                    for (int r = 0; r < 2; r++)
#pragma unroll
                        for (int q = 0; q < 4; q++) { x[8 * r + q] = fb[r][q]; x[8 * r + 4 + q] = fr[r][q]; }
                    store_chroma_f<C::CW>(sCb, sCr, iy, ix, x);
                } else {
                    uint32_t cbs[4] = {0, 0, 0, 0}, crs[4] = {0, 0, 0, 0};
#pragma unroll
                    for (int r = 0; r < VS; r++)
#pragma unroll
                        for (int q = 0; q < 4; q++) { cbs[q] += (uint32_t)fb[r][q]; crs[q] += (uint32_t)fr[r][q]; }
                    store_chroma<HS, VS, C::CW>(sCb, sCr, iy, ix, cbs, crs);
                }
            }
        }
    } else {
        // tile cut by the right or bottom edge of the image
        for (int it = tid; it < items_x * 8; it += THREADS) {
            const int ix = it % items_x, iy = it / items_x;
            uint32_t yw[VS], cbs[4], crs[4];
            convert_item_edge<VS>(src, g, x0 + 4 * ix, y0 + iy * VS, yw, cbs, crs);
#pragma unroll
            for (int r = 0; r < VS; r++) *reinterpret_cast<uint32_t *>(&sY[(iy * VS + r) * C::TW + 4 * ix]) = yw[r];
            store_chroma<HS, VS, C::CW>(sCb, sCr, iy, ix, cbs, crs);
        }
    }
}

// The MCU at pixel (hx0, hy0), converted by the first 2*HS*8 threads of the spare wave (t = index
// in that wave) into hY (8*HS bytes per row), hCb, hCr (8 bytes per row).  In two steps, so that the
// loads can be issued before the tile's own and be used after them.
template <int VS>
struct HaloRegs {
    uint32_t d[VS][3];
    bool full;            // the item lies inside the image: its rows were loaded as dwords
};

template <int HS, int VS>
__device__ __forceinline__ void halo_load(const uint8_t *__restrict__ src, const Geom &g, int hx0, int hy0, int t,
                                          HaloRegs<VS> &h) {
    const int ix = t % (2 * HS), iy = t / (2 * HS);
    const int x = hx0 + 4 * ix, ytop = hy0 + iy * VS;
    h.full = t >= 0 && t < 2 * HS * 8 && x + 4 <= g.W && ytop + VS <= g.H;
#pragma unroll
    for (int r = 0; r < VS; r++) {
        h.d[r][0] = h.d[r][1] = h.d[r][2] = 0;
        if (h.full) {
            const int y = ytop + r;
            const size_t row = (size_t)(g.bottom_up ? g.H - 1 - y : y) * g.stride;
            const global_u32_ptr p = (global_u32_ptr)(src + row + (size_t)3 * x);
            h.d[r][0] = p[0]; h.d[r][1] = p[1]; h.d[r][2] = p[2];
        }
    }
}

template <int HS, int VS>
__device__ __forceinline__ void halo_convert(const uint8_t *__restrict__ src, const Geom &g, int hx0, int hy0, uint8_t *hY,
                                             uint8_t *hCb, uint8_t *hCr, int t, const HaloRegs<VS> &h) {
    if (t < 0 || t >= 2 * HS * 8) return;
    const int ix = t % (2 * HS), iy = t / (2 * HS);
    uint32_t yw[VS], cbs[4] = {0, 0, 0, 0}, crs[4] = {0, 0, 0, 0};
    if (h.full) {
#pragma unroll
        for (int r = 0; r < VS; r++) {
            uint32_t cb[4], cr[4];
            convert4(h.d[r][0], h.d[r][1], h.d[r][2], yw[r], cb, cr);
#pragma unroll
            for (int q = 0; q < 4; q++) { cbs[q] += cb[q]; crs[q] += cr[q]; }
        }
    } else {
        convert_item_edge<VS>(src, g, hx0 + 4 * ix, hy0 + iy * VS, yw, cbs, crs);
    }
#pragma unroll
    for (int r = 0; r < VS; r++) *reinterpret_cast<uint32_t *>(&hY[(iy * VS + r) * 8 * HS + 4 * ix]) = yw[r];
    store_chroma<HS, VS, 8>(hCb, hCr, iy, ix, cbs, crs);
}

// The DC coefficients of the last luma block and of the chroma blocks of the MCU whose planes
// halo_convert left in hY / hCb / hCr, computed by lanes 0..2 of the calling wave into dc[0..2].
template <int HS, int VS, int ALGO>
__device__ __forceinline__ void halo_dc(const uint8_t *hY, const uint8_t *hCb, const uint8_t *hCr, int lane, int16_t *dc) {
    if (lane >= 3) return;
    const uint8_t *pl = lane == 0 ? hY + (8 * (VS - 1)) * 8 * HS + 8 * (HS - 1) : (lane == 1 ? hCb : hCr);
    const int pitch = lane == 0 ? 8 * HS : 8;
    uint32_t px[16];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const uint2 v = *reinterpret_cast<const uint2 *>(&pl[r * pitch]);
        px[2 * r] = v.x;
        px[2 * r + 1] = v.y;
    }
    dc[lane] = (int16_t)(lane ? transform_dc<ALGO, true>(px) : transform_dc<ALGO, false>(px));
}

// The same three DC coefficients for the separable real-dct, from sums alone: the DC path of that transform only
// adds the 64 samples (integers, exact in f32 in any order) before the level shift, the scale and the division
// (dc_real), so the sum of the block's bytes gives the identical value.  Lanes 16b..16b+15 each take one dword of
// block b (0 last luma block, 1 Cb, 2 Cr) and add their sums within their row of 16 lanes; dc[0..2] as halo_dc.
template <int HS, int VS>
__device__ __forceinline__ void halo_dc_sums(const uint8_t *hY, const uint8_t *hCb, const uint8_t *hCr, int lane, int16_t *dc) {
    const int b = lane >> 4, j = lane & 15, r = j >> 1, half = j & 1;
    const uint8_t *p = b == 0 ? hY + (8 * (VS - 1) + r) * 8 * HS + 8 * (HS - 1) + 4 * half
                              : (b == 1 ? hCb : hCr) + r * 8 + 4 * half;
    const uint32_t w = lane < 48 ? *reinterpret_cast<const uint32_t *>(p) : 0u;
    uint32_t sum = __builtin_amdgcn_sad_u8(w, 0u, 0u);                  // the four bytes added up
    sum += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sum, 0x111, 0xf, 0xf, false);  // row_shr:1
    sum += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sum, 0x112, 0xf, 0xf, false);  // row_shr:2
    sum += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sum, 0x114, 0xf, 0xf, false);  // row_shr:4
    sum += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sum, 0x118, 0xf, 0xf, false);  // row_shr:8
    if (j == 15 && b < 3) {
        const float s = dct_scale(0, 0) * ((float)sum - 8192.0f);
        const float q = b == 0 ? (float)quant<false>(0) : (float)quant<true>(0);
        const float z = b == 0 ? rcp_rn(quant<false>(0)) : rcp_rn(quant<true>(0));
        dc[b] = (int16_t)(int)div_const(s, q, z);
    }
}

// One thread per block: samples from the planes at the start of `lds` into registers, then (the
// planes' space being reused) the finished blocks to lds[bi * OUT_STRIDE], bi = position of the block
// in the scan order of the tile.  The last wave also runs spare(); except at 4:2:2 it has no block to transform.
// Contains one barrier; the caller places one after it.
// ROWS_ONCE (real-dct): the first pass, which is the same for luma and chroma blocks, is written once ahead of the
// branch on the component, with the samples kept in f32 (rows_real<true>).
template <int HS, int VS, int ALGO, bool ROWS_ONCE = false, class Spare>
__device__ __forceinline__ void tile_transform(uint8_t *lds, int mcus, int tid, Spare spare) {
    using C = TileCfg<HS, VS>;
    const uint8_t *sY = lds;
    const uint8_t *sCb = lds + C::TILE_H * C::TW;
    const uint8_t *sCr = sCb + 8 * C::CW;
    const int wave = tid >> 6, lane = tid & 63;
    // Every thread loads eight rows of samples, also the threads without a block (from a block that exists): with
    // the loads under a condition the compiler fills the 16 registers with zeros on the other path, 32-48
    // instructions per wave.
    uint32_t px[16];
    int bi = -1;                                               // block index inside the tile, scan order
    if (wave < C::YWAVES) {
        // luma: block row `brow`, block column `bcol` of the tile
        const int brow = wave / C::ROW_WAVES, bcol = (wave % C::ROW_WAVES) * 64 + lane;
        const int m = bcol / HS;                               // MCU inside the tile
        const int col = m < mcus ? bcol : 0;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint2 v = *reinterpret_cast<const uint2 *>(&sY[(brow * 8 + r) * C::TW + col * 8]);
            px[2 * r] = v.x;
            px[2 * r + 1] = v.y;
        }
        if (m < mcus) bi = m * C::NBLK + brow * HS + (bcol % HS);
    } else {
        const int ci = (wave - C::YWAVES) * 64 + lane;         // 0 .. 2*MCUS-1; beyond that: the wave without blocks
        const int comp = ci / C::MCUS;                         // 0 Cb, 1 Cr
        const int m = ci % C::MCUS;
        const bool has = wave < C::YWAVES + C::CWAVES && m < mcus;
        const uint8_t *pl = comp == 1 ? sCr : sCb;
        const int col = has ? m : 0;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint2 v = *reinterpret_cast<const uint2 *>(&pl[r * C::CW + col * 8]);
            px[2 * r] = v.x;
            px[2 * r + 1] = v.y;
        }
        if (has) bi = m * C::NBLK + HS * VS + comp;
    }
    __syncthreads();   // every sample is in registers: the planes' space becomes the output staging area
    if (bi >= 0) {
        uint32_t *slot = reinterpret_cast<uint32_t *>(&lds[bi * OUT_STRIDE]);
        if constexpr (ALGO == ALGO_REAL && ROWS_ONCE) {
            float t[8][8];
            rows_real<true>(px, t);
            if (wave < C::YWAVES) columns_real<false>(t, slot);
            else columns_real<true>(t, slot);
        } else {
            if (wave < C::YWAVES) transform_block<ALGO, false>(px, slot);
            else transform_block<ALGO, true>(px, slot);
        }
    }
    if (wave == C::SPARE_WAVE) spare();
}

}  // namespace jpg
