// Fused front end: BMP rows (B,G,R bytes) -> YCbCr -> chroma box average ->
// level shift -> 8x8 DCT -> quantisation -> zigzag, one kernel, no plane ever
// written to memory.  Replaces the loop of JpegImage::from_bmp
// (src/jpeg/jpeg_image.rs:121-134), chrominance_downsampling
// (src/jpeg/sampling.rs:46-102) and dct_and_quantization
// (src/jpeg/dct_quant.rs:15-61) of the reference.
//
// One workgroup owns a tile of 512 pixels x one MCU row (8*v image rows):
//   phase A  every thread converts groups of 4 pixels x v rows read straight
//            from global memory (12 contiguous bytes per row and lane, so a
//            wave reads 768 contiguous bytes) and leaves Y and the averaged
//            Cb/Cr as bytes in LDS;
//   phase B  one thread per 8x8 block (wave-uniform component class, so the
//            quantisation table is a compile-time constant) transforms it in
//            registers and parks the 128-byte zigzag block in LDS;
//   phase C  the tile's blocks, contiguous in the scan layout, go to global
//            memory as coalesced 16-byte stores.
// HBM traffic is the algorithmic minimum: every input byte is read once, every
// coefficient written once.
#include <hip/hip_ext.h>

#include "internal.hpp"
#include "transform.hpp"

namespace jpg {

constexpr int TILE_W = 512;     // pixels per tile row
constexpr int OUT_STRIDE = 144; // bytes between parked blocks in LDS (128 + 16: conflict-free b128 access)
constexpr int THREADS = 256;

template <int HS, int VS>
struct TileCfg {
    static constexpr int MCUS = TILE_W / (8 * HS);           // MCUs per tile
    static constexpr int NBLK = HS * VS + 2;                  // blocks per MCU
    static constexpr int TILE_H = 8 * VS;                     // image rows per tile
    static constexpr int YWAVES = VS;                         // 64 luma blocks per block row
    static constexpr int CWAVES = (2 * MCUS) / 64;            // chroma blocks: MCUS per component
    static constexpr int CW = TILE_W / HS;                    // chroma samples per tile row
    static constexpr int BLOCKS = MCUS * NBLK;
    static constexpr int ITEMS = (TILE_W / 4) * 8;            // groups of 4 pixels x VS rows
    static constexpr int ROUNDS = ITEMS / THREADS;            // items per thread in a full tile
    static constexpr int PLANE_BYTES = TILE_H * TILE_W + 2 * 8 * CW;
    static constexpr int OUT_BYTES = BLOCKS * OUT_STRIDE;
    static constexpr int LDS_BYTES = PLANE_BYTES > OUT_BYTES ? PLANE_BYTES : OUT_BYTES;
    static_assert(YWAVES + CWAVES <= THREADS / 64, "one thread per block");
    static_assert(ITEMS % THREADS == 0, "whole rounds");
};

// 4 pixels of one row: 12 bytes B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3 -> packed Y, Cb, Cr bytes
__device__ __forceinline__ void convert4(uint32_t d0, uint32_t d1, uint32_t d2, uint32_t &y4, uint32_t (&cb)[4],
                                         uint32_t (&cr)[4]) {
    float fy[4], fb[4], fr[4];
    rgb_to_ycbcr_f((float)((d0 >> 16) & 0xff), (float)((d0 >> 8) & 0xff), (float)(d0 & 0xff), fy[0], fb[0], fr[0]);
    rgb_to_ycbcr_f((float)((d1 >> 8) & 0xff), (float)(d1 & 0xff), (float)(d0 >> 24), fy[1], fb[1], fr[1]);
    rgb_to_ycbcr_f((float)(d2 & 0xff), (float)(d1 >> 24), (float)((d1 >> 16) & 0xff), fy[2], fb[2], fr[2]);
    rgb_to_ycbcr_f((float)(d2 >> 24), (float)((d2 >> 16) & 0xff), (float)((d2 >> 8) & 0xff), fy[3], fb[3], fr[3]);
    // `as u8` (truncation; nothing saturates): the four Y bytes are converted into place
    cvt_pack8<0>(y4, fy[0]);
    cvt_pack8<1>(y4, fy[1]);
    cvt_pack8<2>(y4, fy[2]);
    cvt_pack8<3>(y4, fy[3]);
#pragma unroll
    for (int q = 0; q < 4; q++) {
        cb[q] = (uint32_t)fb[q];
        cr[q] = (uint32_t)fr[q];
    }
}

template <int HS, int VS>
__device__ __forceinline__ void store_chroma(uint8_t *sCb, uint8_t *sCr, int iy, int ix, const uint32_t (&cbs)[4],
                                             const uint32_t (&crs)[4]) {
    using C = TileCfg<HS, VS>;
    // box average, floor(sum / (h*v)) (src/jpeg/sampling.rs:66-73)
    if constexpr (HS == 2) {
        const uint32_t b0 = (cbs[0] + cbs[1]) / (2 * VS), b1 = (cbs[2] + cbs[3]) / (2 * VS);
        const uint32_t r0 = (crs[0] + crs[1]) / (2 * VS), r1 = (crs[2] + crs[3]) / (2 * VS);
        *reinterpret_cast<uint16_t *>(&sCb[iy * C::CW + 2 * ix]) = (uint16_t)(b0 | (b1 << 8));
        *reinterpret_cast<uint16_t *>(&sCr[iy * C::CW + 2 * ix]) = (uint16_t)(r0 | (r1 << 8));
    } else {
        *reinterpret_cast<uint32_t *>(&sCb[iy * C::CW + 4 * ix]) = cbs[0] | (cbs[1] << 8) | (cbs[2] << 16) | (cbs[3] << 24);
        *reinterpret_cast<uint32_t *>(&sCr[iy * C::CW + 4 * ix]) = crs[0] | (crs[1] << 8) | (crs[2] << 16) | (crs[3] << 24);
    }
}

// A batch of equally sized images is one launch: srcs[i] (when given) is the pixel array of image
// i, whose tiles follow those of image i-1 in the grid and whose blocks follow in `coef`.
template <int HS, int VS, int ALGO>
__global__ __launch_bounds__(THREADS) void k_front(const uint8_t *__restrict__ src, const uint8_t *const *__restrict__ srcs,
                                                   Geom g, int16_t *__restrict__ coef) {
    using C = TileCfg<HS, VS>;
    // Y plane, then Cb, Cr planes of the tile; phase B reuses the space for the finished blocks
    __shared__ __attribute__((aligned(16))) uint8_t lds[C::LDS_BYTES];
    uint8_t *sY = lds;
    uint8_t *sCb = lds + C::TILE_H * TILE_W;
    uint8_t *sCr = sCb + 8 * C::CW;

    const int tiles_x = (g.mcux + C::MCUS - 1) / C::MCUS;
    int tile = blockIdx.x;
    if (srcs) {
        const int per_image = tiles_x * g.mcuy;
        const int img = tile / per_image;
        tile -= img * per_image;
        src = srcs[img];
        coef += (size_t)img * (size_t)g.nblocks * 64;
    }
    const int tile_x = tile % tiles_x;
    const int tile_y = tile / tiles_x;
    const int mcu0 = tile_x * C::MCUS;                        // first MCU column of the tile
    const int mcus = min(C::MCUS, g.mcux - mcu0);             // MCUs really present
    const int x0 = mcu0 * 8 * HS;
    const int y0 = tile_y * C::TILE_H;
    const int tid = threadIdx.x;

    // ---------------- phase A: colour conversion + chroma average ----------
    if (mcus == C::MCUS && x0 + TILE_W <= g.W && y0 + C::TILE_H <= g.H) {
        // interior tile: every thread owns ROUNDS items; all loads are issued before the arithmetic
        uint32_t d[C::ROUNDS][VS][3];
        const int ix = tid & 127;
#pragma unroll
        for (int k = 0; k < C::ROUNDS; k++) {
            const int iy = 2 * k + (tid >> 7);
#pragma unroll
            for (int r = 0; r < VS; r++) {
                const int y = y0 + iy * VS + r;
                const size_t row = (size_t)(g.bottom_up ? g.H - 1 - y : y) * g.stride;
                const uint32_t *p = reinterpret_cast<const uint32_t *>(src + row + (size_t)3 * (x0 + 4 * ix));
                d[k][r][0] = p[0]; d[k][r][1] = p[1]; d[k][r][2] = p[2];
            }
        }
#pragma unroll
        for (int k = 0; k < C::ROUNDS; k++) {
            const int iy = 2 * k + (tid >> 7);
            uint32_t cbs[4] = {0, 0, 0, 0}, crs[4] = {0, 0, 0, 0};
#pragma unroll
            for (int r = 0; r < VS; r++) {
                uint32_t y4, cb[4], cr[4];
                convert4(d[k][r][0], d[k][r][1], d[k][r][2], y4, cb, cr);
                *reinterpret_cast<uint32_t *>(&sY[(iy * VS + r) * TILE_W + 4 * ix]) = y4;
#pragma unroll
                for (int q = 0; q < 4; q++) { cbs[q] += cb[q]; crs[q] += cr[q]; }
            }
            store_chroma<HS, VS>(sCb, sCr, iy, ix, cbs, crs);
        }
    } else {
        // edge tile: samples outside W x H stay 0 in all three planes
        // (zero-filled planes of src/jpeg/jpeg_image.rs:59-70)
        const int items_x = mcus * 2 * HS;
        for (int it = tid; it < items_x * 8; it += THREADS) {
            const int ix = it % items_x, iy = it / items_x;
            const int x = x0 + 4 * ix;
            uint32_t cbs[4] = {0, 0, 0, 0}, crs[4] = {0, 0, 0, 0};
#pragma unroll
            for (int r = 0; r < VS; r++) {
                const int ly = iy * VS + r, y = y0 + ly;
                uint32_t yy[4] = {0, 0, 0, 0};
                if (y < g.H) {
                    const uint8_t *p = src + (size_t)(g.bottom_up ? g.H - 1 - y : y) * g.stride;
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        if (x + q < g.W) {
                            const uint8_t *c = p + (size_t)3 * (x + q);
                            uint32_t cb, cr;
                            rgb_to_ycbcr((float)c[2], (float)c[1], (float)c[0], yy[q], cb, cr);
                            cbs[q] += cb;
                            crs[q] += cr;
                        }
                }
                *reinterpret_cast<uint32_t *>(&sY[ly * TILE_W + 4 * ix]) = yy[0] | (yy[1] << 8) | (yy[2] << 16) | (yy[3] << 24);
            }
            store_chroma<HS, VS>(sCb, sCr, iy, ix, cbs, crs);
        }
    }
    __syncthreads();

    // ---------------- phase B: one thread per 8x8 block ---------------------
    const int wave = tid >> 6, lane = tid & 63;
    uint32_t px[16];
    int bi = -1;                                               // block index inside the tile, scan order
    if (wave < C::YWAVES) {
        // luma: block row `wave`, block column `lane` of the tile
        const int m = lane / HS;                               // MCU inside the tile
        if (m < mcus) {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const uint2 v = *reinterpret_cast<const uint2 *>(&sY[(wave * 8 + r) * TILE_W + lane * 8]);
                px[2 * r] = v.x;
                px[2 * r + 1] = v.y;
            }
            bi = m * C::NBLK + wave * HS + (lane % HS);
        }
    } else if (wave < C::YWAVES + C::CWAVES) {
        const int ci = (wave - C::YWAVES) * 64 + lane;         // 0 .. 2*MCUS-1
        const int comp = ci / C::MCUS;                         // 0 Cb, 1 Cr
        const int m = ci % C::MCUS;
        if (m < mcus) {
            const uint8_t *pl = comp ? sCr : sCb;
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const uint2 v = *reinterpret_cast<const uint2 *>(&pl[r * C::CW + m * 8]);
                px[2 * r] = v.x;
                px[2 * r + 1] = v.y;
            }
            bi = m * C::NBLK + HS * VS + comp;
        }
    }
    __syncthreads();   // every sample is in registers: the planes' space becomes the output staging area
    if (bi >= 0) {
        uint32_t *slot = reinterpret_cast<uint32_t *>(&lds[bi * OUT_STRIDE]);
        if (wave < C::YWAVES) transform_block<ALGO, false>(px, slot);
        else transform_block<ALGO, true>(px, slot);
    }
    __syncthreads();

    // ---------------- phase C: coalesced copy-out ---------------------------
    const size_t first_block = ((size_t)tile_y * g.mcux + mcu0) * C::NBLK;
    uint4 *dst = reinterpret_cast<uint4 *>(coef + first_block * 64);
    const int units = mcus * C::NBLK * 8;                      // 16-byte pieces
    for (int u = tid; u < units; u += THREADS)
        dst[u] = *reinterpret_cast<const uint4 *>(&lds[(u >> 3) * OUT_STRIDE + (u & 7) * 16]);
}

// e0/e1 (both or neither): events that receive the kernel's own start and end time stamps
template <int HS, int VS>
static hipError_t launch_front_hv(const Geom &g, int algo, const uint8_t *src, const uint8_t *const *srcs, int nimg,
                                  int16_t *coef, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    using C = TileCfg<HS, VS>;
    const int tiles_x = (g.mcux + C::MCUS - 1) / C::MCUS;
    const long long tiles = (long long)tiles_x * g.mcuy * nimg;
    if (tiles <= 0 || tiles > 0x7fffffffLL) return hipErrorInvalidValue;
    const dim3 grid((unsigned)tiles), block(THREADS);
    switch (algo) {
        case ALGO_REAL: hipExtLaunchKernelGGL((k_front<HS, VS, ALGO_REAL>), grid, block, 0, s, e0, e1, 0, src, srcs, g, coef); break;
        case ALGO_BIN: hipExtLaunchKernelGGL((k_front<HS, VS, ALGO_BIN>), grid, block, 0, s, e0, e1, 0, src, srcs, g, coef); break;
        case ALGO_REAL_STRICT:
            hipExtLaunchKernelGGL((k_front<HS, VS, ALGO_REAL_STRICT>), grid, block, 0, s, e0, e1, 0, src, srcs, g, coef);
            break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_front(const Geom &g, int algo, const uint8_t *src, int16_t *coef, hipStream_t s, hipEvent_t e0,
                        hipEvent_t e1, const uint8_t *const *srcs, int nimg) {
    // the fast path reads rows as dwords (the caller checks the pointers of a batch)
    if ((!srcs && (reinterpret_cast<uintptr_t>(src) & 3)) || (g.stride & 3) || nimg < 1) return hipErrorInvalidValue;
    if (!srcs) nimg = 1;
    if (g.h == 2 && g.v == 2) return launch_front_hv<2, 2>(g, algo, src, srcs, nimg, coef, s, e0, e1);
    if (g.h == 2 && g.v == 1) return launch_front_hv<2, 1>(g, algo, src, srcs, nimg, coef, s, e0, e1);
    if (g.h == 1 && g.v == 1) return launch_front_hv<1, 1>(g, algo, src, srcs, nimg, coef, s, e0, e1);
    return hipErrorInvalidValue;
}

}  // namespace jpg
