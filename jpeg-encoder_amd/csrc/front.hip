// Fused front end: BMP rows (B,G,R bytes) -> YCbCr -> chroma box average ->
// level shift -> 8x8 DCT -> quantisation -> zigzag, one kernel, no plane ever
// written to memory.  Replaces the loop of JpegImage::from_bmp
// (src/jpeg/jpeg_image.rs:121-134), chrominance_downsampling
// (src/jpeg/sampling.rs:46-102) and dct_and_quantization
// (src/jpeg/dct_quant.rs:15-61) of the reference.
//
// One workgroup owns a tile of 512 (4:2:2: 1024) pixels x one MCU row (8*v image rows): it
// converts and transforms the tile (tile.hpp); the tile's blocks, contiguous in
// the scan layout, then go to global memory as coalesced 16-byte stores.
// HBM traffic is the algorithmic minimum: every input byte is read once, every
// coefficient written once.
#include <hip/hip_ext.h>

#include "tile.hpp"

namespace jpg {

// A batch of equally sized images is one launch: srcs[i] (when given) is the pixel array of image
// i, whose tiles follow those of image i-1 in the grid and whose blocks follow in `coef`.
template <int HS, int VS, int ALGO>
__device__ __forceinline__ void front_tile(const uint8_t *__restrict__ src, const uint8_t *const *__restrict__ srcs,
                                           const Geom &g, int16_t *__restrict__ coef) {
    using C = TileCfg<HS, VS>;
    // Y plane, then Cb, Cr planes of the tile; phase B reuses the space for the finished blocks
    __shared__ __attribute__((aligned(16))) uint8_t lds[C::LDS_BYTES];
    uint8_t *sY = lds;
    uint8_t *sCb = lds + C::TILE_H * C::TW;
    uint8_t *sCr = sCb + 8 * C::CW;

    // grid: x = tile column, y = MCU row, z = image of a batch (no divisions to find the tile)
    const int tile_x = blockIdx.x, tile_y = blockIdx.y;
    if (srcs) {
        src = srcs[blockIdx.z];
        coef += (size_t)blockIdx.z * (size_t)g.nblocks * 64;
    }
    const int mcu0 = tile_x * C::MCUS;                        // first MCU column of the tile
    const int mcus = min(C::MCUS, g.mcux - mcu0);             // MCUs really present
    const int x0 = mcu0 * 8 * HS;
    const int y0 = tile_y * C::TILE_H;
    const int tid = threadIdx.x;

    tile_convert<HS, VS>(src, g, x0, y0, mcus, sY, sCb, sCr, tid);
    __syncthreads();
    tile_transform<HS, VS, ALGO>(lds, mcus, tid, [] {});
    __syncthreads();

    // ---------------- phase C: coalesced copy-out ---------------------------
    const size_t first_block = ((size_t)tile_y * g.mcux + mcu0) * C::NBLK;
    uint4 *dst = reinterpret_cast<uint4 *>(coef + first_block * 64);
    const int units = mcus * C::NBLK * 8;                      // 16-byte pieces
    for (int u = tid; u < units; u += THREADS)
        dst[u] = *reinterpret_cast<const uint4 *>(&lds[(u >> 3) * OUT_STRIDE + (u & 7) * 16]);
}

template <int HS, int VS, int ALGO>
__global__ __launch_bounds__(THREADS) void k_front(const uint8_t *__restrict__ src, const uint8_t *const *__restrict__ srcs,
                                                   Geom g, int16_t *__restrict__ coef) {
    front_tile<HS, VS, ALGO>(src, srcs, g, coef);
}

// The same kernel held to 128 registers (a fourth wave per SIMD) for the integer transform, which wants 130.
template <int HS, int VS, int ALGO>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_front_4waves(
    const uint8_t *__restrict__ src, const uint8_t *const *__restrict__ srcs, Geom g, int16_t *__restrict__ coef) {
    front_tile<HS, VS, ALGO>(src, srcs, g, coef);
}

// e0/e1 (both or neither): events that receive the kernel's own start and end time stamps
template <int HS, int VS>
static hipError_t launch_front_hv(const Geom &g, int algo, const uint8_t *src, const uint8_t *const *srcs, int nimg,
                                  int16_t *coef, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    using C = TileCfg<HS, VS>;
    const int tiles_x = (g.mcux + C::MCUS - 1) / C::MCUS;
    const long long tiles = (long long)tiles_x * g.mcuy * nimg;
    if (tiles <= 0 || tiles > 0x7fffffffLL || g.mcuy > 65535 || nimg > 65535) return hipErrorInvalidValue;
    const dim3 grid((unsigned)tiles_x, (unsigned)g.mcuy, (unsigned)nimg), block(THREADS);
    switch (algo) {
        case ALGO_REAL: hipExtLaunchKernelGGL((k_front<HS, VS, ALGO_REAL>), grid, block, 0, s, e0, e1, 0, src, srcs, g, coef); break;
        case ALGO_BIN: hipExtLaunchKernelGGL((k_front_4waves<HS, VS, ALGO_BIN>), grid, block, 0, s, e0, e1, 0, src, srcs, g, coef); break;
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
