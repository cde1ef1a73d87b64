// Staged path: one kernel per reference method, materialising the same
// intermediate planes the reference holds (src/jpeg/jpeg_image.rs:14-19), so
// every stage can be compared with the oracle on its own.  The fused kernel in
// front.hip is the fast path; these share its arithmetic (transform.hpp).
#include "internal.hpp"
#include "transform.hpp"

namespace jpg {

// from_bmp's conversion loop (src/jpeg/jpeg_image.rs:121-134): one thread per
// sample of the padded Wp x Hp planes; samples outside W x H are 0.
__global__ void k_color(const uint8_t *__restrict__ src, Geom g, uint8_t *__restrict__ py, uint8_t *__restrict__ pcb,
                        uint8_t *__restrict__ pcr) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)g.Wp * g.Hp) return;
    const int x = (int)(i % g.Wp), y = (int)(i / g.Wp);
    uint32_t yy = 0, cb = 0, cr = 0;
    if (x < g.W && y < g.H) {
        const uint8_t *p = src + (size_t)(g.bottom_up ? g.H - 1 - y : y) * g.stride + (size_t)3 * x;
        rgb_to_ycbcr((float)p[2], (float)p[1], (float)p[0], yy, cb, cr);
    }
    py[i] = (uint8_t)yy;
    pcb[i] = (uint8_t)cb;
    pcr[i] = (uint8_t)cr;
}

// chrominance_downsampling (src/jpeg/sampling.rs:46-102): floor of the mean of
// each h x v cell of the padded plane.
__global__ void k_downsample(const uint8_t *__restrict__ cb, const uint8_t *__restrict__ cr, Geom g,
                             uint8_t *__restrict__ ocb, uint8_t *__restrict__ ocr) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)g.Wc * g.Hc) return;
    const int x = (int)(i % g.Wc), y = (int)(i / g.Wc);
    uint32_t sb = 0, sr = 0;
    for (int r = 0; r < g.v; r++)
        for (int c = 0; c < g.h; c++) {
            const size_t j = (size_t)(y * g.v + r) * g.Wp + (size_t)(x * g.h + c);
            sb += cb[j];
            sr += cr[j];
        }
    ocb[i] = (uint8_t)(sb / (uint32_t)(g.h * g.v));
    ocr[i] = (uint8_t)(sr / (uint32_t)(g.h * g.v));
}

// dct_and_quantization (src/jpeg/dct_quant.rs:15-61) from the u8 planes into
// the scan layout.  One thread per block; 64 threads per workgroup, all of one
// component class so the quantisation table is a compile-time constant.
template <int ALGO, bool CHROMA>
__global__ __launch_bounds__(64) void k_dct_planes(const uint8_t *__restrict__ p0, const uint8_t *__restrict__ p1,
                                                   Geom g, int16_t *__restrict__ coef) {
    __shared__ __attribute__((aligned(16))) uint8_t sOut[64 * 144];
    const int bw = CHROMA ? g.Wc / 8 : g.Wp / 8;   // blocks per plane row
    const int bh = CHROMA ? g.Hc / 8 : g.Hp / 8;
    const long long per_comp = (long long)bw * bh;
    const long long idx = (long long)blockIdx.x * 64 + threadIdx.x;
    const long long total = CHROMA ? 2 * per_comp : per_comp;
    long long out_block = -1;
    if (idx < total) {
        const int comp = (int)(idx / per_comp);     // chroma: 0 Cb, 1 Cr
        const long long b = idx % per_comp;
        const int bx = (int)(b % bw), by = (int)(b / bw);
        const uint8_t *plane = (CHROMA && comp) ? p1 : p0;
        const int pw = CHROMA ? g.Wc : g.Wp;
        uint32_t px[16];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint2 d = *reinterpret_cast<const uint2 *>(plane + (size_t)(by * 8 + r) * pw + (size_t)bx * 8);
            px[2 * r] = d.x;
            px[2 * r + 1] = d.y;
        }
        transform_block<ALGO, CHROMA>(px, reinterpret_cast<uint32_t *>(&sOut[threadIdx.x * 144]));
        if (CHROMA) out_block = ((long long)by * g.mcux + bx) * g.nb + g.h * g.v + comp;
        else out_block = ((long long)(by / g.v) * g.mcux + bx / g.h) * g.nb + (by % g.v) * g.h + (bx % g.h);
    }
    __syncthreads();
    if (out_block >= 0) {
        uint4 *dst = reinterpret_cast<uint4 *>(coef + out_block * 64);
        const uint4 *s4 = reinterpret_cast<const uint4 *>(&sOut[threadIdx.x * 144]);
#pragma unroll
        for (int i = 0; i < 8; i++) dst[i] = s4[i];
    }
}

// scan layout -> the reference's row-major coefficient plane of one component
// (in-place cell order of src/pixel_matrix/block_iterator.rs:78-90)
__global__ void k_unscan(const int16_t *__restrict__ coef, Geom g, int comp, int16_t *__restrict__ plane) {
    const int pw = comp ? g.Wc : g.Wp, ph = comp ? g.Hc : g.Hp;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)pw * ph) return;
    const int c = (int)(i % pw), r = (int)(i / pw);
    const int bx = c >> 3, by = r >> 3;
    long long b;
    if (comp) b = ((long long)by * g.mcux + bx) * g.nb + g.h * g.v + (comp - 1);
    else b = ((long long)(by / g.v) * g.mcux + bx / g.h) * g.nb + (by % g.v) * g.h + (bx % g.h);
    plane[i] = coef[b * 64 + QUANT_DEV.zz[(r & 7) * 8 + (c & 7)]];
}

// DC of the last Y / Cb / Cr block in scan order (the predictors a following
// stripe starts from, src/jpeg/entropy_coding.rs:149-150)
__global__ void k_last_dc(const int16_t *__restrict__ coef, Geom g, int16_t *__restrict__ dc3) {
    if (threadIdx.x < 3) {
        const long long last_mcu = (long long)g.mcux * g.mcuy - 1;
        const int j = threadIdx.x == 0 ? g.h * g.v - 1 : g.h * g.v + threadIdx.x - 1;
        dc3[threadIdx.x] = coef[(last_mcu * g.nb + j) * 64];
    }
}

hipError_t launch_color(const Geom &g, const uint8_t *src, uint8_t *y, uint8_t *cb, uint8_t *cr, hipStream_t s) {
    const size_t n = (size_t)g.Wp * g.Hp;
    hipLaunchKernelGGL(k_color, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, g, y, cb, cr);
    return hipGetLastError();
}

hipError_t launch_downsample(const Geom &g, const uint8_t *cb, const uint8_t *cr, uint8_t *cb_out, uint8_t *cr_out,
                             hipStream_t s) {
    const size_t n = (size_t)g.Wc * g.Hc;
    hipLaunchKernelGGL(k_downsample, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, cb, cr, g, cb_out, cr_out);
    return hipGetLastError();
}

template <int ALGO>
static hipError_t launch_dct_algo(const Geom &g, const uint8_t *y, const uint8_t *cb, const uint8_t *cr, int16_t *coef,
                                  hipStream_t s) {
    const long long ny = (long long)(g.Wp / 8) * (g.Hp / 8), nc = 2LL * (g.Wc / 8) * (g.Hc / 8);
    hipLaunchKernelGGL((k_dct_planes<ALGO, false>), dim3((unsigned)((ny + 63) / 64)), dim3(64), 0, s, y, y, g, coef);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_dct_planes<ALGO, true>), dim3((unsigned)((nc + 63) / 64)), dim3(64), 0, s, cb, cr, g, coef);
    return hipGetLastError();
}

hipError_t launch_dct_planes(const Geom &g, int algo, const uint8_t *y, const uint8_t *cb, const uint8_t *cr,
                             int16_t *coef, hipStream_t s) {
    switch (algo) {
        case ALGO_REAL: return launch_dct_algo<ALGO_REAL>(g, y, cb, cr, coef, s);
        case ALGO_BIN: return launch_dct_algo<ALGO_BIN>(g, y, cb, cr, coef, s);
        case ALGO_REAL_STRICT: return launch_dct_algo<ALGO_REAL_STRICT>(g, y, cb, cr, coef, s);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_unscan(const Geom &g, const int16_t *coef, int comp, int16_t *plane, hipStream_t s) {
    const size_t n = comp ? (size_t)g.Wc * g.Hc : (size_t)g.Wp * g.Hp;
    hipLaunchKernelGGL(k_unscan, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, coef, g, comp, plane);
    return hipGetLastError();
}

hipError_t launch_last_dc(const Geom &g, const int16_t *coef, int16_t *dc3, hipStream_t s) {
    hipLaunchKernelGGL(k_last_dc, dim3(1), dim3(64), 0, s, coef, g, dc3);
    return hipGetLastError();
}

}  // namespace jpg
