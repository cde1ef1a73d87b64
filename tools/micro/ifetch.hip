// Micro-benchmark: does instruction fetch limit long straight-line code on gfx950?  Every workgroup (256 threads,
// four per CU through their LDS size, i.e. four waves per SIMD, like the encoder's coding kernel) runs once through
// 4096 independent vector instructions without a branch; the grid holds 20 rounds of workgroups.  Variants: 4-byte
// encodings, 8-byte encodings (32-bit literal or VOP3), and the same instructions in a loop of 256 that stays in
// the instruction cache line buffers.  Prints shader cycles per wave-instruction per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/ifetch.hip -o tools/micro/ifetch && tools/micro/ifetch
#include <hip/hip_runtime.h>
#include <cstdio>

#define R4(X) X X X X
#define R16(X) R4(R4(X))
#define R256(X) R16(R16(X))
#define R4096(X) R16(R256(X))

#define ROT16(T) T(0) T(1) T(2) T(3) T(4) T(5) T(6) T(7) T(8) T(9) T(10) T(11) T(12) T(13) T(14) T(15)

enum { MUL4, MUL8, FMA8, ADD8, CVT4, CVT8 };

template <int MODE>
__device__ __forceinline__ void sixteen(float (&r)[16], float a, float b) {
#define T(i) \
    if (MODE == MUL4) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a)); \
    else if (MODE == MUL8) asm volatile("v_mul_f32 %0, 0x3e991687, %0" : "+v"(r[i])); \
    else if (MODE == FMA8) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b)); \
    else if (MODE == ADD8) asm volatile("v_add_f32 %0, 0x4b000000, %0" : "+v"(r[i])); \
    else if (MODE == CVT4) asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(r[i])); \
    else asm volatile("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(r[i]) : "v"(a));
    ROT16(T)
#undef T
}

template <int MODE, bool LOOP>
__global__ __launch_bounds__(256) void k(float *out, float a, float b, int rounds) {
    __shared__ float pad[38 * 256];              // 38 KB: four workgroups per CU
    float r[16];
#pragma unroll
    for (int i = 0; i < 16; i++) r[i] = a * (float)(threadIdx.x + i);
    if (LOOP) {
        for (int it = 0; it < 16; it++) { R16(sixteen<MODE>(r, a, b);) }
    } else {
        R256(sixteen<MODE>(r, a, b);)
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += r[i];
    if (s == 12345.678f) { pad[threadIdx.x] = s; out[blockIdx.x] = pad[(threadIdx.x + 1) & 255]; }
}

template <int MODE, bool LOOP>
static void run(const char *name) {
    float *out;
    hipMalloc(&out, 1 << 20);
    int dev = 0, cus = 0, khz = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, dev);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best[2] = {1e30f, 1e30f};
    for (int rep = 0; rep < 5; rep++)
        for (int v = 0; v < 2; v++) {
            const int rounds = v ? 40 : 20;
            hipEventRecord(e0);
            hipLaunchKernelGGL((k<MODE, LOOP>), dim3(cus * 4 * rounds), dim3(256), 0, 0, out, 1.0f, 0.5f, rounds);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best[v]) best[v] = ms;
        }
    // 20 more rounds of 4 workgroups per CU = 20 * 4 waves per SIMD, 4096 instructions each
    const double cycles = (best[1] - best[0]) * 1e-3 * khz * 1e3;
    printf("%-28s %6.3f cycles per instruction and SIMD (%.3f ms for 20 rounds, clock %d MHz)\n", name,
           cycles / (20.0 * 4 * 4096), best[1] - best[0], khz / 1000);
    hipFree(out);
}

int main() {
    run<MUL4, true>("v_mul_f32 4B loop");
    run<MUL4, false>("v_mul_f32 4B straight");
    run<MUL8, true>("v_mul_f32 literal 8B loop");
    run<MUL8, false>("v_mul_f32 literal 8B straight");
    run<FMA8, false>("v_fma_f32 VOP3 8B straight");
    run<ADD8, false>("v_add_f32 literal 8B straight");
    run<CVT4, true>("v_cvt_u32_f32 4B loop");
    run<CVT4, false>("v_cvt_u32_f32 4B straight");
    run<CVT8, false>("v_cvt_u32_f32_sdwa 8B straight");
    return 0;
}
