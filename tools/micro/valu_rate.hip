// Micro-benchmark: sustained VALU issue rate of a SIMD for a few instruction kinds, as a function
// of the number of resident waves.  Prints cycles per wave-instruction per SIMD (from s_memtime).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP16(X) X X X X X X X X X X X X X X X X

template <int MODE>
__global__ void k(float *out, unsigned long long *cyc, int iters, float seed) {
    float a[16];
    f2 p[16];
    for (int i = 0; i < 16; i++) { a[i] = seed * (i + 1) + threadIdx.x; p[i] = (f2){a[i], a[i] + 1.0f}; }
    unsigned u = __float_as_uint(seed) + threadIdx.x;
    const float c0 = seed * 0.5f, c1 = seed + 0.25f;
    const f2 pc0 = (f2){c0, c0}, pc1 = (f2){c1, c1};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c0), "v"(c1));
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pc0), "v"(pc1));
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c0));
        } else if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c0));
        } else if (MODE == 4) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(a[i]) : "v"(u));
        } else if (MODE == 5) {
#pragma unroll
            for (int i = 0; i < 16; i++) { unsigned r; asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(a[i])); u ^= r; }
        } else if (MODE == 6) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_lshl_or_b32 %0, %0, 8, %1" : "+v"(u) : "v"(u));
        } else if (MODE == 7) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc0));
        } else if (MODE == 8) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc0));
        } else if (MODE == 9) {   // mix: mul, add, fma alternating (like the colour code)
#pragma unroll
            for (int i = 0; i < 16; i += 4) {
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c0));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i + 1]) : "v"(c0));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i + 2]) : "v"(c0), "v"(c1));
                asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i + 3]) : "v"(c1));
            }
        } else if (MODE == 10) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u) : "v"(u));
        } else if (MODE == 12) {   // literal constant in the instruction (8-byte VOP2)
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_mul_f32 %0, 0x3e991687, %0" : "+v"(a[i]));
        } else if (MODE == 13) {   // constants in SGPRs
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "s"(c0));
        } else if (MODE == 14) {   // VOP3 fma with a literal (12 bytes)
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_fmamk_f32 %0, %0, 0x3e991687, %1" : "+v"(a[i]) : "v"(c1));
        } else if (MODE == 15) {   // VOP2 fmac (4 bytes)
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(c0), "v"(c1));
        } else if (MODE == 16) {   // VOP2 fmaak: d = a * b + literal
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3e991687" : "+v"(a[i]) : "v"(c0));
        } else if (MODE == 17) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(c0), "v"(c1));
        } else if (MODE == 18) {   // sdwa byte conversion
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1" : "=v"(a[i]) : "v"(u));
        } else if (MODE == 19) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(a[i]) : "v"(u));
        } else if (MODE == 20) {
#pragma unroll
            for (int i = 0; i < 16; i++) { unsigned r; asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(a[i])); u += r; }
        } else if (MODE == 21) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(u) : "v"(u));
        } else if (MODE == 22) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_lshlrev_b32 %0, 8, %0" : "+v"(u));
        } else if (MODE == 23) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_or_b32 %0, %0, %1" : "+v"(u) : "v"(u));
        } else if (MODE == 24) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_cvt_pk_i16_i32 %0, %0, %1" : "+v"(u) : "v"(u));
        } else if (MODE == 25) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(u) : "v"(u));
        } else if (MODE == 11) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_trunc_f32 %0, %1" : "=v"(a[i]) : "v"(c1));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 16; i++) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)u;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char *name, int waves_per_simd) {
    const int iters = 4096;
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int threads = 256;                         // 4 waves: one per SIMD
    const int grid = cus * waves_per_simd;
    float *out; unsigned long long *cyc;
    hipMalloc(&out, (size_t)grid * threads * 4);
    hipMalloc(&cyc, (size_t)grid * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<grid, threads>>>(out, cyc, 16, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<grid, threads>>>(out, cyc, iters, 1.0f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid);
    hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += (double)v; avg /= grid;
    const double instr_per_wave = (double)iters * 16;
    // wall-clock: instructions issued per SIMD / time
    const double per_simd = instr_per_wave * waves_per_simd;
    printf("%-16s waves/SIMD %d: %.2f ns per wave-instr per SIMD (wall), %.2f counter ticks per instr per wave\n", name, waves_per_simd,
           ms * 1e6 / per_simd, avg / instr_per_wave);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int w : {5}) {
        run<3>("v_mul_f32 vgpr", w);
        run<12>("v_mul_f32 literal", w);
        run<13>("v_mul_f32 sgpr", w);
        run<0>("v_fma_f32 vgpr", w);
        run<14>("v_fmamk_f32", w);
        run<17>("v_fma_f32 sgpr", w);
        run<15>("v_fmac_f32", w);
        run<16>("v_fmaak_f32", w);
        run<4>("v_cvt_f32_ubyte1", w);
        run<18>("v_cvt_f32_u32_sdwa", w);
        run<19>("v_cvt_f32_i32", w);
        run<20>("v_cvt_i32_f32+add", w);
        run<10>("v_add_u32", w);
        run<21>("v_and_or_b32", w);
        run<22>("v_lshlrev_b32", w);
        run<23>("v_or_b32", w);
        run<24>("v_cvt_pk_i16_i32", w);
        run<25>("v_perm_b32", w);
        run<6>("v_lshl_or_b32", w);
        run<11>("v_trunc_f32", w);
    }
    for (int w : {0}) {
        if (w == 0) break;
        run<0>("v_fma_f32", w);
        run<1>("v_pk_fma_f32", w);
        run<2>("v_add_f32", w);
        run<3>("v_mul_f32", w);
        run<7>("v_pk_add_f32", w);
        run<8>("v_pk_mul_f32", w);
        run<4>("v_cvt_f32_ubyte1", w);
        run<5>("v_cvt_u32_f32+xor", w);
        run<6>("v_lshl_or_b32", w);
        run<9>("mul/add/fma/sub", w);
        run<10>("v_add_u32", w);
        run<11>("v_trunc_f32", w);
    }
    return 0;
}
