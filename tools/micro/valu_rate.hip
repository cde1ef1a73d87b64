// Micro-benchmark: sustained VALU issue rate of a SIMD for a few instruction kinds, as a function
// of the number of resident waves.  Prints cycles per wave-instruction per SIMD (from s_memtime).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP16(X) X X X X X X X X X X X X X X X X

template <int MODE>
__global__ void k(float *out, unsigned long long *cyc, unsigned long long *real, int iters, float seed) {
    extern __shared__ float pin[];   // dynamic LDS sized so that exactly `waves_per_simd` workgroups fit a CU
    float a[16];
    f2 p[16];
    for (int i = 0; i < 16; i++) { a[i] = seed * (i + 1) + threadIdx.x; p[i] = (f2){a[i], a[i] + 1.0f}; }
    unsigned u = __float_as_uint(seed) + threadIdx.x;
    const float c0 = seed * 0.5f, c1 = seed + 0.25f;
    const f2 pc0 = (f2){c0, c0}, pc1 = (f2){c1, c1};
    __syncthreads();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it += 16) {
#pragma unroll
      for (int rep_ = 0; rep_ < 16; rep_++) {   // 256 instructions between taken branches
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
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 16; i++) s += a[i] + p[i].x + p[i].y;
    if (seed == 12345.0f) pin[threadIdx.x] = s;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)u;
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; real[blockIdx.x] = r1 - r0; }
}

template <int MODE>
void run(const char *name, int waves_per_simd) {
    const int iters = 16384;
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int threads = 256;                         // 4 waves: one per SIMD
    const int grid = cus * waves_per_simd;
    float *out; unsigned long long *cyc, *real;
    hipMalloc(&out, (size_t)grid * threads * 4);
    hipMalloc(&cyc, (size_t)grid * 8);
    hipMalloc(&real, (size_t)grid * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds = (size_t)(160 * 1024 / waves_per_simd) - 512;
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    k<MODE><<<grid, threads, lds>>>(out, cyc, real, 16, 1.0f);
    hipDeviceSynchronize();
    float ms1 = 0, ms = 0;
    for (int rep = 0; rep < 2; rep++) {   // the same launch at N and 2N iterations: the difference has no fixed costs in it
        hipEventRecord(e0);
        k<MODE><<<grid, threads, lds>>>(out, cyc, real, iters * (rep + 1), 1.0f);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float t; hipEventElapsedTime(&t, e0, e1);
        if (rep == 0) ms1 = t; else ms = t - ms1;
    }
    std::vector<unsigned long long> h(grid);
    hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
    std::vector<unsigned long long> hr(grid);
    hipMemcpy(hr.data(), real, grid * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += (double)v; avg /= grid;
    double avr = 0; for (auto v : hr) avr += (double)v; avr /= grid;
    const double ghz = avg / avr * 0.1;   // s_memrealtime ticks at 100 MHz
    const double instr_per_wave = (double)iters * 16;
    avg /= 2; avr /= 2;   // the counters are those of the 2N launch
    // wall-clock: instructions issued per SIMD / time
    const double per_simd = instr_per_wave * waves_per_simd;
    printf("%-20s waves/SIMD %d: %.3f ns per wave-instr per SIMD (wall), %.2f cycles per instr per SIMD (in-kernel: %.2f per wave), clock %.2f GHz, N-launch %.3f ms\n",
           name, waves_per_simd, ms * 1e6 / per_simd, avg / instr_per_wave / waves_per_simd, avg / instr_per_wave, ghz, ms1);
    hipFree(out); hipFree(cyc); hipFree(real);
}

int main() {
    for (int w : {1, 2, 3, 4, 5, 8}) {
        run<0>("v_fma_f32", w);
        run<2>("v_add_f32", w);
        run<3>("v_mul_f32", w);
        run<9>("mul/add/fma/sub mix", w);
        run<1>("v_pk_fma_f32", w);
        run<7>("v_pk_add_f32", w);
    }
    for (int w : {4}) {
        run<12>("v_mul_f32 literal", w);
        run<14>("v_fmamk_f32", w);
        run<15>("v_fmac_f32", w);
        run<16>("v_fmaak_f32", w);
        run<4>("v_cvt_f32_ubyte1", w);
        run<18>("v_cvt_f32_u32_sdwa", w);
        run<19>("v_cvt_f32_i32", w);
        run<20>("v_cvt_i32_f32+add", w);
        run<5>("v_cvt_u32_f32+xor", w);
        run<10>("v_add_u32", w);
        run<21>("v_and_or_b32", w);
        run<22>("v_lshlrev_b32", w);
        run<23>("v_or_b32", w);
        run<24>("v_cvt_pk_i16_i32", w);
        run<25>("v_perm_b32", w);
        run<6>("v_lshl_or_b32", w);
        run<11>("v_trunc_f32", w);
    }
    return 0;
}
