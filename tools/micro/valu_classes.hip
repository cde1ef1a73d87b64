// Micro-benchmark: issue cost of vector-ALU instruction classes on one SIMD of gfx950 with four waves
// resident (one workgroup of 1024 threads per CU, pinned there by its LDS size): every mode is a stream of
// independent instructions (16 registers in rotation, 256 instructions between branches).  Prints shader
// cycles per wave-instruction per SIMD, from the wall-clock difference between a run of N and of 2N
// iterations (no fixed costs in it) and the shader clock measured by the kernel itself.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_classes.hip -o tools/micro/valu_classes && tools/micro/valu_classes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ROUND16(T) T(0) T(1) T(2) T(3) T(4) T(5) T(6) T(7) T(8) T(9) T(10) T(11) T(12) T(13) T(14) T(15)

// X(name, asm text): %0 = the rotating 32-bit register (in/out), %1 = the rotating 64-bit pair (in/out),
// %2 %3 = two other vgprs, %4 = an sgpr, %5 = another 64-bit pair
#define MODES(X) \
    X(v_fma_f32, "v_fma_f32 %0, %0, %2, %3") \
    X(v_fmac_f32, "v_fmac_f32 %0, %2, %3") \
    X(v_mul_f32, "v_mul_f32 %0, %0, %2") \
    X(v_mul_f32_literal, "v_mul_f32 %0, 0x3e991687, %0") \
    X(v_add_f32, "v_add_f32 %0, %0, %2") \
    X(v_sub_f32, "v_sub_f32 %0, %0, %2") \
    X(v_pk_fma_f32_half_count, "v_pk_fma_f32 %1, %1, %5, %5") \
    X(v_cvt_f32_ubyte0, "v_cvt_f32_ubyte0 %0, %0") \
    X(v_cvt_f32_ubyte1, "v_cvt_f32_ubyte1 %0, %0") \
    X(v_cvt_f32_ubyte3, "v_cvt_f32_ubyte3 %0, %0") \
    X(v_cvt_f32_u32, "v_cvt_f32_u32 %0, %0") \
    X(v_cvt_u32_f32, "v_cvt_u32_f32 %0, %0") \
    X(v_cvt_i32_f32, "v_cvt_i32_f32 %0, %0") \
    X(v_cvt_u32_f32_sdwa_byte1, "v_cvt_u32_f32_sdwa %0, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD") \
    X(v_cvt_i32_f32_sdwa_word1, "v_cvt_i32_f32_sdwa %0, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD") \
    X(v_cvt_pk_u8_f32, "v_cvt_pk_u8_f32 %0, %2, 1, %0") \
    X(v_trunc_f32, "v_trunc_f32 %0, %0") \
    X(v_floor_f32, "v_floor_f32 %0, %0") \
    X(v_rndne_f32, "v_rndne_f32 %0, %0") \
    X(v_mov_b32, "v_mov_b32 %0, %2") \
    X(v_add_u32, "v_add_u32 %0, %0, %2") \
    X(v_sub_u32, "v_sub_u32 %0, %0, %2") \
    X(v_add3_u32, "v_add3_u32 %0, %0, %2, %3") \
    X(v_and_b32, "v_and_b32 %0, %0, %2") \
    X(v_or_b32, "v_or_b32 %0, %0, %2") \
    X(v_xor_b32, "v_xor_b32 %0, %0, %2") \
    X(v_or3_b32, "v_or3_b32 %0, %0, %2, %3") \
    X(v_and_or_b32, "v_and_or_b32 %0, %0, %2, %3") \
    X(v_bfi_b32, "v_bfi_b32 %0, %2, %0, %3") \
    X(v_bfe_u32, "v_bfe_u32 %0, %0, 8, 8") \
    X(v_lshlrev_b32, "v_lshlrev_b32 %0, 3, %0") \
    X(v_lshrrev_b32, "v_lshrrev_b32 %0, 3, %0") \
    X(v_ashrrev_i32, "v_ashrrev_i32 %0, 3, %0") \
    X(v_lshl_or_b32, "v_lshl_or_b32 %0, %0, 8, %2") \
    X(v_lshl_add_u32, "v_lshl_add_u32 %0, %0, 2, %2") \
    X(v_lshlrev_b32_sdwa_byte, "v_lshlrev_b32_sdwa %0, %2, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1") \
    X(v_mul_u32_u24_sdwa_byte, "v_mul_u32_u24_sdwa %0, %2, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2") \
    X(v_mul_u32_u24, "v_mul_u32_u24 %0, %0, %2") \
    X(v_mad_u32_u24, "v_mad_u32_u24 %0, %0, %2, %3") \
    X(v_mul_lo_u32, "v_mul_lo_u32 %0, %0, %2") \
    X(v_mul_hi_u32, "v_mul_hi_u32 %0, %0, %2") \
    X(v_perm_b32, "v_perm_b32 %0, %0, %2, %3") \
    X(v_alignbit_b32, "v_alignbit_b32 %0, %0, %2, 8") \
    X(v_alignbyte_b32, "v_alignbyte_b32 %0, %0, %2, 1") \
    X(v_cmp_lt_u32, "v_cmp_lt_u32 vcc, %0, %2") \
    X(v_cmp_ge_f32, "v_cmp_ge_f32 vcc, %0, %2") \
    X(v_min_u32, "v_min_u32 %0, %0, %2") \
    X(v_max_f32, "v_max_f32 %0, %0, %2") \
    X(v_med3_f32, "v_med3_f32 %0, %0, %2, %3") \
    X(v_pk_min_u16, "v_pk_min_u16 %0, %0, %2") \
    X(v_pk_add_u16, "v_pk_add_u16 %0, %0, %2") \
    X(v_sad_u8, "v_sad_u8 %0, %0, %2, %3") \
    X(v_dot4_u32_u8, "v_dot4_u32_u8 %0, %0, %2, %3") \
    X(v_ffbh_u32, "v_ffbh_u32 %0, %0") \
    X(v_ffbl_b32, "v_ffbl_b32 %0, %0") \
    X(v_bcnt_u32_b32, "v_bcnt_u32_b32 %0, %0, %2") \
    X(v_mov_b32_dpp_row_shr1, "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf") \
    X(v_add_u32_dpp_row_shr1, "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf") \
    X(v_readlane_like_mov_sgpr, "v_mov_b32 %0, %4") \
    X(v_mul_f32_sgpr, "v_mul_f32 %0, %4, %0") \
    X(v_ldexp_f32, "v_ldexp_f32 %0, %0, %2") \
    X(v_rcp_f32, "v_rcp_f32 %0, %0") \
    X(v_add_u32_sdwa_bytes, "v_add_u32_sdwa %0, %2, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_0") \
    X(v_sub_u32_sdwa_bytes, "v_sub_u32_sdwa %0, %2, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2") \
    X(v_not_b32, "v_not_b32 %0, %0") \
    X(v_mov_b64, "v_mov_b64 %1, %5") \
    X(v_cndmask_b32_vcc_set_once, "v_cndmask_b32 %0, %0, %2, vcc") \
    X(v_cndmask_b32_sgpr_pair_mask, "v_cndmask_b32 %0, %0, %2, s[22:23]") \
    X(v_cmp_then_v_cndmask_b32_vcc, "v_cmp_lt_u32 vcc, %0, %2\n\tv_cndmask_b32 %0, %0, %3, vcc") \
    X(v_dot2_u32_u16, "v_dot2_u32_u16 %0, %0, %2, %3") \
    X(v_readfirstlane_b32, "v_readfirstlane_b32 s20, %0") \
    X(v_readlane_b32, "v_readlane_b32 s20, %0, 3") \
    X(v_writelane_b32, "v_writelane_b32 %0, s22, 3") \
    X(v_mbcnt_lo_u32_b32, "v_mbcnt_lo_u32_b32 %0, %2, %0") \
    X(v_mbcnt_hi_u32_b32, "v_mbcnt_hi_u32_b32 %0, %2, %0") \
    X(v_xad_u32, "v_xad_u32 %0, %0, %2, %3") \
    X(v_bfe_i32, "v_bfe_i32 %0, %0, 8, 8") \
    X(v_lshrrev_b64, "v_lshrrev_b64 %1, 3, %1") \
    X(v_max_u32, "v_max_u32 %0, %0, %2") \
    X(v_max3_u32, "v_max3_u32 %0, %0, %2, %3") \
    X(v_min_i32, "v_min_i32 %0, %0, %2") \
    X(v_min_f32, "v_min_f32 %0, %0, %2") \
    X(v_rcp_iflag_f32, "v_rcp_iflag_f32 %0, %0") \
    X(v_sub_u16, "v_sub_u16 %0, %0, %2") \
    X(v_lshlrev_b16, "v_lshlrev_b16 %0, 3, %0") \
    X(v_lshrrev_b16, "v_lshrrev_b16 %0, 3, %0") \
    X(v_ashrrev_i16, "v_ashrrev_i16 %0, 3, %0") \
    X(v_add_u16, "v_add_u16 %0, %0, %2") \
    X(v_mul_lo_u16, "v_mul_lo_u16 %0, %0, %2") \
    X(v_mad_u16, "v_mad_u16 %0, %0, %2, %3") \
    X(v_mul_i32_i24, "v_mul_i32_i24 %0, %0, %2") \
    X(v_mad_i32_i24, "v_mad_i32_i24 %0, %0, %2, %3") \
    X(v_nop, "v_nop") \
    X(v_pk_add_u16_op_sel, "v_pk_add_u16 %0, %0, %2 op_sel:[1,0] op_sel_hi:[0,1]") \
    X(v_pk_sub_i16, "v_pk_sub_i16 %0, %0, %2") \
    X(v_pk_lshlrev_b16, "v_pk_lshlrev_b16 %0, 3, %0") \
    X(v_pk_mad_i16, "v_pk_mad_i16 %0, %0, %2, %3") \
    X(v_cvt_f32_f16, "v_cvt_f32_f16 %0, %0") \
    X(v_cvt_pk_i16_i32, "v_cvt_pk_i16_i32 %0, %0, %2") \
    X(v_cvt_pknorm_i16_f32, "v_cvt_pknorm_i16_f32 %0, %0, %2") \
    X(v_bitop3_b32, "v_bitop3_b32 %0, %0, %2, %3 bitop3:0x6c") \
    X(v_add_lshl_u32, "v_add_lshl_u32 %0, %0, %2, 2") \
    X(v_lshl_add_u64, "v_lshl_add_u64 %1, %1, 0, %5") \
    X(v_cvt_f32_i32, "v_cvt_f32_i32 %0, %0") \
    X(v_max_i32, "v_max_i32 %0, %0, %2") \
    X(v_subrev_u32, "v_subrev_u32 %0, %2, %0") \
    X(v_fmamk_f32, "v_fmamk_f32 %0, %0, 0x3e991687, %2") \
    X(v_fmaak_f32, "v_fmaak_f32 %0, %0, %2, 0x3e991687") \
    X(v_mad_i64_i32, "v_mad_i64_i32 %1, s[20:21], %0, %2, %1") \
    X(v_lshlrev_b64, "v_lshlrev_b64 %1, 3, %1") \
    X(v_mad_u64_u32, "v_mad_u64_u32 %1, vcc, %0, %2, %1") \
    X(v_pk_mul_f32_half_count, "v_pk_mul_f32 %1, %1, %5") \
    X(v_pk_add_f32_half_count, "v_pk_add_f32 %1, %1, %5") \
    X(v_pk_mul_f32_sgpr_low_for_both, "v_pk_mul_f32 %1, %1, s[20:21] op_sel_hi:[1,0]") \
    X(s_mov_then_v_mul_f32, "s_mov_b32 s20, 0x3e991687\n\tv_mul_f32 %0, s20, %0") \
    X(two_s_mov_then_v_pk_mul_f32, "s_mov_b32 s20, 0x3e991687\n\ts_mov_b32 s21, 0x3e991688\n\tv_pk_mul_f32 %1, %1, s[20:21]") \
    X(s_nop_then_v_mul_f32, "s_nop 0\n\tv_mul_f32 %0, %0, %2") \
    X(v_mul_f32_then_dependent_v_add_f32, "v_mul_f32 %0, %0, %2\n\tv_add_f32 %0, %0, %3") \
    X(s_mov_b32_inline_then_v_mul_f32, "s_mov_b32 s20, 3\n\tv_mul_f32 %0, %0, %2") \
    X(s_add_u32_then_v_mul_f32, "s_add_u32 s20, s20, 3\n\tv_mul_f32 %0, %0, %2") \
    X(s_and_b64_then_v_mul_f32, "s_and_b64 s[20:21], s[20:21], exec\n\tv_mul_f32 %0, %0, %2") \
    X(s_or_b64_exec_then_v_mul_f32, "s_or_b64 exec, exec, exec\n\tv_mul_f32 %0, %0, %2") \
    X(s_cmp_cbranch_then_v_mul_f32, "s_cmp_eq_u32 s20, 77\n\ts_cbranch_scc1 1f\n1:\n\tv_mul_f32 %0, %0, %2") \
    X(s_waitcnt_then_v_mul_f32, "s_waitcnt lgkmcnt(0)\n\tv_mul_f32 %0, %0, %2") \
    X(s_mov_then_4_v_mul_f32, "s_mov_b32 s20, 0x3e991687\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %0, %0, %3\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %0, %0, %3") \
    X(v_readfirstlane_then_v_mul_f32, "v_readfirstlane_b32 s20, %2\n\tv_mul_f32 %0, %0, %2") \
    X(ds_read_b32_then_v_mul_f32, "ds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)\n\tv_mul_f32 %0, %0, %2") \
    X(s_mov_then_v_cvt, "s_mov_b32 s20, 3\n\tv_cvt_f32_ubyte0 %0, %0") \
    X(v_mul_f32_sources_in_one_register_bank, "v_mul_f32 v100, v104, v108") \
    X(v_mul_f32_sources_in_two_register_banks, "v_mul_f32 v100, v105, v110") \
    X(v_fma_f32_sources_in_one_register_bank, "v_fma_f32 v100, v104, v108, v112") \
    X(v_fma_f32_sources_in_three_register_banks, "v_fma_f32 v100, v105, v110, v115") \
    X(v_mul_f32_destination_rotates, "v_mul_f32 %0, v105, v110")

typedef float f2 __attribute__((ext_vector_type(2)));

enum Mode {
#define X(name, text) M_##name,
    MODES(X)
#undef X
    M_COUNT
};

template <int MODE>
__global__ __launch_bounds__(1024) void k(float *out, unsigned long long *cyc, unsigned long long *real, int iters, float seed) {
    extern __shared__ float pin[];   // more than half of a CU's LDS: one workgroup per CU
    float a[16];
    f2 p[16];
    for (int i = 0; i < 16; i++) { a[i] = seed * (i + 1) + threadIdx.x; p[i] = (f2){a[i], a[i] + 1.0f}; }
    const float c0 = seed * 0.5f + 0.00001f * threadIdx.x, c1 = seed + 0.25f;
    const f2 pc = (f2){c0, c1};
    const float sg = __builtin_amdgcn_readfirstlane(seed + 3.0f);
    // the masks the select modes read: set here, once; the instruction streams below name vcc and s[22:23] as
    // clobbered, so the compiler keeps nothing of its own in them
    asm volatile("s_mov_b32 vcc_lo, 0x55555555\n\ts_mov_b32 vcc_hi, 0x33333333\n\ts_mov_b32 s22, 0x0f0f0f0f\n\ts_mov_b32 s23, 0x55555555"
                 ::: "vcc", "s22", "s23");
    __syncthreads();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        switch (MODE) {
#define X(name, text)                                                                                           \
    case M_##name: {                                                                                            \
        _Pragma("unroll") for (int r = 0; r < 16; r++) {                                                        \
            _Pragma("unroll") for (int i = 0; i < 16; i++)                                                      \
                asm volatile(text : "+v"(a[i]), "+v"(p[i]) : "v"(c0), "v"(c1), "s"(sg), "v"(pc) : "s20", "s21", "s22", "s23", "vcc", "scc", "v100", "v104", "v105", "v108", "v110", "v112", "v115");       \
        }                                                                                                       \
    } break;
            MODES(X)
#undef X
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 16; i++) s += a[i] + p[i].x + p[i].y;
    if (seed == 12345.0f) pin[threadIdx.x] = s;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; real[blockIdx.x] = r1 - r0; }
}

template <int MODE>
void run(const char *name, int waves_per_simd) {
    const int iters = 1024;                          // x 256 instructions
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int threads = 256 * waves_per_simd, grid = cus;
    float *out; unsigned long long *cyc, *real;
    hipMalloc(&out, (size_t)grid * threads * 4);
    hipMalloc(&cyc, (size_t)grid * 8);
    hipMalloc(&real, (size_t)grid * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds = 96 * 1024;
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    k<MODE><<<grid, threads, lds>>>(out, cyc, real, 8, 1.0f);
    hipDeviceSynchronize();
    float t[2] = {0, 0};
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        k<MODE><<<grid, threads, lds>>>(out, cyc, real, iters * (rep + 1), 1.0f);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        hipEventElapsedTime(&t[rep], e0, e1);
    }
    std::vector<unsigned long long> h(grid), hr(grid);
    hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
    hipMemcpy(hr.data(), real, grid * 8, hipMemcpyDeviceToHost);
    double avg = 0, avr = 0;
    for (auto v : h) avg += (double)v;
    for (auto v : hr) avr += (double)v;
    const double ghz = avg / avr * 0.1;              // s_memrealtime ticks at 100 MHz
    const double n = (double)iters * 256 * waves_per_simd;          // wave-instructions per SIMD in N iterations
    const double ns = (t[1] - t[0]) * 1e6 / n;
    const double incyc = avg / grid / (2.0 * iters * 256) / waves_per_simd;   // from the kernel's own counter (2N launch)
    printf("%-28s %d waves/SIMD: %6.3f ns = %5.2f cycles per wave-instruction per SIMD (in-kernel counter: %5.2f), clock %.2f GHz\n",
           name, waves_per_simd, ns, ns * ghz, incyc, ghz);
    hipFree(out); hipFree(cyc); hipFree(real);
}

int main(int argc, char **argv) {
    const int w = argc > 1 ? atoi(argv[1]) : 4;
#define X(name, text) run<M_##name>(#name, w);
    MODES(X)
#undef X
    return 0;
}
