// Per-thread 8x8 transforms: level shift, DCT (three variants), quantisation,
// zigzag.  One thread owns one block; the 64 samples arrive as 16 dwords
// (row r = px[2r], px[2r+1], bytes little-endian) and the 64 quantised
// coefficients leave as 128 bytes in zigzag order at `slot` (LDS, 16-byte
// aligned).  Everything is fully unrolled so that all array indices are
// compile-time constants and the arrays live in registers.
//
// This translation unit is built with -ffp-contract=off: `a*b+c` never fuses.
// Where a fused multiply-add is wanted it is written as fmaf().
#pragma once
#include <hip/hip_runtime.h>

#include "tables.h"

namespace jpg {

// cos(((2x+1)*u) as f32 * PI / 16) exactly as the reference evaluates it in f32
// (src/jpeg/dct_quant.rs:222-223); index u*8+x.  glibc's cosf is correctly
// rounded on these 64 arguments, so the bit patterns do not depend on the libm.
constexpr uint32_t COS_BITS[64] = {
    0x3f800000u, 0x3f800000u, 0x3f800000u, 0x3f800000u, 0x3f800000u, 0x3f800000u, 0x3f800000u, 0x3f800000u,
    0x3f7b14beu, 0x3f54db31u, 0x3f0e39d9u, 0x3e47c5bcu, 0xbe47c5c2u, 0xbf0e39dcu, 0xbf54db32u, 0xbf7b14bfu,
    0x3f6c835eu, 0x3ec3ef15u, 0xbec3ef18u, 0xbf6c8360u, 0xbf6c835eu, 0xbec3ef0bu, 0x3ec3ef1bu, 0x3f6c835fu,
    0x3f54db31u, 0xbe47c5c2u, 0xbf7b14bfu, 0xbf0e39d6u, 0x3f0e39ddu, 0x3f7b14beu, 0x3e47c5b1u, 0xbf54db34u,
    0x3f3504f3u, 0xbf3504f3u, 0xbf3504f1u, 0x3f3504f7u, 0x3f3504f3u, 0xbf3504fbu, 0xbf3504efu, 0x3f3504f4u,
    0x3f0e39d9u, 0xbf7b14bfu, 0x3e47c5c8u, 0x3f54db2du, 0xbf54db34u, 0xbe47c5bbu, 0x3f7b14bfu, 0xbf0e39e4u,
    0x3ec3ef15u, 0xbf6c835eu, 0x3f6c835fu, 0xbec3ef25u, 0xbec3ef06u, 0x3f6c835bu, 0xbf6c8362u, 0x3ec3ef25u,
    0x3e47c5bcu, 0xbf0e39d6u, 0x3f54db2du, 0xbf7b14bdu, 0x3f7b14c1u, 0xbf54db31u, 0x3f0e39e9u, 0xbe47c614u};

__device__ __host__ constexpr float cosv(int u, int x) { return __builtin_bit_cast(float, COS_BITS[u * 8 + x]); }

// (0.25 * alpha_u) * alpha_v in f32, alpha_0 = 1.0/SQRT_2 (src/jpeg/dct_quant.rs:196,228)
__device__ __host__ constexpr float dct_scale(int u, int v) {
    constexpr float inv_sqrt2 = 1.0f / 1.41421356237309505f;
    float s = 0.25f * (u == 0 ? inv_sqrt2 : 1.0f);
    return s * (v == 0 ? inv_sqrt2 : 1.0f);
}

template <bool CHROMA>
__device__ __host__ constexpr int quant(int n) {
    return CHROMA ? jpgtab::QT_CHROMA[n] : jpgtab::QT_LUMA[n];
}

// tables for the variant that loops over u at run time
__device__ const float COS_DEV[64] = {
    cosv(0, 0), cosv(0, 1), cosv(0, 2), cosv(0, 3), cosv(0, 4), cosv(0, 5), cosv(0, 6), cosv(0, 7),
    cosv(1, 0), cosv(1, 1), cosv(1, 2), cosv(1, 3), cosv(1, 4), cosv(1, 5), cosv(1, 6), cosv(1, 7),
    cosv(2, 0), cosv(2, 1), cosv(2, 2), cosv(2, 3), cosv(2, 4), cosv(2, 5), cosv(2, 6), cosv(2, 7),
    cosv(3, 0), cosv(3, 1), cosv(3, 2), cosv(3, 3), cosv(3, 4), cosv(3, 5), cosv(3, 6), cosv(3, 7),
    cosv(4, 0), cosv(4, 1), cosv(4, 2), cosv(4, 3), cosv(4, 4), cosv(4, 5), cosv(4, 6), cosv(4, 7),
    cosv(5, 0), cosv(5, 1), cosv(5, 2), cosv(5, 3), cosv(5, 4), cosv(5, 5), cosv(5, 6), cosv(5, 7),
    cosv(6, 0), cosv(6, 1), cosv(6, 2), cosv(6, 3), cosv(6, 4), cosv(6, 5), cosv(6, 6), cosv(6, 7),
    cosv(7, 0), cosv(7, 1), cosv(7, 2), cosv(7, 3), cosv(7, 4), cosv(7, 5), cosv(7, 6), cosv(7, 7)};

struct QuantF {
    float q[2][64];
    uint8_t zz[64];
    constexpr QuantF() : q{}, zz{} {
        for (int n = 0; n < 64; n++) {
            q[0][n] = (float)jpgtab::QT_LUMA[n];
            q[1][n] = (float)jpgtab::QT_CHROMA[n];
            zz[n] = jpgtab::ZZ_OF_NATURAL.k[n];
        }
    }
};
__device__ const QuantF QUANT_DEV{};

__device__ __forceinline__ float sample(const uint32_t (&px)[16], int r, int c) {
    // level shift (src/jpeg/dct_quant.rs:63-65): n - 128, exact in f32
    return (float)((px[2 * r + (c >> 2)] >> (8 * (c & 3))) & 0xffu) - 128.0f;
}
__device__ __forceinline__ int sample_i(const uint32_t (&px)[16], int r, int c) {
    return (int)((px[2 * r + (c >> 2)] >> (8 * (c & 3))) & 0xffu) - 128;
}

// res[k] = quantised coefficient at zigzag position k; pack and store 128 bytes
__device__ __forceinline__ void store_zigzag(const int (&res)[64], uint32_t *slot) {
    uint4 *s4 = reinterpret_cast<uint4 *>(slot);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint4 w;
        w.x = ((uint32_t)res[8 * i + 0] & 0xffffu) | ((uint32_t)res[8 * i + 1] << 16);
        w.y = ((uint32_t)res[8 * i + 2] & 0xffffu) | ((uint32_t)res[8 * i + 3] << 16);
        w.z = ((uint32_t)res[8 * i + 4] & 0xffffu) | ((uint32_t)res[8 * i + 5] << 16);
        w.w = ((uint32_t)res[8 * i + 6] & 0xffffu) | ((uint32_t)res[8 * i + 7] << 16);
        s4[i] = w;
    }
}

__device__ __forceinline__ void store_packed(const uint32_t (&w)[32], uint32_t *slot) {
    uint4 *s4 = reinterpret_cast<uint4 *>(slot);
#pragma unroll
    for (int i = 0; i < 8; i++) s4[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}

// ((0.25*au*av*sum) / q) as i16 (src/jpeg/dct_quant.rs:227-230): one unfused
// multiply, one correctly rounded divide, truncation toward zero.
__device__ __forceinline__ int quantise_real(float sum, float scale, float q) {
    float s = scale * sum;
    float d = s / q;
    return (int)d;
}

// --- separable f32 DCT ------------------------------------------------------
// Same cosine table as the reference; the 64-term sum is regrouped as two
// 8-point passes (rows, then columns).  Each pass splits into an even half on
// a_i + a_(7-i) and an odd half on a_i - a_(7-i) (cos(v,7-i) = +-cos(v,i)), and
// the even half once more (v = 0,4 from the sums, v = 2,6 from the
// differences).  35 operations per pass instead of 64.  The DC path
// (v = 0, u = 0) only adds integers and is exact, and the DC coefficient is
// quantised with the reference's own two-step expression, so flat blocks come
// out identical.  Other coefficients may differ from the reference's
// summation order by +-1 where a quotient sits on a truncation boundary.
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __host__ constexpr float rcp_rn(int q) { return 1.0f / (float)q; }  // RN(1/q), folded at compile time

// RN(s / q) for a compile-time divisor: q0 = s*z, r = s - q0*q (exact in an
// FMA), d = q0 + r*z with z = RN(1/q).  Bit-identical to the IEEE quotient for
// every divisor in the two tables (checked exhaustively by tools/check_divide.c).
__device__ __forceinline__ float div_const(float s, float q, float z) {
    const float q0 = s * z;
    const float r = fmaf(-q0, q, s);
    return fmaf(r, z, q0);
}

__device__ __forceinline__ void dct8(float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7,
                                     float (&o)[8]) {
    const float s0 = a0 + a7, s1 = a1 + a6, s2 = a2 + a5, s3 = a3 + a4;
    const float d0 = a0 - a7, d1 = a1 - a6, d2 = a2 - a5, d3 = a3 - a4;
    const float ss0 = s0 + s3, ss1 = s1 + s2, sd0 = s0 - s3, sd1 = s1 - s2;
    o[0] = ss0 + ss1;                                   // cos(0,.) = 1
    o[4] = cosv(4, 0) * (ss0 - ss1);                    // cos(4,i) = +c4, -c4, -c4, +c4
    o[2] = fmaf(cosv(2, 1), sd1, cosv(2, 0) * sd0);     // cos(2,i) = c2, c6, -c6, -c2
    o[6] = fmaf(cosv(6, 1), sd1, cosv(6, 0) * sd0);     // cos(6,i) = c6, -c2, c2, -c6
#pragma unroll
    for (int v = 1; v < 8; v += 2)
        o[v] = fmaf(cosv(v, 3), d3, fmaf(cosv(v, 2), d2, fmaf(cosv(v, 1), d1, cosv(v, 0) * d0)));
}

// scale and quantiser of coefficient (u, v) as one factor
template <bool CHROMA>
__device__ __host__ constexpr float qfold(int u, int v) {
    return dct_scale(u, v) / (float)quant<CHROMA>(u * 8 + v);
}

// Second (vertical) pass for column V with that factor folded into the constants of the last
// operation of every output, so that o[u] is the quantised value before truncation.  o[0] is
// left unscaled (its last operation is an addition; the DC coefficient takes the exact path).
template <bool CHROMA, int V>
__device__ __forceinline__ void dct8_quant(float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7,
                                           float (&o)[8]) {
    const float s0 = a0 + a7, s1 = a1 + a6, s2 = a2 + a5, s3 = a3 + a4;
    const float d0 = a0 - a7, d1 = a1 - a6, d2 = a2 - a5, d3 = a3 - a4;
    const float ss0 = s0 + s3, ss1 = s1 + s2, sd0 = s0 - s3, sd1 = s1 - s2;
    o[0] = ss0 + ss1;
    o[4] = (cosv(4, 0) * qfold<CHROMA>(4, V)) * (ss0 - ss1);
    o[2] = fmaf(cosv(2, 1) * qfold<CHROMA>(2, V), sd1, (cosv(2, 0) * qfold<CHROMA>(2, V)) * sd0);
    o[6] = fmaf(cosv(6, 1) * qfold<CHROMA>(6, V), sd1, (cosv(6, 0) * qfold<CHROMA>(6, V)) * sd0);
#pragma unroll
    for (int u = 1; u < 8; u += 2) {
        const float k = qfold<CHROMA>(u, V);
        o[u] = fmaf(cosv(u, 3) * k, d3, fmaf(cosv(u, 2) * k, d2, fmaf(cosv(u, 1) * k, d1, (cosv(u, 0) * k) * d0)));
    }
}

// f32 -> i32 (truncation, like `as i16` on an in-range value) written straight into one 16-bit half
// of a dword: conversion and packing in one instruction.  `fresh`: the other half is not written yet.
template <int HALF>
__device__ __forceinline__ void cvt_pack16(uint32_t &d, float x, bool fresh) {
    if (fresh) {
        if (HALF) asm("v_cvt_i32_f32_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:DWORD" : "=v"(d) : "v"(x));
        else asm("v_cvt_i32_f32_sdwa %0, %1 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD" : "=v"(d) : "v"(x));
    } else {
        if (HALF) asm("v_cvt_i32_f32_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(d) : "v"(x));
        else asm("v_cvt_i32_f32_sdwa %0, %1 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(d) : "v"(x));
    }
}

template <bool CHROMA, int V>
__device__ __forceinline__ void column_real(const float (&t)[8][8], uint32_t (&w)[32], bool (&seen)[32]) {
    float o[8];
    dct8_quant<CHROMA, V>(t[0][V], t[1][V], t[2][V], t[3][V], t[4][V], t[5][V], t[6][V], t[7][V], o);
    if (V == 0) {
        // level shift (src/jpeg/dct_quant.rs:63-65): subtracting 128 from every sample moves only
        // the DC sum, by exactly 64*128; then the reference's ((0.25*au*av*sum)/q) as i16
        // (src/jpeg/dct_quant.rs:227-230)
        const float s = dct_scale(0, 0) * (o[0] - 8192.0f);
        o[0] = div_const(s, (float)quant<CHROMA>(0), rcp_rn(quant<CHROMA>(0)));
    } else {
        o[0] = o[0] * qfold<CHROMA>(0, V);
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
        const int k = jpgtab::ZZ_OF_NATURAL.k[u * 8 + V];
        if (k & 1) cvt_pack16<1>(w[k >> 1], o[u], !seen[k >> 1]);
        else cvt_pack16<0>(w[k >> 1], o[u], !seen[k >> 1]);
        seen[k >> 1] = true;
    }
}

// first pass: t[x][v] = sum_y e[x][y] cos(v,y); the level shift is applied to the DC sum later.  The same for luma
// and chroma blocks.
// KEEP_F32: the samples stay f32 from the conversion on.  Left alone the compiler forms the first two butterfly stages on
// the bytes (sums of small integers are exact either way) and converts the results: eight 4-cycle byte-select
// additions and eight 4-cycle conversions per row where eight conversions and eight 2-cycle f32 additions do.
// Measured per kernel: 1.2 % faster for the coding kernel at 512-pixel tiles, slower for the front kernel and at
// 1024-pixel tiles, hence a parameter.
template <bool KEEP_F32>
__device__ __forceinline__ void rows_real(const uint32_t (&px)[16], float (&t)[8][8]) {
#pragma unroll
    for (int x = 0; x < 8; x++) {
        float e[8];
#pragma unroll
        for (int y = 0; y < 8; y++) {
            e[y] = (float)((px[2 * x + (y >> 2)] >> (8 * (y & 3))) & 0xffu);
            if constexpr (KEEP_F32) asm("" : "+v"(e[y]));
        }
        dct8(e[0], e[1], e[2], e[3], e[4], e[5], e[6], e[7], t[x]);
    }
}

// second pass with the quantiser folded in, truncation, zigzag, 128 bytes to `slot`
template <bool CHROMA>
__device__ __forceinline__ void columns_real(const float (&t)[8][8], uint32_t *slot) {
    uint32_t w[32];
    bool seen[32] = {};
    column_real<CHROMA, 0>(t, w, seen);
    column_real<CHROMA, 1>(t, w, seen);
    column_real<CHROMA, 2>(t, w, seen);
    column_real<CHROMA, 3>(t, w, seen);
    column_real<CHROMA, 4>(t, w, seen);
    column_real<CHROMA, 5>(t, w, seen);
    column_real<CHROMA, 6>(t, w, seen);
    column_real<CHROMA, 7>(t, w, seen);
    store_packed(w, slot);
}

template <bool CHROMA>
__device__ __forceinline__ void transform_real(const uint32_t (&px)[16], uint32_t *slot) {
    float t[8][8];
    rows_real<false>(px, t);
    columns_real<CHROMA>(t, slot);
}

// --- the reference's own summation order, bit-exact -------------------------
// sum over x (outer), y (inner) of (e*cos(u,x))*cos(v,y), accumulated left to
// right in f32 with separate multiply and add (src/jpeg/dct_quant.rs:217-225).
// The u loop stays rolled to bound code size; results go to LDS as int16.
template <bool CHROMA>
__device__ __forceinline__ void transform_real_strict(const uint32_t (&px)[16], uint32_t *slot) {
    float e[64];
#pragma unroll
    for (int x = 0; x < 8; x++)
#pragma unroll
        for (int y = 0; y < 8; y++) e[x * 8 + y] = sample(px, x, y);
    int16_t *s16 = reinterpret_cast<int16_t *>(slot);
    const float inv_sqrt2 = 1.0f / 1.41421356237309505f;
#pragma unroll 1
    for (int u = 0; u < 8; u++) {
        float cu[8];
#pragma unroll
        for (int x = 0; x < 8; x++) cu[x] = COS_DEV[u * 8 + x];
        float t[64];
#pragma unroll
        for (int x = 0; x < 8; x++)
#pragma unroll
            for (int y = 0; y < 8; y++) t[x * 8 + y] = e[x * 8 + y] * cu[x];
        const float su = 0.25f * (u == 0 ? inv_sqrt2 : 1.0f);
#pragma unroll
        for (int v = 0; v < 8; v++) {
            float sum = 0.0f;
#pragma unroll
            for (int x = 0; x < 8; x++)
#pragma unroll
                for (int y = 0; y < 8; y++) {
                    float p = t[x * 8 + y] * cosv(v, y);
                    sum = sum + p;
                }
            const float scale = su * (v == 0 ? inv_sqrt2 : 1.0f);
            const int n = u * 8 + v;
            int r = quantise_real(sum, scale, QUANT_DEV.q[CHROMA ? 1 : 0][n]);
            s16[QUANT_DEV.zz[n]] = (int16_t)r;
        }
    }
}

// --- all-lifting binDCT-C (src/jpeg/dct_quant.rs:93-129) --------------------
__device__ __forceinline__ void bin8(int &a0, int &a1, int &a2, int &a3, int &a4, int &a5, int &a6, int &a7) {
    int x7_1 = a0 - a7, x0_1 = a0 - (x7_1 >> 1);
    int x6_1 = a1 - a6, x1_1 = a1 - (x6_1 >> 1);
    int x5_1 = a2 - a5, x2_1 = a2 - (x5_1 >> 1);
    int x4_1 = a3 - a4, x3_1 = a3 - (x4_1 >> 1);
    x6_1 = ((x5_1 * 3) >> 3) + x6_1;
    x5_1 = ((x6_1 * 5) >> 3) - x5_1;
    int x0_2 = x0_1 + x3_1, x3_2 = x0_1 - x3_1;
    int x1_2 = x1_1 + x2_1, x2_2 = x1_1 - x2_1;
    int x4_2 = x4_1 + x5_1, x5_2 = x4_1 - x5_1;
    int x6_2 = x7_1 - x6_1, x7_2 = x7_1 + x6_1;
    x4_2 = x4_2 - (x7_2 >> 3);
    x0_2 = x0_2 + x1_2;
    x1_2 = -x1_2 + (x0_2 >> 1);
    x2_2 = x2_2 - ((x3_2 * 3) >> 3);
    x3_2 = x3_2 + ((x2_2 * 3) >> 3);
    x5_2 = x5_2 + ((x6_2 * 7) >> 3);
    x6_2 = x6_2 - (x5_2 >> 1);
    // output order of src/jpeg/dct_quant.rs:122-129
    a0 = x0_2; a4 = x1_2; a6 = x2_2; a2 = x3_2;
    a7 = x4_2; a5 = x5_2; a3 = x6_2; a1 = x7_2;
}

// The same butterfly in f32.  Every value of the transform is an integer far below 2^22, so f32 additions and
// subtractions are exact; what integer code spends on each lifting step -- a shift-and-add (4 cycles of issue
// time on a gfx950 SIMD) and an arithmetic shift (2) for floor(k*x/8) -- is one fused multiply-add (2) here: with
// the f32 rounding mode set to "toward minus infinity", fma(x, k/8, 1.5*2^23) is exactly 1.5*2^23 + floor(k*x/8)
// (the sum is formed exactly and rounded once, on a grid of whole numbers).  The magic constant is not taken off
// again: a value travels as n + D*M with D in {-1, 0, +1} (M = 1.5*2^23, all sums still exact), the domain D is
// part of the type, a multiply-add accepts any domain (its constant absorbs k/8 of the bias) and delivers +M or -M
// as asked, and the additions are arranged so that the biases cancel: 37 operations per 8 points on biased inputs,
// 40 or 41 where the results have to be plain.  tools/check_bin_f32.c checks every multiply-add form over its whole
// input range, the reciprocal quantiser over every dividend and divisor, and the whole transform against the
// integer code above on random and extreme blocks, on the CPU under the same rounding mode.
namespace binf {
constexpr float M = 12582912.0f;   // 1.5 * 2^23: ulp 1 on [2^23, 2^24), room for |n| < 2^22 on either side

template <int D>
struct V {
    float v;   // n + D * M
};
template <int DA, int DB>
__device__ __forceinline__ V<DA + DB> operator+(V<DA> a, V<DB> b) {
    static_assert(DA + DB >= -1 && DA + DB <= 1, "biases must cancel");
    return {a.v + b.v};
}
template <int DA, int DB>
__device__ __forceinline__ V<DA - DB> operator-(V<DA> a, V<DB> b) {
    static_assert(DA - DB >= -1 && DA - DB <= 1, "biases must cancel");
    return {a.v - b.v};
}
// floor(NUM * x / 8) + OFF in domain DH (+1 or -1).  Round-toward-minus-infinity must be in force.
template <int DH, int NUM, int OFF = 0, int DX>
__device__ __forceinline__ V<DH> fl(V<DX> x) {
    static_assert(DH == 1 || DH == -1, "the floor needs the magic constant");
    constexpr float c = NUM / 8.0f;
    constexpr float K = (float)((double)M * (8 * DH - NUM * DX) / 8.0 + OFF);   // exact: a multiple of 2^19 below 2^25
    return {__builtin_fmaf(x.v, c, K)};
}
template <int D>
__device__ __forceinline__ V<0> plain(V<D> x) {
    if constexpr (D == 0) return x;
    else return {x.v - (float)D * M};
}

// One 8-point pass (src/jpeg/dct_quant.rs:93-129) on r[0..7], all in domain DI, in place.  OFF: a level shift to
// take off the inputs on the way (they hold n + OFF; only the four sums see it, the differences cancel it).
// PLAIN: every result in domain 0; otherwise results 1, 2, 4, 6 stay in domain +1 (row_out_domain).
template <int DI, int OFF, bool PLAIN>
__device__ __forceinline__ void bin8f(float &r0, float &r1, float &r2, float &r3, float &r4, float &r5, float &r6, float &r7) {
    static_assert(DI != 0 || OFF == 0, "a level shift rides on biased inputs only");
    const V<DI> a0{r0}, a1{r1}, a2{r2}, a3{r3}, a4{r4}, a5{r5}, a6{r6}, a7{r7};
    const V<0> x7_1 = a0 - a7, x6_1 = a1 - a6, x5_1 = a2 - a5, x4_1 = a3 - a4;
    // the odd half: x6_1 += 3/8 x5_1; x5_1 = 5/8 x6_1 - x5_1; butterflies; x4_2 -= x7_2/8; x5_2 += 7/8 x6_2; x6_2 -= x5_2/2
    const V<1> y6 = x6_1 + fl<1, 3>(x5_1);
    const V<1> y5 = fl<1, 5>(y6) - x5_1;
    const V<1> x4_2 = x4_1 + y5;
    const V<-1> x5_2 = x4_1 - y5;
    const V<-1> x6_2 = x7_1 - y6;
    const V<1> x7_2 = x7_1 + y6;
    const V<0> o7 = x4_2 - fl<1, 1>(x7_2);
    const V<0> o5 = x5_2 + fl<1, 7>(x6_2);
    const V<0> o3 = x6_2 - fl<-1, 4>(o5);
    // the even half: x_i_1 = a_i - (x_(7-i)_1 >> 1); butterflies; x0_2 += x1_2; x1_2 = (x0_2 >> 1) - x1_2;
    // x2_2 -= 3/8 x3_2; x3_2 += 3/8 x2_2
    if constexpr (DI != 0) {
        const V<0> x0_1 = a0 - fl<DI, 4, OFF>(x7_1), x1_1 = a1 - fl<DI, 4, OFF>(x6_1);
        const V<0> x2_1 = a2 - fl<DI, 4, OFF>(x5_1), x3_1 = a3 - fl<DI, 4, OFF>(x4_1);
        const V<0> e0 = x0_1 + x3_1, e3 = x0_1 - x3_1, e1 = x1_1 + x2_1, e2 = x1_1 - x2_1;
        const V<0> o0 = e0 + e1;
        const V<1> o4 = fl<1, 4>(o0) - e1;
        const V<1> o6 = e2 - fl<-1, 3>(e3);
        const V<1> o2 = e3 + fl<1, 3>(o6);
        r0 = o0.v;
        if constexpr (PLAIN) { r2 = plain(o2).v; r4 = plain(o4).v; r6 = plain(o6).v; }
        else { r2 = o2.v; r4 = o4.v; r6 = o6.v; }
    } else {
        const V<0> x0_1 = plain(a0 - fl<1, 4>(x7_1)), x1_1 = plain(a1 - fl<1, 4>(x6_1));
        const V<-1> x2_1 = a2 - fl<1, 4>(x5_1);
        const V<1> x3_1 = a3 - fl<-1, 4>(x4_1);
        const V<1> e0 = x0_1 + x3_1;
        const V<-1> e3 = x0_1 - x3_1;
        const V<-1> e1 = x1_1 + x2_1;
        const V<1> e2 = x1_1 - x2_1;
        const V<0> o0 = e0 + e1;
        const V<0> o4 = fl<-1, 4>(o0) - e1;
        const V<0> o6 = e2 - fl<1, 3>(e3);
        const V<0> o2 = e3 + fl<1, 3>(o6);
        r0 = o0.v; r2 = o2.v; r4 = o4.v; r6 = o6.v;
    }
    // output order of src/jpeg/dct_quant.rs:122-129
    r1 = PLAIN ? plain(x7_2).v : x7_2.v;
    r3 = o3.v; r5 = o5.v; r7 = o7.v;
}
// domain of result k of a pass with PLAIN = false
__device__ __host__ constexpr int row_out_domain(int k) { return (k == 1 || k == 2 || k == 4 || k == 6) ? 1 : 0; }

// The smallest f32 >= 1/q: a * z truncated is a / q (i32 / i32, src/jpeg/dct_quant.rs:182-186) for every
// |a| <= 2^16 and every divisor of the two tables, whatever the rounding mode of the multiplication.
__device__ __host__ constexpr float rcp_up(int q) {
    float z = 1.0f / (float)q;
    if ((double)z * (double)q < 1.0) z = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, z) + 1u);
    return z;
}

// M + byte N of w: a byte placed under the upper bytes of M's bit pattern (0x4B400000) is that sum
template <int N>
__device__ __forceinline__ float biased_byte(uint32_t w) {
    return __uint_as_float(__builtin_amdgcn_perm(0x4b400000u, w, 0x07060500u | (uint32_t)N));
}

template <bool CHROMA, int V_>
__device__ __forceinline__ void column_bin(float (&t)[8][8], uint32_t (&w)[32], bool (&seen)[32]) {
    bin8f<row_out_domain(V_), 0, true>(t[0][V_], t[1][V_], t[2][V_], t[3][V_], t[4][V_], t[5][V_], t[6][V_], t[7][V_]);
#pragma unroll
    for (int u = 0; u < 8; u++) {
        const int k = jpgtab::ZZ_OF_NATURAL.k[u * 8 + V_];
        const float q = t[u][V_] * rcp_up(quant<CHROMA>(u * 8 + V_));
        if (k & 1) cvt_pack16<1>(w[k >> 1], q, !seen[k >> 1]);
        else cvt_pack16<0>(w[k >> 1], q, !seen[k >> 1]);
        seen[k >> 1] = true;
    }
}
}  // namespace binf

// The rounding mode of f32 arithmetic: 0 nearest even (the default everywhere else), 2 toward minus infinity.
// The instructions that need the mode are tied to the switches by data: the block's samples pass through the
// first, its finished words through the second, so nothing computed from the one or feeding the other can be
// scheduled outside.  (Only the two bits of the f32 rounding mode are written.)
__device__ __forceinline__ void round_down_begin(uint32_t (&p)[16]) {
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 2\n\ts_nop 1"
                 : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]), "+v"(p[8]),
                   "+v"(p[9]), "+v"(p[10]), "+v"(p[11]), "+v"(p[12]), "+v"(p[13]), "+v"(p[14]), "+v"(p[15]));
}
__device__ __forceinline__ void round_down_end(uint32_t (&w)[32]) {
    asm volatile("" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]), "+v"(w[8]),
                      "+v"(w[9]), "+v"(w[10]), "+v"(w[11]), "+v"(w[12]), "+v"(w[13]), "+v"(w[14]), "+v"(w[15]));
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0\n\ts_nop 1"
                 : "+v"(w[16]), "+v"(w[17]), "+v"(w[18]), "+v"(w[19]), "+v"(w[20]), "+v"(w[21]), "+v"(w[22]), "+v"(w[23]),
                   "+v"(w[24]), "+v"(w[25]), "+v"(w[26]), "+v"(w[27]), "+v"(w[28]), "+v"(w[29]), "+v"(w[30]), "+v"(w[31])
                 :
                 : "memory");
}

template <bool CHROMA>
__device__ __forceinline__ void transform_bin(const uint32_t (&px_in)[16], uint32_t *slot) {
    uint32_t px[16];
#pragma unroll
    for (int i = 0; i < 16; i++) px[i] = px_in[i];
    round_down_begin(px);
    float t[8][8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        // samples as M + byte: domain +1 with the level shift (src/jpeg/dct_quant.rs:63-65) still to come off
        t[r][0] = binf::biased_byte<0>(px[2 * r]); t[r][1] = binf::biased_byte<1>(px[2 * r]);
        t[r][2] = binf::biased_byte<2>(px[2 * r]); t[r][3] = binf::biased_byte<3>(px[2 * r]);
        t[r][4] = binf::biased_byte<0>(px[2 * r + 1]); t[r][5] = binf::biased_byte<1>(px[2 * r + 1]);
        t[r][6] = binf::biased_byte<2>(px[2 * r + 1]); t[r][7] = binf::biased_byte<3>(px[2 * r + 1]);
        binf::bin8f<1, 128, false>(t[r][0], t[r][1], t[r][2], t[r][3], t[r][4], t[r][5], t[r][6], t[r][7]);
    }
    uint32_t w[32];
    bool seen[32] = {};
    binf::column_bin<CHROMA, 0>(t, w, seen);
    binf::column_bin<CHROMA, 1>(t, w, seen);
    binf::column_bin<CHROMA, 2>(t, w, seen);
    binf::column_bin<CHROMA, 3>(t, w, seen);
    binf::column_bin<CHROMA, 4>(t, w, seen);
    binf::column_bin<CHROMA, 5>(t, w, seen);
    binf::column_bin<CHROMA, 6>(t, w, seen);
    binf::column_bin<CHROMA, 7>(t, w, seen);
    round_down_end(w);
    store_packed(w, slot);
}

// --- the DC coefficient alone ------------------------------------------------
// The same operations on the same values as the full transforms above, with everything that does
// not reach coefficient 0 left out, so the result is the one the full transform stores at slot[0].
template <bool CHROMA>
__device__ __forceinline__ int dc_real(const uint32_t (&px)[16]) {
    float t0[8];
#pragma unroll
    for (int x = 0; x < 8; x++) {
        float e[8], o[8];
#pragma unroll
        for (int y = 0; y < 8; y++) e[y] = (float)((px[2 * x + (y >> 2)] >> (8 * (y & 3))) & 0xffu);
        dct8(e[0], e[1], e[2], e[3], e[4], e[5], e[6], e[7], o);
        t0[x] = o[0];
    }
    float o[8];
    dct8_quant<CHROMA, 0>(t0[0], t0[1], t0[2], t0[3], t0[4], t0[5], t0[6], t0[7], o);
    const float s = dct_scale(0, 0) * (o[0] - 8192.0f);
    return (int)div_const(s, (float)quant<CHROMA>(0), rcp_rn(quant<CHROMA>(0)));
}

template <bool CHROMA>
__device__ __forceinline__ int dc_real_strict(const uint32_t (&px)[16]) {
    const float inv_sqrt2 = 1.0f / 1.41421356237309505f;
    float sum = 0.0f;
#pragma unroll
    for (int x = 0; x < 8; x++)
#pragma unroll
        for (int y = 0; y < 8; y++) {
            const float t = sample(px, x, y) * cosv(0, x);
            const float p = t * cosv(0, y);
            sum = sum + p;
        }
    const float su = 0.25f * inv_sqrt2;
    return quantise_real(sum, su * inv_sqrt2, (float)quant<CHROMA>(0));
}

template <bool CHROMA>
__device__ __forceinline__ int dc_bin(const uint32_t (&px)[16]) {
    int a[64];
#pragma unroll
    for (int r = 0; r < 8; r++)
#pragma unroll
        for (int c = 0; c < 8; c++) a[r * 8 + c] = sample_i(px, r, c);
#pragma unroll
    for (int i = 0; i < 8; i++)
        bin8(a[i * 8 + 0], a[i * 8 + 1], a[i * 8 + 2], a[i * 8 + 3], a[i * 8 + 4], a[i * 8 + 5], a[i * 8 + 6],
             a[i * 8 + 7]);
    bin8(a[0], a[8], a[16], a[24], a[32], a[40], a[48], a[56]);
    return a[0] / quant<CHROMA>(0);
}

template <int ALGO, bool CHROMA>
__device__ __forceinline__ int transform_dc(const uint32_t (&px)[16]) {
    if constexpr (ALGO == 0) return dc_real<CHROMA>(px);
    else if constexpr (ALGO == 1) return dc_bin<CHROMA>(px);
    else return dc_real_strict<CHROMA>(px);
}

template <int ALGO, bool CHROMA>
__device__ __forceinline__ void transform_block(const uint32_t (&px)[16], uint32_t *slot) {
    if constexpr (ALGO == 0) transform_real<CHROMA>(px, slot);
    else if constexpr (ALGO == 1) transform_bin<CHROMA>(px, slot);
    else transform_real_strict<CHROMA>(px, slot);
}

// BT.601 full-range conversion in the reference's f32 evaluation order
// (src/utils/colorspace.rs:5-15), `as u8` = truncate toward zero and saturate.
__device__ __forceinline__ void rgb_to_ycbcr_f(float r, float g, float b, float &fy, float &fb, float &fr) {
    fy = 0.299f * r + 0.587f * g;
    fy = fy + 0.114f * b;
    fb = 128.0f - 0.168736f * r;
    fb = fb - 0.331264f * g;
    fb = fmaf(0.5f, b, fb);        // 0.5*b is exact, so one rounding either way
    fr = fmaf(0.5f, r, 128.0f);    // exact: k/2 + 128 is representable
    fr = fr - 0.418688f * g;
    fr = fr - 0.081312f * b;
}

__device__ __forceinline__ void rgb_to_ycbcr(float r, float g, float b, uint32_t &y, uint32_t &cb, uint32_t &cr) {
    float fy, fb, fr;
    rgb_to_ycbcr_f(r, g, b, fy, fb, fr);
    // `as u8`: truncation; the three values lie in [0, 256) for every input, so nothing saturates
    y = (uint32_t)fy;
    cb = (uint32_t)fb;
    cr = (uint32_t)fr;
}

// f32 -> u32 (truncation) written straight into byte N of a dword: conversion and packing in one
// instruction; byte 0 starts the dword (the other bytes become 0).
template <int N>
__device__ __forceinline__ void cvt_pack8(uint32_t &d, float x) {
    if (N == 0) asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD" : "=v"(d) : "v"(x));
    else if (N == 1) asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(d) : "v"(x));
    else if (N == 2) asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(d) : "v"(x));
    else asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(d) : "v"(x));
}


}  // namespace jpg
