// CPU proof of the f32 formulation of the integer lifting transform (csrc/transform.hpp, namespace binf),
// which replaces i32 shifts and the i32 division of src/jpeg/dct_quant.rs:67-187 of the reference.
//
//   1. every multiply-add form the device code uses -- fma(n + DX*M, NUM/8, K) with M = 1.5*2^23 and
//      K = M*(8*DH - NUM*DX)/8 + OFF under round-toward-minus-infinity -- equals DH*M + floor(NUM*n/8) + OFF
//      for every integer |n| <= 2^18 (exhaustive);
//   2. trunc(a * z) with z = the smallest f32 >= 1/q equals a / q (C integer division, which truncates like
//      Rust's) for every |a| <= 2^16, every divisor of the two quantisation tables and all four rounding modes
//      of the multiplication (exhaustive);
//   3. the whole 8x8 transform written with those forms (same operation order and the same domains as the device
//      code) equals the integer code on random and extreme blocks, and the intermediate values stay far inside the
//      range the two proofs above cover.
//
// build and run:  gcc -O1 -frounding-math -ffp-contract=off -o /tmp/check_bin_f32 tools/check_bin_f32.c -lm && /tmp/check_bin_f32
#include <fenv.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static const int QT_LUMA[64] = {16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,
                                14, 13, 16, 24, 40,  57,  69,  56,  14, 17, 22, 29, 51,  87,  80,  62,
                                18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
                                49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
static const int QT_CHROMA[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99,
                                  24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
                                  99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                                  99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
static const float M = 12582912.0f;

static int floordiv8(long long x) { return (int)((x >= 0) ? x / 8 : -((-x + 7) / 8)); }

static float Kconst(int DH, int NUM, int DX, int OFF) { return (float)((double)M * (8 * DH - NUM * DX) / 8.0 + OFF); }
// floor(NUM*x/8) + OFF in domain DH from x in domain DX (round-down mode in force)
static float fl(int DH, int NUM, int OFF, int DX, float x) {
    volatile float c = NUM / 8.0f, k = Kconst(DH, NUM, DX, OFF), xv = x;
    return fmaf(xv, c, k);
}

static long long max_abs = 0;
static void note(float v, int D) {
    double n = (double)v - D * (double)M;
    long long a = llabs((long long)n);
    if (a > max_abs) max_abs = a;
}

// mirrors binf::bin8f<DI, OFF, PLAIN>
static void bin8f(int DI, int OFF, int PLAIN, float *r0, float *r1, float *r2, float *r3, float *r4, float *r5, float *r6, float *r7) {
    volatile float a0 = *r0, a1 = *r1, a2 = *r2, a3 = *r3, a4 = *r4, a5 = *r5, a6 = *r6, a7 = *r7;
    volatile float x7_1 = a0 - a7, x6_1 = a1 - a6, x5_1 = a2 - a5, x4_1 = a3 - a4;          // domain 0
    volatile float y6 = x6_1 + fl(1, 3, 0, 0, x5_1);                                          // +1
    volatile float y5 = fl(1, 5, 0, 1, y6) - x5_1;                                            // +1
    volatile float x4_2 = x4_1 + y5;                                                          // +1
    volatile float x5_2 = x4_1 - y5;                                                          // -1
    volatile float x6_2 = x7_1 - y6;                                                          // -1
    volatile float x7_2 = x7_1 + y6;                                                          // +1
    volatile float o7 = x4_2 - fl(1, 1, 0, 1, x7_2);                                          // 0
    volatile float o5 = x5_2 + fl(1, 7, 0, -1, x6_2);                                         // 0
    volatile float o3 = x6_2 - fl(-1, 4, 0, 0, o5);                                           // 0
    note(y6, 1); note(y5, 1); note(x4_2, 1); note(x5_2, -1); note(x6_2, -1); note(x7_2, 1); note(o7, 0); note(o5, 0); note(o3, 0);
    volatile float o0, o2, o4, o6;
    if (DI != 0) {
        volatile float x0_1 = a0 - fl(DI, 4, OFF, 0, x7_1), x1_1 = a1 - fl(DI, 4, OFF, 0, x6_1);
        volatile float x2_1 = a2 - fl(DI, 4, OFF, 0, x5_1), x3_1 = a3 - fl(DI, 4, OFF, 0, x4_1);
        volatile float e0 = x0_1 + x3_1, e3 = x0_1 - x3_1, e1 = x1_1 + x2_1, e2 = x1_1 - x2_1;
        o0 = e0 + e1;
        o4 = fl(1, 4, 0, 0, o0) - e1;                 // +1
        o6 = e2 - fl(-1, 3, 0, 0, e3);                // +1
        o2 = e3 + fl(1, 3, 0, 1, o6);                 // +1
        note(o0, 0); note(o4, 1); note(o6, 1); note(o2, 1);
        if (PLAIN) { o2 = o2 - M; o4 = o4 - M; o6 = o6 - M; }
    } else {
        volatile float x0_1 = (a0 - fl(1, 4, 0, 0, x7_1)) + M, x1_1 = (a1 - fl(1, 4, 0, 0, x6_1)) + M;   // 0
        volatile float x2_1 = a2 - fl(1, 4, 0, 0, x5_1);                                                  // -1
        volatile float x3_1 = a3 - fl(-1, 4, 0, 0, x4_1);                                                 // +1
        volatile float e0 = x0_1 + x3_1;   // +1
        volatile float e3 = x0_1 - x3_1;   // -1
        volatile float e1 = x1_1 + x2_1;   // -1
        volatile float e2 = x1_1 - x2_1;   // +1
        o0 = e0 + e1;                                  // 0
        o4 = fl(-1, 4, 0, 0, o0) - e1;                 // 0
        o6 = e2 - fl(1, 3, 0, -1, e3);                 // 0
        o2 = e3 + fl(1, 3, 0, 0, o6);                  // 0
        note(o0, 0); note(o4, 0); note(o6, 0); note(o2, 0);
    }
    *r0 = o0; *r2 = o2; *r4 = o4; *r6 = o6;
    *r1 = PLAIN ? x7_2 - M : x7_2;
    *r3 = o3; *r5 = o5; *r7 = o7;
}

static int row_out_domain(int k) { return (k == 1 || k == 2 || k == 4 || k == 6) ? 1 : 0; }

static float rcp_up(int q) {
    fesetround(FE_TONEAREST);
    volatile float z = 1.0f / (float)q;
    if ((double)z * (double)q < 1.0) z = nextafterf(z, 2.0f);
    return z;
}

// the reference's integer butterfly (src/jpeg/dct_quant.rs:93-129)
static void bin8i(int *a0, int *a1, int *a2, int *a3, int *a4, int *a5, int *a6, int *a7) {
    int x7_1 = *a0 - *a7, x0_1 = *a0 - (x7_1 >> 1);
    int x6_1 = *a1 - *a6, x1_1 = *a1 - (x6_1 >> 1);
    int x5_1 = *a2 - *a5, x2_1 = *a2 - (x5_1 >> 1);
    int x4_1 = *a3 - *a4, x3_1 = *a3 - (x4_1 >> 1);
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
    *a0 = x0_2; *a4 = x1_2; *a6 = x2_2; *a2 = x3_2;
    *a7 = x4_2; *a5 = x5_2; *a3 = x6_2; *a1 = x7_2;
}

static void block_int(const uint8_t *px, const int *qt, int *out) {
    int a[64];
    for (int i = 0; i < 64; i++) a[i] = (int)px[i] - 128;
    for (int i = 0; i < 8; i++) bin8i(&a[i * 8], &a[i * 8 + 1], &a[i * 8 + 2], &a[i * 8 + 3], &a[i * 8 + 4], &a[i * 8 + 5], &a[i * 8 + 6], &a[i * 8 + 7]);
    for (int i = 0; i < 8; i++) bin8i(&a[i], &a[8 + i], &a[16 + i], &a[24 + i], &a[32 + i], &a[40 + i], &a[48 + i], &a[56 + i]);
    for (int n = 0; n < 64; n++) out[n] = a[n] / qt[n];
}

static void block_f32(const uint8_t *px, const float *z, int *out) {
    fesetround(FE_DOWNWARD);
    float t[64];
    for (int r = 0; r < 8; r++) {
        for (int c = 0; c < 8; c++) t[r * 8 + c] = M + (float)px[r * 8 + c];   // what v_perm_b32 assembles: exact
        bin8f(1, 128, 0, &t[r * 8], &t[r * 8 + 1], &t[r * 8 + 2], &t[r * 8 + 3], &t[r * 8 + 4], &t[r * 8 + 5], &t[r * 8 + 6], &t[r * 8 + 7]);
    }
    for (int v = 0; v < 8; v++) {
        bin8f(row_out_domain(v), 0, 1, &t[v], &t[8 + v], &t[16 + v], &t[24 + v], &t[32 + v], &t[40 + v], &t[48 + v], &t[56 + v]);
        for (int u = 0; u < 8; u++) {
            volatile float q = t[u * 8 + v] * z[u * 8 + v];
            out[u * 8 + v] = (int)q;   // truncation, like v_cvt_i32_f32
        }
    }
    fesetround(FE_TONEAREST);
}

static uint64_t rng = 0x9E3779B97F4A7C15ull;
static uint32_t rnd(void) {
    rng += 0x9E3779B97F4A7C15ull;
    uint64_t z = rng;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (uint32_t)((z ^ (z >> 31)) >> 16);
}

int main(void) {
    long long bad = 0, n1 = 0, n2 = 0;
    // 1. the multiply-add forms
    fesetround(FE_DOWNWARD);
    const int forms[][4] = {// DH, NUM, DX, OFF -- every fl<> of binf::bin8f
        {1, 3, 0, 0}, {1, 5, 1, 0}, {1, 1, 1, 0}, {1, 7, -1, 0}, {-1, 4, 0, 0}, {1, 4, 0, 128}, {1, 4, 0, 0},
        {-1, 3, 0, 0}, {1, 3, 1, 0}, {1, 3, -1, 0}, {-1, 4, 0, 128}};
    for (unsigned f = 0; f < sizeof forms / sizeof forms[0]; f++) {
        const int DH = forms[f][0], NUM = forms[f][1], DX = forms[f][2], OFF = forms[f][3];
        const float K = Kconst(DH, NUM, DX, OFF);
        if ((double)K != (double)M * (8 * DH - NUM * DX) / 8.0 + OFF) { printf("constant of form %u is not exact\n", f); bad++; }
        for (int n = -(1 << 18); n <= (1 << 18); n++) {
            const float x = (float)((double)n + DX * (double)M);
            const float got = fl(DH, NUM, OFF, DX, x);
            const double want = DH * (double)M + floordiv8((long long)NUM * n) + OFF;
            n1++;
            if ((double)got != want) { if (bad < 10) printf("form %u n %d: got %.1f want %.1f\n", f, n, got, want); bad++; }
        }
    }
    // 2. the reciprocal quantiser
    const int modes[4] = {FE_TONEAREST, FE_DOWNWARD, FE_UPWARD, FE_TOWARDZERO};
    for (int tb = 0; tb < 2; tb++)
        for (int i = 0; i < 64; i++) {
            const int q = tb ? QT_CHROMA[i] : QT_LUMA[i];
            const float z = rcp_up(q);
            for (int m = 0; m < 4; m++) {
                fesetround(modes[m]);
                for (int a = -65536; a <= 65536; a++) {
                    volatile float fa = (float)a, fz = z;
                    volatile float p = fa * fz;
                    n2++;
                    if ((int)p != a / q) { if (bad < 10) printf("q %d a %d mode %d: %d != %d\n", q, a, m, (int)p, a / q); bad++; }
                }
            }
        }
    fesetround(FE_TONEAREST);
    // 3. whole blocks
    float zl[64], zc[64];
    for (int i = 0; i < 64; i++) { zl[i] = rcp_up(QT_LUMA[i]); zc[i] = rcp_up(QT_CHROMA[i]); }
    long long nblk = 0;
    uint8_t px[64];
    int wi[64], wf[64];
    for (int it = 0; it < 400000; it++) {
        const int kind = it % 8;
        for (int i = 0; i < 64; i++) {
            switch (kind) {
                case 0: px[i] = (uint8_t)rnd(); break;                                   // noise
                case 1: px[i] = (rnd() & 1) ? 255 : 0; break;                            // extremes
                case 2: px[i] = ((i ^ (i >> 3)) & 1) ? 255 : 0; break;                   // checkerboard (+ noise below)
                case 3: px[i] = (uint8_t)(128 + 100 * sin(i * 0.37 + it)); break;        // smooth
                case 4: px[i] = (i == (int)(it / 8 % 64)) ? 255 : 0; break;              // impulse
                case 5: px[i] = (uint8_t)((i % 8) * 36 + (rnd() & 3)); break;            // ramps
                case 6: px[i] = (uint8_t)(((i / 8) & 1) ? 255 - (rnd() & 7) : (rnd() & 7)); break;   // stripes
                default: px[i] = (uint8_t)(it >> 3); break;                              // flat
            }
        }
        if (kind == 2 && (it & 8)) for (int i = 0; i < 64; i++) px[i] ^= (uint8_t)(rnd() & 1);
        for (int tb = 0; tb < 2; tb++) {
            block_int(px, tb ? QT_CHROMA : QT_LUMA, wi);
            block_f32(px, tb ? zc : zl, wf);
            nblk++;
            if (memcmp(wi, wf, sizeof wi)) {
                if (bad < 10) { printf("block %d table %d differs:", it, tb); for (int i = 0; i < 64; i++) if (wi[i] != wf[i]) printf(" [%d] %d != %d", i, wf[i], wi[i]); printf("\n"); }
                bad++;
            }
        }
    }
    printf("multiply-add forms: %lld cases; quantiser: %lld cases; blocks: %lld; largest |value| inside the transform: %lld; mismatches: %lld\n",
           n1, n2, nblk, max_abs, bad);
    return bad ? 1 : 0;
}
