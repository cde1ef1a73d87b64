/*
 * check_divide.c -- exhaustive check of the 3-operation division the real-dct
 * kernels use for quantisation.
 *
 * The reference quantises with `((scale*sum) / q) as i16` (src/jpeg/dct_quant.rs:
 * 227-230): an IEEE f32 division followed by truncation.  The kernels replace
 * the division by a constant q with
 *     q0 = s * z;  r = fma(-q0, q, s);  d = fma(r, z, q0);     z = RN(1/q)
 * which is the classic correctly-rounded-quotient sequence.  This program
 * checks d == s / q bit for bit for EVERY f32 s with lo <= |s| <= hi and every
 * divisor that occurs in the two quantisation tables.
 *
 * build: gcc -O2 -mfma -fopenmp -ffp-contract=off tools/check_divide.c -o /tmp/check_divide -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static const int QS[] = {10, 11, 12, 13, 14, 16, 17, 18, 19, 21, 22, 24, 26, 29, 35, 37, 40, 47, 49, 51, 55, 56, 57,
                         58, 60, 61, 62, 64, 66, 68, 69, 72, 77, 78, 80, 81, 87, 92, 95, 98, 99, 100, 101, 103, 104,
                         109, 112, 113, 120, 121};

static inline float bits2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

int main(int argc, char **argv) {
    float lo = argc > 1 ? (float)atof(argv[1]) : 0.5f;
    float hi = argc > 2 ? (float)atof(argv[2]) : 4096.0f;
    long long bad_total = 0, n_total = 0;
    for (size_t qi = 0; qi < sizeof QS / sizeof QS[0]; qi++) {
        const float q = (float)QS[qi];
        const float z = 1.0f / q;
        long long bad = 0, n = 0;
        const uint32_t b0 = f2bits(lo), b1 = f2bits(hi);
#pragma omp parallel for reduction(+ : bad, n) schedule(static)
        for (uint32_t b = b0; b <= b1; b++) {
            for (int sign = 0; sign < 2; sign++) {
                const float s = bits2f(b | ((uint32_t)sign << 31));
                const float ref = s / q;
                const float q0 = s * z;
                const float r = fmaf(-q0, q, s);
                const float d = fmaf(r, z, q0);
                n++;
                if (f2bits(d) != f2bits(ref)) bad++;
            }
        }
        if (bad) printf("q=%d: %lld mismatches of %lld\n", QS[qi], bad, n);
        bad_total += bad;
        n_total += n;
    }
    printf("checked %lld quotients, %lld mismatches\n", n_total, bad_total);
    return bad_total != 0;
}
