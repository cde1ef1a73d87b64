// Host arithmetic of the stripe mode (one large image cut between MCU rows, one stripe per
// GPU): which rows a stripe gets, and where its bits go in the whole scan.  The reference codes
// an image serially (src/jpeg/entropy_coding.rs:16-125); this is the bookkeeping that lets
// independent coders reproduce that stream.  Declared in include/jpeg_host.h.
#include "jpeg_host.h"

extern "C" int jph_plan_stripes(int32_t height, int v, int n, int32_t *y0, int32_t *y1) {
    if (height <= 0 || (v != 1 && v != 2) || n <= 0 || !y0 || !y1) return JPH_ERR_ARG;
    const int32_t mcu_h = 8 * v;
    const int32_t mcu_rows = (height + mcu_h - 1) / mcu_h;
    const int32_t base = mcu_rows / n, extra = mcu_rows % n;
    int32_t r = 0;
    for (int k = 0; k < n; k++) {
        const int32_t cnt = base + (k < extra ? 1 : 0);
        const int64_t a = (int64_t)r * mcu_h, b = (int64_t)(r + cnt) * mcu_h;
        y0[k] = (int32_t)(a < height ? a : height);
        y1[k] = (int32_t)(b < height ? b : height);
        r += cnt;
    }
    return JPH_OK;
}

extern "C" int jph_stitch_plan(int n, const uint64_t *nbits, const uint32_t *tails, uint64_t *offsets, uint8_t *heads) {
    if (n <= 0 || !nbits || !tails || !offsets || !heads) return JPH_ERR_ARG;
    uint64_t total = 0;
    for (int k = 0; k < n; k++) {
        offsets[k] = total;
        const unsigned r = (unsigned)(total % 8);
        uint32_t acc = 0;
        unsigned got = 0;
        for (int j = k - 1; got < r && j >= 0; j--) {   // walk back over (possibly tiny) stripes
            unsigned take = r - got;
            if (nbits[j] < take) take = (unsigned)nbits[j];
            acc |= (tails[j] & ((1u << take) - 1u)) << got;
            got += take;
        }
        heads[k] = r ? (uint8_t)((acc << (8 - r)) & 0xffu) : 0;
        total += nbits[k];
    }
    return JPH_OK;
}
