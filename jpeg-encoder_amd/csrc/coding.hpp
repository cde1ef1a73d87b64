// Device-side pieces of the entropy coder shared by the kernels that code blocks
// (entropy.hip: from coefficients in memory; coded.hip: straight from the transform):
// Huffman table, wave scan, bit writers and the per-block coder
// (src/jpeg/entropy_coding.rs:144-227, src/utils/bitvec_utils.rs:3-8).
#pragma once
#include "../../include/jpeg_hip.h"
#include "internal.hpp"
#include "tables.h"

namespace jpg {

// (one copy per translation unit: device symbols are not linked across objects)
static __device__ const jpgtab::HuffLut HUFF_DEV{};
constexpr int LUT_WORDS = 2 * jpgtab::HUFF_LUT_STRIDE;
// inclusive prefix sum across the 64 lanes of a wavefront (DPP row shifts and broadcasts)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1,3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2,3
    return v;
}

// Errors are recorded as -(epoch << 8 | code) with atomicMin, so a newer launch always
// overrides an older one and the word never needs clearing between launches.
__device__ __forceinline__ void raise_error(DevStatus *st, uint32_t epoch, int status_code) {
    atomicMin(&st->error, -(int)((epoch << 8) | (uint32_t)(-status_code)));
}

// MSB-first bit writer into a zeroed LDS word array (words may be shared with
// neighbouring blocks, hence OR)
struct BitWriter {
    uint32_t *out;
    uint32_t wpos;
    uint64_t acc;   // the low `n` bits are pending
    uint32_t n;
    __device__ __forceinline__ void init(uint32_t *o, uint32_t bitpos) {
        out = o; wpos = bitpos >> 5; n = bitpos & 31; acc = 0;
    }
    // len <= 32, value < 2^len, n < 32 on entry
    __device__ __forceinline__ void put(uint32_t value, uint32_t len) {
        acc = (acc << len) | value;
        n += len;
        if (n >= 32) {
            n -= 32;
            atomicOr(&out[wpos++], (uint32_t)(acc >> n));
        }
    }
    __device__ __forceinline__ void finish() {
        if (n) atomicOr(&out[wpos], (uint32_t)(acc << (32 - n)));
    }
};

// Collects the last 128 bits put; a block of at most 128 bits is then complete in registers.
struct AccWriter {
    uint64_t hi = 0, lo = 0;
    // 2 <= len <= 26
    __device__ __forceinline__ void put(uint32_t value, uint32_t len) {
        hi = (hi << len) | (lo >> (64 - len));
        lo = (lo << len) | value;
    }
    __device__ __forceinline__ void finish() {}
    // ORs the n collected bits (1 <= n <= 128) into `out` starting at bit `pos`
    __device__ __forceinline__ void emit(uint32_t *out, uint32_t pos, uint32_t n) const {
        uint64_t H, L;
        if (n <= 64) {
            H = lo << (64 - n);
            L = 0;
        } else {
            const uint32_t s = 128 - n;                          // 0..63
            H = (hi << s) | (s ? lo >> (64 - s) : 0);
            L = lo << s;
        }
        const uint32_t w0 = (uint32_t)(H >> 32), w1 = (uint32_t)H, w2 = (uint32_t)(L >> 32), w3 = (uint32_t)L;
        const uint32_t sh = pos & 31;
        uint32_t *o = out + (pos >> 5);
        const uint32_t x0 = w0 >> sh, x1 = __funnelshift_r(w1, w0, sh), x2 = __funnelshift_r(w2, w1, sh),
                       x3 = __funnelshift_r(w3, w2, sh), x4 = __funnelshift_r(0u, w3, sh);
        atomicOr(&o[0], x0);
        if (x1) atomicOr(&o[1], x1);
        if (x2) atomicOr(&o[2], x2);
        if (x3) atomicOr(&o[3], x3);
        if (x4) atomicOr(&o[4], x4);
    }
};
constexpr uint32_t ACC_BITS = 128;

// Codes one block held in LDS at `slot` (64 int16, zigzag order).  mask: bit k
// set where coefficient k is non-zero (k >= 1).  Returns the coded length.
template <class W>
__device__ __forceinline__ uint32_t code_block(const uint16_t *slot, uint32_t mlo, uint32_t mhi, int dc_diff,
                                               const uint32_t *lut, DevStatus *st, uint32_t epoch, W &w) {
    // DC: size = bit length of |diff| (src/jpeg/entropy_coding.rs:127-134,149-160)
    uint32_t mag = (uint32_t)(dc_diff < 0 ? -dc_diff : dc_diff);
    uint32_t size = 32 - __clz(mag);
    if (size > 11) {
        raise_error(st, epoch, JPG_ERR_DC_RANGE);
        size = 0;
    }
    // negative values: v + (1 << size) - 1 (src/jpeg/entropy_coding.rs:140-142)
    uint32_t amp = (uint32_t)(dc_diff < 0 ? dc_diff - 1 : dc_diff) & ((1u << size) - 1u);
    uint32_t t = lut[(size << 4) | 15];                              // jpgtab::huff_dc_index
    uint32_t bits = (t >> 16) + size;
    w.put(((t & 0xffff) << size) | amp, (t >> 16) + size);
    const uint32_t zrl = lut[0xf0];
    int prevk = 0;
    uint64_t m = ((uint64_t)mhi << 32) | mlo;
    // the coefficient of the next symbol is read while the current one is coded
    int k = m ? __ffsll((unsigned long long)m) - 1 : 0;
    int v = (int16_t)slot[k];
    while (m) {
        m &= m - 1;
        const int kn = m ? __ffsll((unsigned long long)m) - 1 : 0;
        const int vn = (int16_t)slot[kn];
        uint32_t run = (uint32_t)(k - prevk - 1);
        prevk = k;
        while (run >= 16) {                                      // ZRL per 16 zeros (src/jpeg/entropy_coding.rs:179-185)
            w.put(zrl & 0xffff, zrl >> 16);
            bits += zrl >> 16;
            run -= 16;
        }
        mag = (uint32_t)(v < 0 ? -v : v);
        size = 32 - __clz(mag);
        if (size > 10) {
            raise_error(st, epoch, JPG_ERR_AC_RANGE);
            size = 0;
        }
        amp = (uint32_t)(v < 0 ? v - 1 : v) & ((1u << size) - 1u);
        t = lut[(run << 4) | size];
        bits += (t >> 16) + size;
        w.put(((t & 0xffff) << size) | amp, (t >> 16) + size);
        k = kn;
        v = vn;
    }
    if (prevk != 63) {                                           // trailing zeros -> EOB (src/jpeg/entropy_coding.rs:171-177)
        t = lut[0x00];
        bits += t >> 16;
        w.put(t & 0xffff, t >> 16);
    }
    w.finish();
    return bits;
}

// bit 0 set when the low half of w is non-zero, bit 16 when the high half is: min(half, 1) on both
// halves at once (written out because the compiler turns the vector min into compares and selects)
__device__ __forceinline__ uint32_t halves_nonzero(uint32_t w) {
    uint32_t r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(w), "s"(0x00010001u));
    return r;
}

// acc + lo(flags) * lo(k) + hi(flags) * hi(k): with flags 0/1 in the halves of a word and two powers of two in k,
// both flags land on their bits of acc in one instruction (written out: the compiler has no pattern for it)
__device__ __forceinline__ uint32_t flags_place(uint32_t acc, uint32_t flags, uint32_t k) {
    uint32_t r;
    asm("v_dot2_u32_u16 %0, %1, %2, %3" : "=v"(r) : "v"(flags), "s"(k), "v"(acc));
    return r;
}

// bit k set where the k-th int16 of the block (8 uint4) is non-zero: the block is four groups of eight dwords; dword
// t of a group holds coefficients 2t and 2t+1 of the group, whose flags go to bits 2t and 2t+1 of the group's 16
// bits (no interleave: the constant of dword t is 2^(2t) in its low and 2^(2t+1) in its high half)
__device__ __forceinline__ void nonzero_mask(const uint4 (&q)[8], uint32_t &mlo, uint32_t &mhi) {
    uint32_t g[4];
#pragma unroll
    for (int h = 0; h < 4; h++) {
        uint32_t f = 0;
#pragma unroll
        for (int t = 0; t < 8; t++) {
            const uint4 &v = q[2 * h + (t >> 2)];
            const uint32_t w = (t & 3) == 0 ? v.x : ((t & 3) == 1 ? v.y : ((t & 3) == 2 ? v.z : v.w));
            f = flags_place(f, halves_nonzero(w), (1u << (2 * t)) | (1u << (2 * t + 17)));
        }
        g[h] = f;
    }
    mlo = g[0] | (g[1] << 16);
    mhi = g[2] | (g[3] << 16);
}

// block `jb` of its MCU (0..nb-1): the block before it in the same component's DC chain,
// relative to the block itself (src/jpeg/entropy_coding.rs:74-124)
__device__ __forceinline__ int prev_distance(int jb, int nb, int hv) {
    return jb >= hv ? nb : (jb > 0 ? 1 : nb - hv + 1);
}

}  // namespace jpg
