// Entropy coding on the device: DC-difference / AC run-length symbols
// (src/jpeg/entropy_coding.rs:144-199), Huffman codes + amplitude bits
// (src/jpeg/entropy_coding.rs:201-227), MSB-first bit packing
// (src/utils/bitvec_utils.rs:3-8) and 0xFF byte stuffing
// (src/jpeg/file.rs:92-103).  The reference does this serially; here the
// variable-length output is placed by prefix sums:
//
//   k_entropy<false>  one thread per block: coded length of the block; one
//                     wavefront per chunk of 64 blocks: length of the chunk
//   k_scan_chunks     exclusive scan of the chunk lengths (one workgroup); also
//                     clears the first and last word of every chunk, which
//                     neighbouring chunks share and therefore OR into
//   k_entropy<true>   codes the blocks again, assembles each chunk's bits in
//                     LDS at their final alignment and writes them out
//   k_ff_count / k_ff_scan / k_ff_scatter   the same three steps for stuffing
//
// A thread walks only the non-zero coefficients of its block (a 64-bit mask
// built while the block is read from LDS), so the work follows the number of
// symbols, not the number of coefficients.
#include "../../include/jpeg_hip.h"
#include "internal.hpp"
#include "tables.h"

namespace jpg {

__device__ const jpgtab::HuffLut HUFF_DEV{};
constexpr int LUT_WORDS = 2 * jpgtab::HUFF_LUT_STRIDE;
constexpr int ENT_THREADS = 256;
constexpr int ENT_WAVES = ENT_THREADS / 64;
constexpr int SLOT_BYTES = 144;                 // 128-byte block + 16: conflict-free 16-byte LDS access
constexpr int OUT_WORDS = 1024;                 // LDS words a chunk's bits may take on the fast path (4 KiB)

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

struct EntArgs {
    const int16_t *coef;
    long long nblocks;
    int nb, hv;
    int16_t pred[3];
    uint16_t *blk_bits;
    uint32_t *chunk_len;
    const uint64_t *chunk_base;
    uint32_t *bits;
    DevStatus *status;
};

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

// same interface, every word OR-ed straight into global memory (chunks whose
// bits do not fit the LDS window)
struct BitWriterGlobal {
    uint32_t *out;
    uint64_t wpos;
    uint64_t acc;
    uint32_t n;
    __device__ __forceinline__ void init(uint32_t *o, uint64_t bitpos) {
        out = o; wpos = bitpos >> 5; n = (uint32_t)(bitpos & 31); acc = 0;
    }
    __device__ __forceinline__ void put(uint32_t value, uint32_t len) {
        acc = (acc << len) | value;
        n += len;
        if (n >= 32) {
            n -= 32;
            atomicOr(&out[wpos++], __builtin_bswap32((uint32_t)(acc >> n)));
        }
    }
    __device__ __forceinline__ void finish() {
        if (n) atomicOr(&out[wpos], __builtin_bswap32((uint32_t)(acc << (32 - n))));
    }
};

struct NullWriter {
    __device__ __forceinline__ void put(uint32_t, uint32_t) {}
    __device__ __forceinline__ void finish() {}
};

// Codes one block held in LDS at `slot` (64 int16, zigzag order).  mask: bit k
// set where coefficient k is non-zero (k >= 1).  Returns the coded length.
template <class W>
__device__ __forceinline__ uint32_t code_block(const uint16_t *slot, uint32_t mlo, uint32_t mhi, int dc_diff,
                                               const uint32_t *lut, int *error, W &w) {
    // DC: size = bit length of |diff| (src/jpeg/entropy_coding.rs:127-134,149-160)
    uint32_t mag = (uint32_t)(dc_diff < 0 ? -dc_diff : dc_diff);
    uint32_t size = 32 - __clz(mag);
    if (size > 11) {
        atomicMin(error, (int)JPG_ERR_DC_RANGE);
        size = 0;
    }
    // negative values: v + (1 << size) - 1 (src/jpeg/entropy_coding.rs:140-142)
    uint32_t amp = (uint32_t)(dc_diff < 0 ? dc_diff - 1 : dc_diff) & ((1u << size) - 1u);
    uint32_t t = lut[256 + size];
    uint32_t bits = (t >> 16) + size;
    w.put(((t & 0xffff) << size) | amp, (t >> 16) + size);
    const uint32_t zrl = lut[0xf0];
    int prevk = 0;
#pragma unroll
    for (int half = 0; half < 2; half++) {
        uint32_t m = half ? mhi : mlo;
        while (m) {
            const int k = 32 * half + (__ffs(m) - 1);
            m &= m - 1;
            const int v = (int16_t)slot[k];
            uint32_t run = (uint32_t)(k - prevk - 1);
            prevk = k;
            while (run >= 16) {                                  // ZRL per 16 zeros (src/jpeg/entropy_coding.rs:179-185)
                w.put(zrl & 0xffff, zrl >> 16);
                bits += zrl >> 16;
                run -= 16;
            }
            mag = (uint32_t)(v < 0 ? -v : v);
            size = 32 - __clz(mag);
            if (size > 10) {
                atomicMin(error, (int)JPG_ERR_AC_RANGE);
                size = 0;
            }
            amp = (uint32_t)(v < 0 ? v - 1 : v) & ((1u << size) - 1u);
            t = lut[(run << 4) | size];
            bits += (t >> 16) + size;
            w.put(((t & 0xffff) << size) | amp, (t >> 16) + size);
        }
    }
    if (prevk != 63) {                                           // trailing zeros -> EOB (src/jpeg/entropy_coding.rs:171-177)
        t = lut[0x00];
        bits += t >> 16;
        w.put(t & 0xffff, t >> 16);
    }
    w.finish();
    return bits;
}

// bit k set where the k-th int16 of the block (8 uint4) is non-zero
__device__ __forceinline__ void nonzero_mask(const uint4 (&q)[8], uint32_t &mlo, uint32_t &mhi) {
    uint32_t m[2] = {0, 0};
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t w[4] = {q[i].x, q[i].y, q[i].z, q[i].w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int k = 8 * i + 2 * j;
            const uint32_t lo = (w[j] & 0xffffu) ? 1u : 0u, hi = (w[j] >> 16) ? 1u : 0u;
            m[k >> 5] |= (lo << (k & 31)) | (hi << ((k + 1) & 31));
        }
    }
    mlo = m[0];
    mhi = m[1];
}

template <bool PACK>
__global__ __launch_bounds__(ENT_THREADS) void k_entropy(EntArgs a) {
    __shared__ uint32_t sLut[LUT_WORDS];
    __shared__ __attribute__((aligned(16))) uint8_t sCoef[ENT_WAVES][CHUNK_BLOCKS * SLOT_BYTES];
    __shared__ uint32_t sOut[PACK ? ENT_WAVES : 1][PACK ? OUT_WORDS + 2 : 1];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < LUT_WORDS; i += ENT_THREADS) sLut[i] = (&HUFF_DEV.e[0][0])[i];
    __syncthreads();

    const long long chunk = (long long)blockIdx.x * ENT_WAVES + wave;
    const long long first = chunk * CHUNK_BLOCKS;
    if (first >= a.nblocks) return;                               // wave-uniform; no barrier follows
    const int nvalid = (int)min((long long)CHUNK_BLOCKS, a.nblocks - first);

    // stage the chunk's coefficients: coalesced 16-byte loads, one padded slot per block
    const uint4 *src = reinterpret_cast<const uint4 *>(a.coef) + first * 8;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int u = i * 64 + lane;                              // 16-byte piece inside the chunk
        if ((u >> 3) < nvalid)
            *reinterpret_cast<uint4 *>(&sCoef[wave][(u >> 3) * SLOT_BYTES + (u & 7) * 16]) = src[u];
    }
    __builtin_amdgcn_wave_barrier();

    const long long blk = first + lane;
    const bool valid = lane < nvalid;
    const uint16_t *slot = reinterpret_cast<const uint16_t *>(&sCoef[wave][lane * SLOT_BYTES]);
    uint32_t mlo = 0, mhi = 0;
    int dc_diff = 0;
    uint32_t chroma = 0;
    if (valid) {
        uint4 q[8];
#pragma unroll
        for (int i = 0; i < 8; i++) q[i] = reinterpret_cast<const uint4 *>(slot)[i];
        nonzero_mask(q, mlo, mhi);
        mlo &= ~1u;                                               // the DC symbol is coded separately
        // DC difference against the previous block of the same component
        // (three predictor chains, src/jpeg/entropy_coding.rs:38-40,149-150)
        const int jb = (int)(blk % a.nb);
        chroma = jb >= a.hv ? 1u : 0u;
        const long long prev = chroma ? blk - a.nb : (jb > 0 ? blk - 1 : blk - a.nb + a.hv - 1);
        const int pdc = prev >= 0 ? (int)a.coef[prev * 64] : (int)a.pred[chroma ? jb - a.hv + 1 : 0];
        dc_diff = (int16_t)((int)(int16_t)(q[0].x & 0xffff) - pdc);
    }
    const uint32_t *lut = sLut + chroma * jpgtab::HUFF_LUT_STRIDE;

    if (!PACK) {
        NullWriter w;
        const uint32_t bits = valid ? code_block(slot, mlo, mhi, dc_diff, lut, &a.status->error, w) : 0u;
        if (valid) a.blk_bits[blk] = (uint16_t)bits;
        const uint32_t sc = wave_incl_scan(bits);
        if (lane == 63) a.chunk_len[chunk] = sc;
    } else {
        const uint32_t len = valid ? (uint32_t)a.blk_bits[blk] : 0u;
        const uint32_t sc = wave_incl_scan(len);
        const uint32_t total = __shfl(sc, 63);
        const uint64_t O = a.chunk_base[chunk];
        const uint32_t shift = (uint32_t)(O & 31);
        const uint64_t W0 = O >> 5;
        const uint32_t nwords = (shift + total + 31) >> 5;
        if (nwords <= OUT_WORDS) {
            // fast path: assemble in LDS at the final bit alignment, then write whole words
            for (uint32_t i = lane; i < nwords; i += 64) sOut[wave][i] = 0;
            __builtin_amdgcn_wave_barrier();
            if (valid) {
                BitWriter w;
                w.init(sOut[wave], shift + sc - len);
                code_block(slot, mlo, mhi, dc_diff, lut, &a.status->error, w);
            }
            __builtin_amdgcn_wave_barrier();
            for (uint32_t i = lane; i < nwords; i += 64) {
                const uint32_t v = __builtin_bswap32(sOut[wave][i]);
                if (i == 0 || i == nwords - 1) atomicOr(&a.bits[W0 + i], v);   // shared with the neighbouring chunk
                else a.bits[W0 + i] = v;
            }
        } else {
            // dense chunk: clear the words this chunk owns alone, then OR every word in global memory
            for (uint32_t i = 1 + lane; i + 1 < nwords; i += 64) a.bits[W0 + i] = 0;
            __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_wave_barrier();
            if (valid) {
                BitWriterGlobal w;
                w.init(a.bits, O + sc - len);
                code_block(slot, mlo, mhi, dc_diff, lut, &a.status->error, w);
            }
        }
    }
}

// One workgroup: exclusive scan of the chunk lengths into 64-bit bit offsets,
// total into status->nbits, and zero of every chunk's first and last word.
__global__ __launch_bounds__(1024) void k_scan_chunks(const uint32_t *__restrict__ len, long long n,
                                                      uint64_t *__restrict__ base, DevStatus *__restrict__ status,
                                                      uint32_t *__restrict__ bits) {
    __shared__ uint64_t s[1024];
    const int tid = threadIdx.x;
    const long long per = (n + 1023) / 1024;
    const long long lo = min(n, tid * per), hi = min(n, lo + per);
    uint64_t sum = 0;
    for (long long i = lo; i < hi; i++) sum += len[i];
    s[tid] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const uint64_t t = tid >= d ? s[tid - d] : 0;
        __syncthreads();
        s[tid] += t;
        __syncthreads();
    }
    uint64_t run = s[tid] - sum;
    for (long long i = lo; i < hi; i++) {
        const uint32_t L = len[i];
        base[i] = run;
        if (L) {
            bits[run >> 5] = 0;
            bits[(run + L - 1) >> 5] = 0;
        }
        run += L;
    }
    if (tid == 1023) status->nbits = s[1023];
}

// ---------------------------------------------------------------- stuffing

struct StuffArgs {
    const uint32_t *in;      // unstuffed stream (whole words are defined up to the last one)
    DevStatus *status;       // nbits in, scan_bytes and tail out
    uint32_t shift;          // place the stream `shift` (0..7) bits later
    uint32_t head;           // bits OR-ed into the first byte (top `shift` bits)
    int is_last;             // keep the trailing partial byte
    uint32_t *seg_cnt;
    uint64_t *seg_off;
    uint8_t *out;
};

__device__ __forceinline__ uint64_t shifted_bytes(const StuffArgs &a) {
    const uint64_t tbits = a.status->nbits + a.shift;
    return a.is_last ? (tbits + 7) >> 3 : tbits >> 3;
}

// 16 bytes of the shifted stream starting at byte 16*q, as 4 big-endian words
__device__ __forceinline__ void load16(const StuffArgs &a, uint64_t q, uint64_t in_bytes, uint32_t (&t)[4]) {
    uint32_t w[4];
    const uint4 raw = 16 * q < in_bytes ? reinterpret_cast<const uint4 *>(a.in)[q] : make_uint4(0, 0, 0, 0);
    w[0] = __builtin_bswap32(raw.x);
    w[1] = 16 * q + 4 < in_bytes ? __builtin_bswap32(raw.y) : 0;
    w[2] = 16 * q + 8 < in_bytes ? __builtin_bswap32(raw.z) : 0;
    w[3] = 16 * q + 12 < in_bytes ? __builtin_bswap32(raw.w) : 0;
    if (a.shift) {
        uint32_t prev = 0;
        if (q > 0 && 16 * q - 4 < in_bytes) prev = __builtin_bswap32(a.in[4 * q - 1]);
        t[0] = __funnelshift_r(w[0], prev, a.shift);
        t[1] = __funnelshift_r(w[1], w[0], a.shift);
        t[2] = __funnelshift_r(w[2], w[1], a.shift);
        t[3] = __funnelshift_r(w[3], w[2], a.shift);
        if (q == 0) t[0] |= a.head << 24;
    } else {
        t[0] = w[0]; t[1] = w[1]; t[2] = w[2]; t[3] = w[3];
    }
}

// number of 0xFF bytes in a word
__device__ __forceinline__ uint32_t count_ff(uint32_t w) {
    const uint32_t z = ~w;
    uint32_t t = (z & 0x7f7f7f7fu) + 0x7f7f7f7fu;
    t = ~(t | z | 0x7f7f7f7fu);
    return __popc(t);
}

__global__ __launch_bounds__(256) void k_ff_count(StuffArgs a) {
    __shared__ uint32_t s[4];
    const uint64_t in_bytes = (a.status->nbits + 7) >> 3;
    const uint64_t nbytes = shifted_bytes(a);
    const uint64_t nseg = (nbytes + STUFF_SEG - 1) / STUFF_SEG;
    for (uint64_t seg = blockIdx.x; seg < nseg; seg += gridDim.x) {
        const uint64_t q = seg * (STUFF_SEG / 16) + threadIdx.x;
        uint32_t t[4];
        load16(a, q, in_bytes, t);
        uint32_t c = 0;
        if (16 * q < nbytes) c = count_ff(t[0]) + count_ff(t[1]) + count_ff(t[2]) + count_ff(t[3]);
        const uint32_t sc = wave_incl_scan(c);
        if ((threadIdx.x & 63) == 63) s[threadIdx.x >> 6] = sc;
        __syncthreads();
        if (threadIdx.x == 0) a.seg_cnt[seg] = s[0] + s[1] + s[2] + s[3];
        __syncthreads();
    }
}

__global__ __launch_bounds__(1024) void k_ff_scan(StuffArgs a) {
    __shared__ uint64_t s[1024];
    const int tid = threadIdx.x;
    const uint64_t nbytes = shifted_bytes(a);
    const long long n = (long long)((nbytes + STUFF_SEG - 1) / STUFF_SEG);
    const long long per = (n + 1023) / 1024;
    const long long lo = min(n, tid * per), hi = min(n, lo + per);
    uint64_t sum = 0;
    for (long long i = lo; i < hi; i++) sum += a.seg_cnt[i];
    s[tid] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const uint64_t t = tid >= d ? s[tid - d] : 0;
        __syncthreads();
        s[tid] += t;
        __syncthreads();
    }
    uint64_t run = s[tid] - sum;
    for (long long i = lo; i < hi; i++) {
        a.seg_off[i] = (uint64_t)i * STUFF_SEG + run;
        run += a.seg_cnt[i];
    }
    if (tid == 1023) a.status->scan_bytes = nbytes + s[1023];
    if (tid == 0) {
        // bits of the partial byte a non-final stripe leaves to its successor
        const uint64_t tbits = a.status->nbits + a.shift;
        uint32_t tail = 0;
        if (!a.is_last && (tbits & 7)) {
            const uint64_t in_bytes = (a.status->nbits + 7) >> 3;
            uint32_t t[4];
            load16(a, nbytes >> 4, in_bytes, t);
            tail = (t[(nbytes >> 2) & 3] >> (24 - 8 * (nbytes & 3))) & 0xffu;
        }
        a.status->tail = tail;
    }
}

__global__ __launch_bounds__(256) void k_ff_scatter(StuffArgs a) {
    __shared__ uint32_t s[4];
    const uint64_t in_bytes = (a.status->nbits + 7) >> 3;
    const uint64_t nbytes = shifted_bytes(a);
    const uint64_t nseg = (nbytes + STUFF_SEG - 1) / STUFF_SEG;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint64_t seg = blockIdx.x; seg < nseg; seg += gridDim.x) {
        const uint64_t q = seg * (STUFF_SEG / 16) + threadIdx.x;
        uint32_t t[4];
        load16(a, q, in_bytes, t);
        uint32_t c = 0;
        if (16 * q < nbytes) c = count_ff(t[0]) + count_ff(t[1]) + count_ff(t[2]) + count_ff(t[3]);
        const uint32_t sc = wave_incl_scan(c);
        if (lane == 63) s[wave] = sc;
        __syncthreads();
        uint32_t before = sc - c;
        for (int w = 0; w < wave; w++) before += s[w];
        uint8_t *o = a.out + a.seg_off[seg] + 16 * threadIdx.x + before;
        if (16 * q + 16 <= nbytes && c == 0 && ((reinterpret_cast<uintptr_t>(o) & 3) == 0)) {
            // common case: 16 plain bytes, destination word-aligned
            uint32_t *o4 = reinterpret_cast<uint32_t *>(o);
#pragma unroll
            for (int i = 0; i < 4; i++) o4[i] = __builtin_bswap32(t[i]);
        } else {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (16 * q + i < nbytes) {
                    const uint8_t b = (uint8_t)(t[i >> 2] >> (24 - 8 * (i & 3)));
                    *o++ = b;
                    if (b == 0xff) *o++ = 0;
                }
            }
        }
        __syncthreads();
    }
}

hipError_t launch_entropy(const Geom &g, const int16_t *coef, const int16_t pred[3], const EntropyBufs &b,
                          hipStream_t s) {
    const long long nchunks = (g.nblocks + CHUNK_BLOCKS - 1) / CHUNK_BLOCKS;
    const unsigned grid = (unsigned)((nchunks + ENT_WAVES - 1) / ENT_WAVES);
    hipError_t e = hipMemsetAsync(b.status, 0, sizeof(DevStatus), s);
    if (e != hipSuccess) return e;
    EntArgs a{coef, g.nblocks, g.nb, g.h * g.v, {pred[0], pred[1], pred[2]}, b.blk_bits, b.chunk_len, b.chunk_base, b.bits, b.status};
    hipLaunchKernelGGL((k_entropy<false>), dim3(grid), dim3(ENT_THREADS), 0, s, a);
    hipLaunchKernelGGL(k_scan_chunks, dim3(1), dim3(1024), 0, s, b.chunk_len, nchunks, b.chunk_base, b.status, b.bits);
    hipLaunchKernelGGL((k_entropy<true>), dim3(grid), dim3(ENT_THREADS), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_stuff(const EntropyBufs &b, uint64_t bit_offset, uint8_t head_bits, int is_last, hipStream_t s) {
    StuffArgs a{b.bits, b.status, (uint32_t)(bit_offset & 7), head_bits, is_last, b.seg_cnt, b.seg_off, b.scan};
    const size_t max_seg = b.nseg_cap;
    const unsigned grid = (unsigned)(max_seg < 2048 ? (max_seg ? max_seg : 1) : 2048);
    hipLaunchKernelGGL(k_ff_count, dim3(grid), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_ff_scan, dim3(1), dim3(1024), 0, s, a);
    hipLaunchKernelGGL(k_ff_scatter, dim3(grid), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace jpg
