// Entropy coding on the device: DC-difference / AC run-length symbols
// (src/jpeg/entropy_coding.rs:144-199), Huffman codes + amplitude bits
// (src/jpeg/entropy_coding.rs:201-227), MSB-first bit packing
// (src/utils/bitvec_utils.rs:3-8) and 0xFF byte stuffing
// (src/jpeg/file.rs:92-103).  The reference does this serially; here the
// variable-length output is placed by prefix sums:
//
//   k_entropy<false>  bits of every chunk of 8 blocks, scanned inside a group
//   k_scan_groups     exclusive scan of the group totals (one workgroup)
//   k_zero_edges      clears the first and last word of every chunk, which
//                     neighbouring chunks share and therefore OR into
//   k_entropy<true>   codes the blocks again and writes the bits in place
//   k_ff_count / k_ff_scan / k_ff_scatter   the same three steps for stuffing
//
// A wavefront codes one chunk at a time: lane (g, j) holds zigzag
// coefficients 8j..8j+7 of block g.  The non-zero coefficients of the chunk
// are compacted through LDS so that the Huffman lookups and the bit placement
// run on dense lanes, in stream order, and one wave-level prefix sum of the
// code lengths gives every code its bit offset.
#include "../../include/jpeg_hip.h"
#include "internal.hpp"
#include "tables.h"

namespace jpg {

__device__ const jpgtab::HuffLut HUFF_DEV{};
constexpr int LUT_WORDS = 2 * jpgtab::HUFF_LUT_STRIDE;
constexpr int ENT_PER_WAVE = CHUNK_BLOCKS * 64;
// words one chunk can occupy: 8 worst-case blocks + up to 31 bits of offset
constexpr int BITS_WORDS = (CHUNK_BLOCKS * MAX_BLOCK_BITS + 31 + 31) / 32 + 1;

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(v, d);
        if (lane >= d) v += t;
    }
    return v;
}

// OR the low `len` bits of `pay` into the MSB-first bit string `buf` at bit P
__device__ __forceinline__ void or_bits(uint32_t *buf, uint32_t P, uint64_t pay, uint32_t len) {
    const uint64_t hi = pay << (64 - len);
    const uint32_t w = P >> 5, s = P & 31;
    const uint64_t a = hi >> s;
    const uint32_t w0 = (uint32_t)(a >> 32), w1 = (uint32_t)a;
    const uint32_t w2 = s ? (uint32_t)((hi << (64 - s)) >> 32) : 0u;
    if (w0) atomicOr(&buf[w], w0);
    if (w1) atomicOr(&buf[w + 1], w1);
    if (w2) atomicOr(&buf[w + 2], w2);
}

struct EntArgs {
    const int16_t *coef;
    long long nblocks;
    int nb, hv;
    const int16_t *pred;
    uint32_t *chunk_len, *chunk_off, *group_len;
    const uint64_t *group_base;
    uint32_t *bits;
    int *error;
};

template <bool PACK>
__global__ __launch_bounds__(256) void k_entropy(EntArgs a) {
    __shared__ uint32_t sLut[LUT_WORDS];
    __shared__ uint32_t sEnt[4][ENT_PER_WAVE];
    __shared__ uint32_t sBits[PACK ? 4 : 1][PACK ? BITS_WORDS : 1];
    __shared__ uint32_t sChunk[GROUP_CHUNKS];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int g = lane >> 3, j = lane & 7;
    for (int i = tid; i < LUT_WORDS; i += 256) sLut[i] = (&HUFF_DEV.e[0][0])[i];
    if (PACK)
        for (int i = lane; i < BITS_WORDS; i += 64) sBits[wave][i] = 0;
    __syncthreads();

    const long long group = blockIdx.x;
    for (int ci = 0; ci < GROUP_CHUNKS / 4; ci++) {
        const int cig = wave * (GROUP_CHUNKS / 4) + ci;        // chunk inside the group
        const long long chunk = group * GROUP_CHUNKS + cig;
        if (chunk * CHUNK_BLOCKS >= a.nblocks) {               // wave-uniform
            if (!PACK && lane == 0) sChunk[cig] = 0;
            continue;
        }
        const long long blk = chunk * CHUNK_BLOCKS + g;
        const bool valid = blk < a.nblocks;
        uint4 raw = make_uint4(0, 0, 0, 0);
        if (valid) raw = reinterpret_cast<const uint4 *>(a.coef)[blk * 8 + j];
        int c[8];
        c[0] = (int16_t)(raw.x & 0xffff); c[1] = (int16_t)(raw.x >> 16);
        c[2] = (int16_t)(raw.y & 0xffff); c[3] = (int16_t)(raw.y >> 16);
        c[4] = (int16_t)(raw.z & 0xffff); c[5] = (int16_t)(raw.z >> 16);
        c[6] = (int16_t)(raw.w & 0xffff); c[7] = (int16_t)(raw.w >> 16);
        const int jb = (int)(blk % a.nb);
        const uint32_t chroma = jb >= a.hv ? 1u : 0u;
        uint32_t m8 = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) m8 |= (c[i] != 0 ? 1u : 0u) << i;
        if (j == 0 && valid) {
            // DC difference against the previous block of the same component
            // (three predictor chains, src/jpeg/entropy_coding.rs:38-40,149-150)
            const long long prev = chroma ? blk - a.nb : (jb > 0 ? blk - 1 : blk - a.nb + a.hv - 1);
            const int pdc = prev >= 0 ? (int)a.coef[prev * 64] : (int)a.pred[chroma ? jb - a.hv + 1 : 0];
            c[0] = (int16_t)(c[0] - pdc);
            m8 |= 1u;                                          // the DC symbol is always coded
        }
        const uint32_t cnt = __popc(m8);
        const uint32_t incl = wave_incl_scan(cnt, lane);
        const uint32_t base = incl - cnt;
        const uint32_t total = __shfl(incl, 63);
        // zigzag index of the last non-zero coefficient before this lane's, within the block
        int hi = m8 ? 8 * j + (31 - __clz(m8)) : -1;
#pragma unroll
        for (int d = 1; d < 8; d <<= 1) {
            const int t = __shfl_up(hi, d, 8);
            if (j >= d) hi = max(hi, t);
        }
        int prevk = __shfl_up(hi, 1, 8);
        if (j == 0) prevk = -1;
        // every lane must take part in the shuffle: an inactive source lane reads as 0
        const uint32_t next_base = (uint32_t)__shfl(base, ((g + 1) * 8) & 63);
        const uint32_t blk_end = g < 7 ? next_base : total;
        uint32_t pos = base;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (m8 & (1u << i)) {
                const int k = 8 * j + i;
                const uint32_t run = (uint32_t)(k - prevk - 1);
                prevk = k;
                const uint32_t eob = (pos == blk_end - 1 && k < 63) ? 1u : 0u;  // trailing zeros -> EOB
                sEnt[wave][pos] = ((uint32_t)c[i] & 0xffffu) | (run << 16) | ((k == 0 ? 1u : 0u) << 22) | (eob << 23) |
                                  (chroma << 24);
                pos++;
            }
        }
        __builtin_amdgcn_wave_barrier();

        uint64_t O = 0;
        uint32_t shift = 0;
        if (PACK) {
            O = a.group_base[group] + a.chunk_off[chunk];
            shift = (uint32_t)(O & 31);
        }
        uint32_t running = 0;
        for (uint32_t e0 = 0; e0 < total; e0 += 64) {
            const uint32_t idx = e0 + lane;
            uint64_t pay = 0;
            uint32_t len = 0;
            if (idx < total) {
                const uint32_t e = sEnt[wave][idx];
                const int v = (int16_t)(e & 0xffff);
                const uint32_t run = (e >> 16) & 63, is_dc = (e >> 22) & 1, eob = (e >> 23) & 1, ch = (e >> 24) & 1;
                const uint32_t mag = (uint32_t)(v < 0 ? -v : v);
                uint32_t size = 32 - __clz(mag);               // bit_length, src/jpeg/entropy_coding.rs:127-134
                if (size > (is_dc ? 11u : 10u)) {
                    atomicMin(a.error, is_dc ? (int)JPG_ERR_DC_RANGE : (int)JPG_ERR_AC_RANGE);
                    size = 0;
                }
                // negative values: v + (1 << size) - 1 (src/jpeg/entropy_coding.rs:140-142)
                const uint32_t amp = (uint32_t)(v < 0 ? v - 1 : v) & ((1u << size) - 1u);
                const uint32_t *lut = sLut + ch * jpgtab::HUFF_LUT_STRIDE;
                const uint32_t t = lut[is_dc ? 256 + size : (((run & 15) << 4) | size)];
                pay = ((uint64_t)(t & 0xffff) << size) | amp;
                len = (t >> 16) + size;
                const uint32_t z = lut[0xf0];                  // ZRL, one per 16 zeros
                for (uint32_t n = run >> 4; n > 0; n--) {
                    pay |= (uint64_t)(z & 0xffff) << len;
                    len += z >> 16;
                }
                if (eob) {
                    const uint32_t eo = lut[0x00];
                    pay = (pay << (eo >> 16)) | (eo & 0xffff);
                    len += eo >> 16;
                }
            }
            const uint32_t sc = wave_incl_scan(len, lane);
            if (PACK && len) or_bits(sBits[wave], shift + running + sc - len, pay, len);
            running += __shfl(sc, 63);
        }
        __builtin_amdgcn_wave_barrier();

        if (!PACK) {
            if (lane == 0) sChunk[cig] = running;
        } else {
            const uint64_t W0 = O >> 5;
            const uint32_t nwords = (shift + running + 31) >> 5;
            for (uint32_t w = lane; w < nwords; w += 64) {
                const uint32_t v = __builtin_bswap32(sBits[wave][w]);
                sBits[wave][w] = 0;
                if (w == 0 || w == nwords - 1) atomicOr(&a.bits[W0 + w], v);
                else a.bits[W0 + w] = v;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (!PACK) {
        __syncthreads();
        if (wave == 0) {
            const uint32_t v = lane < GROUP_CHUNKS ? sChunk[lane] : 0;
            const uint32_t sc = wave_incl_scan(v, lane);
            if (lane < GROUP_CHUNKS) {
                const long long chunk = group * GROUP_CHUNKS + lane;
                if (chunk * CHUNK_BLOCKS < a.nblocks) {
                    a.chunk_len[chunk] = v;
                    a.chunk_off[chunk] = sc - v;
                }
            }
            if (lane == GROUP_CHUNKS - 1) a.group_len[group] = sc;
        }
    }
}

// exclusive scan of n 32-bit lengths into 64-bit offsets by one workgroup;
// out_total[0] = sum
__global__ __launch_bounds__(1024) void k_scan_groups(const uint32_t *__restrict__ len, long long n,
                                                      uint64_t *__restrict__ base, uint64_t *__restrict__ out_total) {
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
        base[i] = run;
        run += len[i];
    }
    if (tid == 1023) out_total[0] = s[1023];
}

__global__ void k_zero_edges(const uint32_t *__restrict__ chunk_len, const uint32_t *__restrict__ chunk_off,
                             const uint64_t *__restrict__ group_base, long long nchunks, uint32_t *__restrict__ bits) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchunks) return;
    const uint64_t O = group_base[c / GROUP_CHUNKS] + chunk_off[c];
    const uint32_t L = chunk_len[c];
    if (L == 0) return;
    bits[O >> 5] = 0;
    bits[(O + L - 1) >> 5] = 0;
}

// ---------------------------------------------------------------- stuffing

struct StuffArgs {
    const uint32_t *in;      // unstuffed stream (whole words are defined up to the last one)
    const uint64_t *totals;  // [0] = bits in `in`
    uint64_t *totals_out;    // [1] = stuffed bytes
    uint32_t shift;          // place the stream `shift` (0..7) bits later
    uint32_t head;           // bits OR-ed into the first byte (top `shift` bits)
    int is_last;             // keep the trailing partial byte
    uint32_t *seg_cnt;
    uint64_t *seg_off;
    uint8_t *out;
    uint8_t *tail;
};

__device__ __forceinline__ uint64_t shifted_bytes(const StuffArgs &a) {
    const uint64_t tbits = a.totals[0] + a.shift;
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
    uint32_t prev = 0;
    if (a.shift) {
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
    const uint64_t in_bytes = (a.totals[0] + 7) >> 3;
    const uint64_t nbytes = shifted_bytes(a);
    const uint64_t nseg = (nbytes + STUFF_SEG - 1) / STUFF_SEG;
    for (uint64_t seg = blockIdx.x; seg < nseg; seg += gridDim.x) {
        const uint64_t q = seg * (STUFF_SEG / 16) + threadIdx.x;
        uint32_t t[4];
        load16(a, q, in_bytes, t);
        uint32_t c = 0;
        if (16 * q < nbytes) c = count_ff(t[0]) + count_ff(t[1]) + count_ff(t[2]) + count_ff(t[3]);
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d);
        if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = c;
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
    if (tid == 1023) a.totals_out[1] = nbytes + s[1023];
    if (tid == 0) {
        // bits of the partial byte a non-final stripe leaves to its successor
        const uint64_t tbits = a.totals[0] + a.shift;
        uint8_t tail = 0;
        if (!a.is_last && (tbits & 7)) {
            const uint64_t in_bytes = (a.totals[0] + 7) >> 3;
            uint32_t t[4];
            load16(a, nbytes >> 4, in_bytes, t);
            tail = (uint8_t)(t[(nbytes >> 2) & 3] >> (24 - 8 * (nbytes & 3)));
        }
        a.tail[0] = tail;
    }
}

__global__ __launch_bounds__(256) void k_ff_scatter(StuffArgs a) {
    __shared__ uint32_t s[4];
    const uint64_t in_bytes = (a.totals[0] + 7) >> 3;
    const uint64_t nbytes = shifted_bytes(a);
    const uint64_t nseg = (nbytes + STUFF_SEG - 1) / STUFF_SEG;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint64_t seg = blockIdx.x; seg < nseg; seg += gridDim.x) {
        const uint64_t q = seg * (STUFF_SEG / 16) + threadIdx.x;
        uint32_t t[4];
        load16(a, q, in_bytes, t);
        uint32_t c = 0;
        if (16 * q < nbytes) c = count_ff(t[0]) + count_ff(t[1]) + count_ff(t[2]) + count_ff(t[3]);
        uint32_t sc = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t u = __shfl_up(sc, d);
            if (lane >= d) sc += u;
        }
        if (lane == 63) s[wave] = sc;
        __syncthreads();
        uint32_t before = sc - c;
        for (int w = 0; w < wave; w++) before += s[w];
        uint8_t *o = a.out + a.seg_off[seg] + 16 * threadIdx.x + before;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (16 * q + i < nbytes) {
                const uint8_t b = (uint8_t)(t[i >> 2] >> (24 - 8 * (i & 3)));
                *o++ = b;
                if (b == 0xff) *o++ = 0;
            }
        }
        __syncthreads();
    }
}

hipError_t launch_entropy(const Geom &g, const int16_t *coef, const int16_t *pred, const EntropyBufs &b,
                          hipStream_t s) {
    const long long nchunks = (g.nblocks + CHUNK_BLOCKS - 1) / CHUNK_BLOCKS;
    const long long ngroups = (nchunks + GROUP_CHUNKS - 1) / GROUP_CHUNKS;
    hipError_t e = hipMemsetAsync(b.error, 0, sizeof(int), s);
    if (e != hipSuccess) return e;
    EntArgs a{coef, g.nblocks, g.nb, g.h * g.v, pred, b.chunk_len, b.chunk_off, b.group_len, b.group_base, b.bits, b.error};
    hipLaunchKernelGGL((k_entropy<false>), dim3((unsigned)ngroups), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_scan_groups, dim3(1), dim3(1024), 0, s, b.group_len, ngroups, b.group_base, b.totals);
    hipLaunchKernelGGL(k_zero_edges, dim3((unsigned)((nchunks + 255) / 256)), dim3(256), 0, s, b.chunk_len, b.chunk_off,
                       b.group_base, nchunks, b.bits);
    hipLaunchKernelGGL((k_entropy<true>), dim3((unsigned)ngroups), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_stuff(const EntropyBufs &b, uint64_t bit_offset, uint8_t head_bits, int is_last, hipStream_t s) {
    StuffArgs a{b.bits, b.totals, b.totals, (uint32_t)(bit_offset & 7), head_bits, is_last, b.seg_cnt, b.seg_off, b.scan, b.tail};
    const size_t max_seg = b.nseg_cap;
    const unsigned grid = (unsigned)(max_seg < 2048 ? (max_seg ? max_seg : 1) : 2048);
    hipLaunchKernelGGL(k_ff_count, dim3(grid), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_ff_scan, dim3(1), dim3(1024), 0, s, a);
    hipLaunchKernelGGL(k_ff_scatter, dim3(grid), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace jpg
