// Entropy coding on the device: DC-difference / AC run-length symbols
// (src/jpeg/entropy_coding.rs:144-199), Huffman codes + amplitude bits
// (src/jpeg/entropy_coding.rs:201-227), MSB-first bit packing
// (src/utils/bitvec_utils.rs:3-8) and 0xFF byte stuffing
// (src/jpeg/file.rs:92-103).  The reference does this serially; here the
// variable-length output is placed by prefix sums.
//
// k_entropy: one thread codes one block, one wavefront a chunk of 64
// consecutive blocks.  A thread walks only the non-zero coefficients of its
// block (a 64-bit mask built while the block is read from LDS), so the work
// follows the number of symbols, not the number of coefficients.  The wave
//   1. measures its blocks, scans the lengths, announces the chunk's length,
//   2. codes the blocks into LDS (bit 0 = first bit of the chunk),
//   3. learns the chunk's bit offset in the stream from the lengths announced
//      by the chunks before it (three levels of sums, so nobody waits on a chain),
//   4. writes the bits to global memory shifted to that offset, as whole
//      words: the word a chunk shares with its predecessor is written by the
//      later chunk, which gets the predecessor's last 32 bits handed over.
// A wave only ever waits for chunks with a lower number, which belong to
// workgroups dispatched before its own.
//
// k_stuff then inserts 0x00 after every 0xFF, placing its segments the same way.
#include "../../include/jpeg_hip.h"
#include "internal.hpp"
#include "tables.h"

namespace jpg {

__device__ const jpgtab::HuffLut HUFF_DEV{};
constexpr int LUT_WORDS = 2 * jpgtab::HUFF_LUT_STRIDE;
#ifndef JPG_ENT_THREADS
#define JPG_ENT_THREADS 128
#endif
constexpr int ENT_THREADS = JPG_ENT_THREADS;
constexpr int ENT_WAVES = ENT_THREADS / 64;
constexpr uint32_t SPIN_LIMIT = 1u << 22;       // polls before a wave gives up waiting (never hang the GPU)
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
    long long nblocks, nchunks;
    int nb, hv;
    int16_t pred[3];
    uint32_t epoch;
    uint64_t *agg, *grp, *hyp, *tail;
    uint32_t *bits;
    DevStatus *status;
};

// Errors are recorded as -(epoch << 8 | code) with atomicMin, so a newer launch always
// overrides an older one and the word never needs clearing between launches.
__device__ __forceinline__ void raise_error(DevStatus *st, uint32_t epoch, int status_code) {
    atomicMin(&st->error, -(int)((epoch << 8) | (uint32_t)(-status_code)));
}

// hand-off word: {epoch:14 | valid:1 | value:49}
__device__ __forceinline__ uint64_t hw_pack(uint32_t epoch, uint64_t value) {
    return ((uint64_t)((epoch << 1) | 1u) << 49) | value;
}
__device__ __forceinline__ bool hw_valid(uint64_t w, uint32_t epoch) { return (w >> 49) == (uint64_t)((epoch << 1) | 1u); }
__device__ __forceinline__ uint64_t hw_value(uint64_t w) { return w & ((1ull << 49) - 1); }
__device__ __forceinline__ void hw_store(uint64_t *p, uint64_t w) {
    __hip_atomic_store(p, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint64_t hw_load(const uint64_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

struct NullWriter {
    __device__ __forceinline__ void put(uint32_t, uint32_t) {}
    __device__ __forceinline__ void finish() {}
};

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
                raise_error(st, epoch, JPG_ERR_AC_RANGE);
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

// Three levels of running sums over items 0..n-1 (chunks of blocks here, segments
// of bytes in the stuffing kernel), each word {epoch | valid | value}:
//   agg[i]  value of item i                (published by i as soon as it is known)
//   grp[g]  sum of items 64g..64g+63       (published by the last item of the group)
//   hyp[h]  sum of groups 64h..64h+63      (published by the last item of the hyper-group)
// prefix(i) = sum hyp[< h] + sum grp[64h .. g-1] + sum agg[64g .. i-1].  Every sum is
// published as soon as it is known and before its publisher waits for anything else, so
// the publications never form a chain; an item only waits for items with a lower number.
struct Sums {
    uint64_t *agg, *grp, *hyp;
    uint32_t epoch;
    long long n;
};

// words fetched ahead of time (they may not be valid yet)
struct Probe {
    uint64_t wa, wg, wh;
};

__device__ __forceinline__ Probe sums_probe(const Sums &s, long long i, int lane) {
    const long long g = i >> 6, h = g >> 6;
    Probe p{0, 0, 0};
    if ((g << 6) + lane < i) p.wa = hw_load(&s.agg[(g << 6) + lane]);
    if ((h << 6) + lane < g) p.wg = hw_load(&s.grp[(h << 6) + lane]);
    if (lane < h) p.wh = hw_load(&s.hyp[lane]);
    return p;
}

// poll a hand-off word until it carries this launch's epoch; `w` is an earlier look at it
__device__ __forceinline__ uint64_t hw_wait(const uint64_t *p, bool need, uint64_t w, uint32_t epoch, bool &timeout) {
    bool ok = !need || hw_valid(w, epoch);
    for (uint32_t spin = 0;; spin++) {
        if (__ballot(!ok) == 0) break;
        if (spin >= SPIN_LIMIT) { timeout = true; break; }
        if (spin) __builtin_amdgcn_s_sleep(1);
        if (!ok) {
            w = hw_load(p);
            ok = hw_valid(w, epoch);
        }
    }
    return need && ok ? hw_value(w) : 0;
}

__device__ __forceinline__ uint64_t wave_sum64(uint64_t v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const uint32_t lo = __shfl_xor((uint32_t)v, d), hi = __shfl_xor((uint32_t)(v >> 32), d);
        v += ((uint64_t)hi << 32) | lo;
    }
    return v;
}

// exclusive prefix of item i whose own value is `own`; called by a whole wave
__device__ __forceinline__ uint64_t sums_prefix(const Sums &s, long long i, uint64_t own, const Probe &pr, int lane,
                                                bool &timeout) {
    const long long g = i >> 6, h = g >> 6;
    const bool closes_group = (i & 63) == 63 || i == s.n - 1;
    const bool closes_hyper = closes_group && ((g & 63) == 63 || i == s.n - 1);
    const long long ia = (g << 6) + lane;
    const uint64_t sum_a = wave_sum64(hw_wait(&s.agg[ia], ia < i, pr.wa, s.epoch, timeout));
    const uint64_t gv = sum_a + own;
    if (closes_group && lane == 0) hw_store(&s.grp[g], hw_pack(s.epoch, gv));
    const long long ig = (h << 6) + lane;
    const uint64_t sum_g = wave_sum64(hw_wait(&s.grp[ig], ig < g, pr.wg, s.epoch, timeout));
    if (closes_hyper && lane == 0) hw_store(&s.hyp[h], hw_pack(s.epoch, sum_g + gv));
    uint64_t sh = 0;
    for (long long j0 = 0; j0 < h; j0 += 64)
        sh += hw_wait(&s.hyp[j0 + lane], j0 + lane < h, j0 ? 0 : pr.wh, s.epoch, timeout);
    return wave_sum64(sh) + sum_g + sum_a;
}

__global__ __launch_bounds__(ENT_THREADS) void k_entropy(EntArgs a) {
    __shared__ uint32_t sLut[LUT_WORDS];
    __shared__ __attribute__((aligned(16))) uint8_t sCoef[ENT_WAVES][CHUNK_BLOCKS * SLOT_BYTES];
    __shared__ uint32_t sOut[ENT_WAVES][OUT_WORDS + 2];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < LUT_WORDS; i += ENT_THREADS) sLut[i] = (&HUFF_DEV.e[0][0])[i];

    __syncthreads();
    // A wave waits only for chunks with a lower number (see chunk_offset and the tail
    // hand-off), i.e. for workgroups with a lower blockIdx.  Workgroups are dispatched in
    // blockIdx order, so those are already running or finished; every wait is bounded all
    // the same and ends in JPG_ERR_HIP rather than a hang.
    const long long chunk = (long long)blockIdx.x * ENT_WAVES + wave;
    if (chunk >= a.nchunks) return;                              // wave-uniform; no workgroup barrier follows
    {
    const long long first = chunk * CHUNK_BLOCKS;
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

    // 1. lengths
    NullWriter nw;
    const uint32_t len = valid ? code_block(slot, mlo, mhi, dc_diff, lut, a.status, a.epoch, nw) : 0u;
    const uint32_t sc = wave_incl_scan(len);                      // bits up to and including this lane's block
    const uint32_t total = __shfl(sc, 63);
    const Sums sums{a.agg, a.grp, a.hyp, a.epoch, a.nchunks};
    if (lane == 0) hw_store(&a.agg[chunk], hw_pack(a.epoch, total));

    // 2..4. code and write out; a chunk too long for the LDS window goes in several passes
    uint32_t *out = sOut[wave];
    constexpr uint32_t WINDOW = OUT_WORDS * 32;                   // bits per pass
    uint64_t O = 0;                                               // bit offset of the chunk, known after the first pass
    uint32_t prev_tail = 0;                                       // the 32 stream bits before the current pass
    uint32_t done = 0;                                            // bits of the chunk already written out
    bool timeout = false;
    for (int pass = 0; done < total || pass == 0; pass++) {
        // lanes whose block ends within the window starting at `done` (at least one: a block is < WINDOW bits)
        const bool mine = valid && sc > done && sc - done <= WINDOW;
        const uint32_t pass_bits = __shfl(mine ? sc - done : 0u, 63 - __clzll(__ballot(mine) | 1ull));
        const uint32_t nw_local = (pass_bits + 31) >> 5;
        for (uint32_t i = lane; i < nw_local + 1; i += 64) out[i] = 0;
        __builtin_amdgcn_wave_barrier();
        if (mine) {
            BitWriter w;
            w.init(out, sc - len - done);
            code_block(slot, mlo, mhi, dc_diff, lut, a.status, a.epoch, w);
        }
        __builtin_amdgcn_wave_barrier();
        const bool last_pass = done + pass_bits >= total;
        // last 32 bits of the stream after this pass
        uint32_t new_tail;
        {
            const uint32_t e = pass_bits;                         // > 0
            const uint32_t wi = (e - 1) >> 5, r = e & 31;         // r = bits used in the last word (0 = full)
            const uint32_t lastw = out[wi], beforew = wi ? out[wi - 1] : prev_tail;
            new_tail = r ? __funnelshift_r(lastw, beforew, 32 - r) : lastw;
            if (e < 32) new_tail = (prev_tail << e) | (out[0] >> (32 - e));
        }
        if (last_pass && lane == 0) hw_store(&a.tail[chunk], hw_pack(a.epoch, new_tail));
        if (pass == 0) {
            // 3. where does the chunk start?
            // (looking at the predecessors' words any earlier only adds traffic: measured slower)
            O = sums_prefix(sums, chunk, total, Probe{0, 0, 0}, lane, timeout);
            if (chunk > 0) {
                const uint64_t tail_word = hw_wait(&a.tail[chunk - 1], lane == 0, 0, a.epoch, timeout);
                prev_tail = __shfl((uint32_t)tail_word, 0);
            }
            if (chunk == a.nchunks - 1 && lane == 0) a.status->nbits = O + total;
            if (__ballot(timeout) && lane == 0) raise_error(a.status, a.epoch, JPG_ERR_HIP);
        }
        // 4. write the pass at bit O + done: word i = stream bits, preceded by prev_tail
        const uint64_t P = O + done;
        const uint32_t shift = (uint32_t)(P & 31);
        const uint64_t W0 = P >> 5;
        const uint32_t end_bits = shift + pass_bits;
        // whole words only, except at the very end of the stream
        const uint32_t nout = (last_pass && chunk == a.nchunks - 1) ? (end_bits + 31) >> 5 : end_bits >> 5;
        for (uint32_t i = lane; i < nout; i += 64) {
            const uint32_t hi = i ? out[i - 1] : prev_tail;
            const uint32_t lo = i < nw_local ? out[i] : 0u;
            a.bits[W0 + i] = __builtin_bswap32(shift ? __funnelshift_r(lo, hi, shift) : lo);
        }
        __builtin_amdgcn_wave_barrier();
        prev_tail = new_tail;
        done += pass_bits;
    }
    }
}

// ---------------------------------------------------------------- stuffing

struct StuffArgs {
    const uint32_t *in;      // unstuffed stream (whole words are defined up to the last one)
    DevStatus *status;       // nbits in, scan_bytes and tail out
    uint32_t shift;          // place the stream `shift` (0..7) bits later
    uint32_t head;           // bits OR-ed into the first byte (top `shift` bits)
    int is_last;             // keep the trailing partial byte
    uint64_t *agg, *grp, *hyp;  // three-level sums of 0xFF counts per segment
    uint32_t epoch;          // generation of those words
    uint32_t err_epoch;      // epoch errors are tagged with (that of the entropy launch)
    uint8_t *out;
    DevStatus *host_status;  // pinned host copy, filled at the end (may be null)
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

// Byte stuffing in one pass: a workgroup takes segments of STUFF_SEG input bytes
// (blockIdx.x, + gridDim.x, ...), counts their 0xFF bytes, learns how many precede the
// segment from the three-level sums, and writes the segment with 0x00 after every 0xFF.
__global__ __launch_bounds__(256) void k_stuff(StuffArgs a) {
    __shared__ uint32_t s[4];
    __shared__ uint64_t sBefore;
    const uint64_t in_bytes = (a.status->nbits + 7) >> 3;
    const uint64_t nbytes = shifted_bytes(a);
    const uint64_t nseg = (nbytes + STUFF_SEG - 1) / STUFF_SEG;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const Sums sums{a.agg, a.grp, a.hyp, a.epoch, (long long)nseg};
    if (nseg == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
        a.status->scan_bytes = 0;
        a.status->tail = 0;
        if (a.host_status) *a.host_status = *a.status;
    }
    for (uint64_t seg = blockIdx.x; seg < nseg; seg += gridDim.x) {
        const uint64_t q = seg * (STUFF_SEG / 16) + threadIdx.x;
        uint32_t t[4];
        load16(a, q, in_bytes, t);
        uint32_t c = 0;
        if (16 * q < nbytes) c = count_ff(t[0]) + count_ff(t[1]) + count_ff(t[2]) + count_ff(t[3]);
        const uint32_t sc = wave_incl_scan(c);
        if (lane == 63) s[wave] = sc;
        __syncthreads();
        const uint32_t total = s[0] + s[1] + s[2] + s[3];
        if (wave == 0) {
            bool timeout = false;
            if (lane == 0) hw_store(&a.agg[seg], hw_pack(a.epoch, total));
            const Probe probe = sums_probe(sums, (long long)seg, lane);
            const uint64_t before = sums_prefix(sums, (long long)seg, total, probe, lane, timeout);
            if (lane == 0) {
                sBefore = before;
                if (timeout) raise_error(a.status, a.err_epoch, JPG_ERR_HIP);
                if (seg == nseg - 1) {
                    a.status->scan_bytes = nbytes + before + total;
                    // bits of the partial byte a non-final stripe leaves to its successor
                    const uint64_t tbits = a.status->nbits + a.shift;
                    uint32_t tail = 0;
                    if (!a.is_last && (tbits & 7)) {
                        uint32_t u[4];
                        load16(a, nbytes >> 4, in_bytes, u);
                        tail = (u[(nbytes >> 2) & 3] >> (24 - 8 * (nbytes & 3))) & 0xffu;
                    }
                    a.status->tail = tail;
                    if (a.host_status) *a.host_status = *a.status;
                }
            }
        }
        __syncthreads();
        uint32_t before = sc - c;
        for (int w = 0; w < wave; w++) before += s[w];
        uint8_t *o = a.out + seg * STUFF_SEG + sBefore + 16 * threadIdx.x + before;
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
    EntArgs a{coef, g.nblocks, nchunks, g.nb, g.h * g.v, {pred[0], pred[1], pred[2]}, b.epoch, b.agg, b.grp, b.hyp, b.tail,
              b.bits, b.status};
    hipLaunchKernelGGL(k_entropy, dim3(grid), dim3(ENT_THREADS), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_stuff(const EntropyBufs &b, uint64_t bit_offset, uint8_t head_bits, int is_last, hipStream_t s) {
    StuffArgs a{b.bits, b.status, (uint32_t)(bit_offset & 7), head_bits, is_last, b.sagg, b.sgrp, b.shyp, b.stuff_epoch,
                b.epoch, b.scan, b.host_status};
    const size_t cap = b.stuff_grid ? b.stuff_grid : 1024;
    const unsigned grid = (unsigned)(b.nseg_cap < cap ? (b.nseg_cap ? b.nseg_cap : 1) : cap);
    hipLaunchKernelGGL(k_stuff, dim3(grid), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace jpg
