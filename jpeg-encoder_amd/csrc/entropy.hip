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
// measures its blocks, scans the lengths, codes the blocks into LDS (bit 0 =
// first bit of the chunk) and stores the words to the chunk's own fixed-size
// slot of the "sparse" buffer, together with the chunk's length in bits.  No
// wave depends on another one.
//
// k_stream<true> (byte stuffing) and k_stream<false> (dense copy) read the
// concatenation of the chunks: from the running sums of the chunk lengths (per
// group of 64 chunks: k_group_sums) a workgroup finds the chunks under its 4 KiB
// segment of the stream and gathers the bits, shifted to their place, straight
// from the chunk slots.  Stuffing then inserts 0x00 after every 0xFF; the output
// offset of a segment comes from three levels of running sums of the 0xFF
// counts, published by the segments before it.
#include <atomic>

#include "coding.hpp"

namespace jpg {

#ifndef JPG_ENT_THREADS
#define JPG_ENT_THREADS 256
#endif
constexpr int ENT_THREADS = JPG_ENT_THREADS;
constexpr int ENT_WAVES = ENT_THREADS / 64;
constexpr uint32_t SPIN_LIMIT = 1u << 22;       // polls before a wave gives up waiting (never hang the GPU)
constexpr int SLOT_BYTES = 144;                 // 128-byte block + 16: conflict-free 16-byte LDS access
#ifndef JPG_ENT_OUT_WORDS
#define JPG_ENT_OUT_WORDS 64
#endif
constexpr int OUT_WORDS = JPG_ENT_OUT_WORDS;    // LDS words of a chunk's bits coded per pass

struct EntArgs {
    const int16_t *coef;
    long long nblocks, nchunks;   // per image
    int nimg;                // images in this launch (their blocks follow each other in coef)
    long long cstride;       // chunk slots per image in clen / sparse (>= nchunks)
    int nb, hv;
    int16_t pred[3];
    uint32_t epoch;          // tags errors (see raise_error)
    uint32_t *clen;          // [nimg][cstride] out: bits of each chunk
    uint32_t *sparse;        // [nimg][cstride][CHUNK_STRIDE_WORDS] out: the chunk's bits, MSB first in each word
    DevStatus *status;       // [nimg]
};

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

// Three levels of running sums over items 0..n-1 (the segments of bytes of the stuffing kernel), each word
// {epoch | valid | value}:
//   agg[i]  value of item i                (published by i as soon as it is known)
//   grp[g]  sum of items 64g..64g+63       (published by the last item of the group)
//   hyp[h]  sum of groups 64h..64h+63      (published by the last item of the hyper-group)
// prefix(i) = sum hyp[< h] + sum grp[64h .. g-1] + sum agg[64g .. i-1].  An item publishes its value, and the last
// item of a group the sums it closes, in the first half of its work (sums_close), where it only waits for the
// values of items before it; the prefix is taken in the second half (sums_prefix), a turn later, and publishes
// nothing: no item ever waits for the second half of another.
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
        // (polls that go on back off to one a microsecond: a thousand workgroups polling flat out slow down the
        // very stores they wait for)
        for (uint32_t k = 0; k < (spin < 8u ? spin : 8u); k++) __builtin_amdgcn_s_sleep(4);
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

// sum over the wave of values whose sum fits 32 bits, in every lane
__device__ __forceinline__ uint32_t wave_sum32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan(v), 63);
}

// Item i, whose value `own` has been published, is the last of its group (and perhaps of its hyper-group): the sums
// it closes.  Called by a whole wave.  VALUE_BITS: an item's value is below 2^VALUE_BITS; 64 items, 64 groups or 64
// hyper-groups must add up to less than 2^32.
template <int VALUE_BITS>
__device__ __forceinline__ void sums_close(const Sums &s, long long i, uint32_t own, int lane, bool &timeout) {
    static_assert(VALUE_BITS + 6 + 6 + 6 <= 32, "sums of 64 hyper-groups in 32 bits");
    const long long g = i >> 6, h = g >> 6;
    const bool closes_hyper = (g & 63) == 63;
    const long long ia = (g << 6) + lane, ig = (h << 6) + lane;
    const uint64_t wa = ia < i ? hw_load(&s.agg[ia]) : 0;
    const uint64_t wg = closes_hyper && ig < g ? hw_load(&s.grp[ig]) : 0;
    const uint32_t gv = wave_sum32((uint32_t)hw_wait(&s.agg[ia], ia < i, wa, s.epoch, timeout)) + own;
    if (lane == 0) hw_store(&s.grp[g], hw_pack(s.epoch, gv));
    if (!closes_hyper) return;
    const uint32_t sum_g = wave_sum32((uint32_t)hw_wait(&s.grp[ig], ig < g, wg, s.epoch, timeout));
    if (lane == 0) hw_store(&s.hyp[h], hw_pack(s.epoch, (uint64_t)sum_g + gv));
}

// exclusive prefix of item i; called by a whole wave
__device__ __forceinline__ uint64_t sums_prefix(const Sums &s, long long i, const Probe &pr, int lane, bool &timeout) {
    const long long g = i >> 6, h = g >> 6;
    const long long ia = (g << 6) + lane, ig = (h << 6) + lane;
    const uint32_t va = (uint32_t)hw_wait(&s.agg[ia], ia < i, pr.wa, s.epoch, timeout);
    const uint32_t vg = (uint32_t)hw_wait(&s.grp[ig], ig < g, pr.wg, s.epoch, timeout);
    uint64_t sh = 0;   // (up to 2^30 per hyper-group)
    for (long long j0 = 0; j0 < h; j0 += 64)
        sh += wave_sum32((uint32_t)hw_wait(&s.hyp[j0 + lane], j0 + lane < h, j0 ? 0 : pr.wh, s.epoch, timeout));
    return sh + wave_sum32(va + vg);
}

__device__ __forceinline__ uint32_t mod_small(uint32_t x, uint32_t nb, uint32_t magic) {   // x < 1024
    return x - nb * ((x * magic) >> 16);
}

// Loads of one chunk, issued one chunk ahead: the 8 KiB of coefficients (8 coalesced 16-byte
// pieces per lane) into R0..R7 and, for the few blocks whose DC predecessor lies before the chunk,
// that DC into PDC.  Branch-free (addresses are clamped instead) so that the loads can stay in
// flight across the coding of the previous chunk.
#define JPG_CHUNK_LOAD(COEF, LC, FM, R, PDC)                                                        \
    do {                                                                                            \
        const long long first_ = (LC) * CHUNK_BLOCKS;                                               \
        const uint4 *src_ = reinterpret_cast<const uint4 *>(COEF);                                  \
        R##0 = src_[min(first_ * 8 + 0 * 64 + lane, last_piece)];                                   \
        R##1 = src_[min(first_ * 8 + 1 * 64 + lane, last_piece)];                                   \
        R##2 = src_[min(first_ * 8 + 2 * 64 + lane, last_piece)];                                   \
        R##3 = src_[min(first_ * 8 + 3 * 64 + lane, last_piece)];                                   \
        R##4 = src_[min(first_ * 8 + 4 * 64 + lane, last_piece)];                                   \
        R##5 = src_[min(first_ * 8 + 5 * 64 + lane, last_piece)];                                   \
        R##6 = src_[min(first_ * 8 + 6 * 64 + lane, last_piece)];                                   \
        R##7 = src_[min(first_ * 8 + 7 * 64 + lane, last_piece)];                                   \
        const int jb_ = (int)mod_small((FM) + lane, a.nb, magic);                                   \
        const long long prev_ = first_ + lane - prev_distance(jb_, a.nb, a.hv);                     \
        /* lanes that need nothing from outside read the chunk's first line, loaded anyway */       \
        PDC = (COEF)[((prev_ >= 0 && prev_ < first_) ? prev_ : min(first_, a.nblocks - 1)) * 64];   \
    } while (0)

// Persistent: a wave takes chunks blockIdx.x * ENT_WAVES + wave, + gridDim.x * ENT_WAVES, ...
// BATCH: the chunks of nimg equally sized images are numbered through (image i has chunks
// i * nchunks .. (i + 1) * nchunks - 1); every image starts its own DC predictor chains.
// (LDS limits a SIMD to 3 waves, so the register budget may be spent on keeping the prefetch in registers)
template <bool BATCH>
__global__ __launch_bounds__(ENT_THREADS) __attribute__((amdgpu_waves_per_eu(2, 3))) void k_entropy(EntArgs a) {
    __shared__ uint32_t sLut[LUT_WORDS];
    __shared__ __attribute__((aligned(16))) uint8_t sCoef[ENT_WAVES][CHUNK_BLOCKS * SLOT_BYTES];
    __shared__ uint32_t sOut[ENT_WAVES][OUT_WORDS + 2];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const long long stride = (long long)gridDim.x * ENT_WAVES;
    long long chunk = (long long)blockIdx.x * ENT_WAVES + wave;
    const long long total_chunks = a.nchunks * a.nimg;
    const uint32_t magic = (65536u + a.nb - 1) / a.nb;            // x / nb == (x * magic) >> 16 for small x
    const uint32_t fm_step = (uint32_t)((stride * CHUNK_BLOCKS) % a.nb);
    const long long last_piece = a.nblocks * 8 - 1;
    // image, chunk inside the image and (first block of the chunk) mod nb, for chunk number c
    long long img, lc;
    uint32_t fm;
    auto locate = [&](long long c) {
        if (BATCH) {
            img = c / a.nchunks;
            lc = c - img * a.nchunks;
            fm = (uint32_t)(lc * CHUNK_BLOCKS) % (uint32_t)a.nb;  // (a batch image has fewer than 2^31 blocks)
        } else {
            img = 0;
            lc = c;
        }
    };
    if (!BATCH) fm = (uint32_t)((chunk * CHUNK_BLOCKS) % a.nb);
    locate(min(chunk, total_chunks - 1));
    uint4 pre0, pre1, pre2, pre3, pre4, pre5, pre6, pre7;   // (an array would be kept in scratch memory)
    int pre_pdc;
    JPG_CHUNK_LOAD(a.coef + img * a.nblocks * 64, lc, fm, pre, pre_pdc);

    for (int i = tid; i < LUT_WORDS; i += ENT_THREADS) sLut[i] = (&HUFF_DEV.e[0][0])[i];
    __syncthreads();

    uint8_t *coefs = sCoef[wave];
    uint32_t *out = sOut[wave];
    const uint16_t *slot = reinterpret_cast<const uint16_t *>(&coefs[lane * SLOT_BYTES]);
    for (; chunk < total_chunks; chunk += stride) {
        const long long first = lc * CHUNK_BLOCKS;                // within the image
        const int nvalid = (int)min((long long)CHUNK_BLOCKS, a.nblocks - first);
        const long long slot_index = img * a.cstride + lc;        // where the chunk's results go
        DevStatus *status = a.status + img;

        // stage the chunk's coefficients, one padded slot per block
        {
            // 16-byte piece i * 64 + lane of the chunk goes to block (piece >> 3), offset (piece & 7) * 16
            uint8_t *d = &coefs[(lane >> 3) * SLOT_BYTES + (lane & 7) * 16];
            constexpr int STEP = 8 * SLOT_BYTES;
            *reinterpret_cast<uint4 *>(d + 0 * STEP) = pre0;
            *reinterpret_cast<uint4 *>(d + 1 * STEP) = pre1;
            *reinterpret_cast<uint4 *>(d + 2 * STEP) = pre2;
            *reinterpret_cast<uint4 *>(d + 3 * STEP) = pre3;
            *reinterpret_cast<uint4 *>(d + 4 * STEP) = pre4;
            *reinterpret_cast<uint4 *>(d + 5 * STEP) = pre5;
            *reinterpret_cast<uint4 *>(d + 6 * STEP) = pre6;
            *reinterpret_cast<uint4 *>(d + 7 * STEP) = pre7;
        }
        const int pdc_outside = pre_pdc;
        const uint32_t fm_now = fm;
        // the next chunk's loads fly while this one is coded
        if (!BATCH) fm = mod_small(fm + fm_step, a.nb, magic);
        locate(min(chunk + stride, total_chunks - 1));
        JPG_CHUNK_LOAD(a.coef + img * a.nblocks * 64, lc, fm, pre, pre_pdc);
        __builtin_amdgcn_wave_barrier();

        const bool valid = lane < nvalid;
        uint32_t mlo = 0, mhi = 0;
        int dc_diff = 0;
        uint32_t chroma = 0;
        {
            uint4 q[8];
#pragma unroll
            for (int i = 0; i < 8; i++) q[i] = reinterpret_cast<const uint4 *>(slot)[i];
            nonzero_mask(q, mlo, mhi);
            mlo &= ~1u;                                           // the DC symbol is coded separately
            // DC difference against the previous block of the same component
            // (three predictor chains, src/jpeg/entropy_coding.rs:38-40,149-150)
            const int jb = (int)mod_small(fm_now + lane, a.nb, magic);
            chroma = jb >= a.hv ? 1u : 0u;
            const int back = prev_distance(jb, a.nb, a.hv);
            int pdc;
            // (selects, not an indexed read of the argument block: that would be a memory load)
            if (first + lane - back < 0) pdc = jb < a.hv ? a.pred[0] : (jb == a.hv ? a.pred[1] : a.pred[2]);
            else if (lane < back) pdc = pdc_outside;
            else pdc = (int16_t)reinterpret_cast<const uint16_t *>(&coefs[(lane - back) * SLOT_BYTES])[0];
            dc_diff = (int16_t)((int)(int16_t)(q[0].x & 0xffff) - pdc);
            if (!valid) { mlo = 0; mhi = 0; dc_diff = 0; }
        }
        const uint32_t *lut = sLut + chroma * jpgtab::HUFF_LUT_STRIDE;

        // 1. lengths
        AccWriter acc;                                            // blocks of up to ACC_BITS bits are kept whole
        const uint32_t len = valid ? code_block(slot, mlo, mhi, dc_diff, lut, status, a.epoch, acc) : 0u;
        const uint32_t sc = wave_incl_scan(len);                  // bits up to and including this lane's block
        const uint32_t total = __shfl(sc, 63);
        if (lane == 0) a.clen[slot_index] = total;

        // 2. code and store; a chunk too long for the LDS window goes in several passes, each
        // starting in the word the previous one ended in (that word is carried over and stored again)
        uint32_t *dst = a.sparse + (size_t)slot_index * CHUNK_STRIDE_WORDS;
        constexpr uint32_t WINDOW = OUT_WORDS * 32;               // bits per pass
        uint32_t done = 0;                                        // bits of the chunk already stored
        uint32_t carry = 0;                                       // the bits before `done` in its word
        for (int pass = 0; done < total || pass == 0; pass++) {
            const uint32_t start = done & 31;
            // lanes whose block ends within the window (at least one: a block is shorter than WINDOW - 31 bits)
            const bool mine = valid && sc > done && sc - done + start <= WINDOW;
            const uint32_t pass_bits = __shfl(mine ? sc - done : 0u, 63 - __clzll(__ballot(mine) | 1ull));
            const uint32_t nwords = (start + pass_bits + 31) >> 5;
            for (uint32_t i = lane; i < nwords + 1; i += 64) out[i] = i ? 0u : carry;
            __builtin_amdgcn_wave_barrier();
            if (mine) {
                const uint32_t pos = sc - len - done + start;
                if (len <= ACC_BITS) {
                    acc.emit(out, pos, len);
                } else {                                          // long block: walk it again
                    BitWriter w;
                    w.init(out, pos);
                    code_block(slot, mlo, mhi, dc_diff, lut, status, a.epoch, w);
                }
            }
            __builtin_amdgcn_wave_barrier();
            for (uint32_t i = lane; i < nwords; i += 64) dst[(done >> 5) + i] = out[i];
            const uint32_t endb = start + pass_bits;
            carry = (endb & 31) ? out[endb >> 5] : 0u;
            __builtin_amdgcn_wave_barrier();
            done += pass_bits;
        }
    }
}

// ---------------------------------------------------------------- stream readers

struct StreamArgs {
    const uint32_t *sparse;  // chunk slots written by k_entropy
    const uint32_t *clen;    // bits per chunk
    const uint32_t *gsum;    // bits per group of 64 chunks (k_group_sums)
    size_t gsum_stride;      // groups per image in gsum
    long long nchunks;       // per image
    int ngroups;             // groups of 64 chunks of an image (<= MAX_GROUPS)
    // a batch: image i is handled by workgroups i * wpi .. (i + 1) * wpi - 1 and owns a stride of each array
    int wpi;
    long long cstride;       // chunk slots per image in sparse / clen
    size_t slot_words;       // words per chunk slot
    size_t out_stride;       // bytes per image in `out`
    size_t agg_stride, grp_stride, hyp_stride;   // words per image in agg / grp / hyp
    DevStatus *status;       // [nimg] nbits, scan_bytes and tail out
    uint32_t shift;          // place the stream `shift` (0..7) bits later
    uint32_t head;           // bits OR-ed into the first byte (top `shift` bits)
    const ChainParams *chain;   // both are read from here instead when not null (left there by a kernel queued before this one)
    int is_last;             // keep the trailing partial byte
    uint64_t *agg, *grp, *hyp;  // three-level sums of 0xFF counts per segment
    uint32_t *ticket;        // [nimg][2] next segment to hand out, workgroups that have left (both 0 between launches)
    uint32_t epoch;          // generation of those words
    uint32_t err_epoch;      // epoch errors are tagged with (that of the entropy launch)
    uint8_t *out;            // stuffed bytes, or the dense stream for k_stream<false>
    DevStatus *host_status;  // [nimg] pinned host copy, filled at the end (may be null)
    unsigned long long *trace;  // debug builds only
};

typedef uint16_t u16_unaligned __attribute__((aligned(1)));

// number of 0xFF bytes in a word
__device__ __forceinline__ uint32_t count_ff(uint32_t w) {
    const uint32_t z = ~w;
    uint32_t t = (z & 0x7f7f7f7fu) + 0x7f7f7f7fu;
    t = ~(t | z | 0x7f7f7f7fu);
    return __popc(t);
}

constexpr int SEG_BITS = STUFF_SEG * 8;
constexpr int TABLE = 256;   // chunks whose offsets a workgroup tabulates per segment
constexpr int SEG_WORDS = SEG_BITS / 32;
constexpr int DENSE_WORDS = SEG_WORDS + 2;   // the segment, plus the byte after it (tail of a stripe)
// words a segment can need from the chunk slots: its own, plus a partial one at either end of each
// of the at most 130 chunks under it (a full chunk has at least 64 x 4 bits)

// index of the last entry <= r in the ascending table T[0..TABLE], or -1 if there is none
__device__ __forceinline__ int find_last_le(const int *T, int r) {
    int lo = -1, hi = TABLE;             // answer in [lo, hi]
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (T[mid] <= r) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// Workgroup barrier that orders LDS accesses only.  __syncthreads() also waits for every global
// store of the wave to be acknowledged (about 2 us for a store that has to reach the other XCDs);
// inside k_stream the barriers only guard LDS, and data loaded from global memory is waited for
// where it is used.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

#ifdef JPG_TRACE
// debug build: wall-clock time (100 MHz) a workgroup spends in each phase, summed over its segments: phase K runs
// from the stamp before it to JPG_STAMP(K) (single images; stored when the workgroup leaves)
#define JPG_STAMP(K) do { const unsigned long long now_ = wall_clock64(); stamps[K] += now_ - last_stamp; last_stamp = now_; } while (0)
#else
#define JPG_STAMP(K) do { } while (0)
#endif

// bits of every group of 64 chunks of every image (blockIdx.y): a wave per group
__global__ __launch_bounds__(1024) void k_group_sums(const uint32_t *__restrict__ clen, long long nchunks, long long cstride,
                                                     int ngroups, uint32_t *__restrict__ gsum, size_t gstride) {
    const int lane = threadIdx.x & 63, g = blockIdx.x * 16 + (threadIdx.x >> 6);
    if (g >= ngroups) return;
    const long long c = (long long)g * 64 + lane;
    const uint32_t v = c < nchunks ? clen[blockIdx.y * cstride + c] : 0u;
    const uint32_t sum = wave_incl_scan(v);
    if (lane == 63) gsum[blockIdx.y * gstride + g] = sum;
}

// Threads per segment: 512 for a single image, whose stuffing nothing else hides (the kernel is a chain of short
// dependent steps, and more waves keep the per-thread instruction count of every step small); 256 for a batch,
// which is stuffed beside the coding kernel of the next batch: one wave per SIMD, 76 registers and under 8 KiB of
// LDS fit into what four workgroups of the coding kernel leave free on a CU.
#ifndef JPG_ST_THREADS_SINGLE
#define JPG_ST_THREADS_SINGLE 512
#endif
constexpr int ST_THREADS_SINGLE = JPG_ST_THREADS_SINGLE, ST_THREADS_BATCH = 256;
// teams of workgroups that share a ticket counter (words of an image's counters: [0] the image's counter, [1] workgroups
// that have left, from TEAM_BASE on a record of TEAM_WORDS per team)
constexpr int TEAM = 32, TEAM_RING = 8, TEAM_WORDS = 64, TEAM_BASE = 64, MAX_TEAMS = (TICKET_WORDS_SINGLE - TEAM_BASE) / TEAM_WORDS;
// fronts of later segments between a segment's front and its back: two for a single image (measured: 1 % better than
// one; 256 threads per segment 15 % worse than 512), one for a batch (fewer registers)
#ifndef JPG_ST_DEPTH
#define JPG_ST_DEPTH 2
#endif
// waves per SIMD the register budget of the stuffing kernel is set for.  It needs 76 registers: within 64 (eight
// waves) it spills, which costs a batch of small images 5 %; with six waves (single images: three workgroups of
// 512 threads per CU) or four (batches: one wave per SIMD beside the coding kernel's) it does not, at the same speed.
#ifndef JPG_ST_WAVES_BATCH
#define JPG_ST_WAVES_BATCH 4
#endif
#ifndef JPG_ST_WAVES_SINGLE
#define JPG_ST_WAVES_SINGLE 6
#endif
constexpr int ST_DEPTH_SINGLE = JPG_ST_DEPTH;

// STUFF: byte stuffing of the (shifted) stream into a.out, a workgroup per segment of STUFF_SEG stream bytes at a
// time (segments are drawn from ticket counters).  !STUFF: the stream itself, dense (segments blockIdx.x,
// + gridDim.x, ...).  Dynamic LDS: (ngroups + 1) 64-bit prefix sums of the chunk lengths per group.
// The work on a segment is a chain of short dependent steps; a workgroup works on several segments in turn.
// DEPTH: fronts of later segments between a segment's front and its back (see the three steps below).
// TEAMS: ticket counters per team of workgroups (single images).
template <bool STUFF, int ST_THREADS, int DEPTH, bool TEAMS>
// (JPG_ST_WAVES_*: waves per SIMD the register budget is set for)
__global__ __launch_bounds__(ST_THREADS)
__attribute__((amdgpu_waves_per_eu(ST_THREADS == ST_THREADS_BATCH ? JPG_ST_WAVES_BATCH : JPG_ST_WAVES_SINGLE, 8))) void k_stream(StreamArgs a) {
    constexpr int ST_WAVES = ST_THREADS / 64;
    constexpr int WPT = SEG_WORDS / ST_THREADS;   // 32-bit words of the segment per thread
    constexpr int GATHER_ROUNDS = (DENSE_WORDS + 2 * 132 + ST_THREADS - 1) / ST_THREADS;
    static_assert(SEG_WORDS == WPT * ST_THREADS && ST_THREADS >= TABLE, "whole words per thread");
    static_assert(STUFF_SEG <= (1 << 13) && MAX_GROUPS <= 64 * 64 * 64, "count of a segment; two rounds of 64 probes");
    if (a.chain) {
        a.shift = a.chain->shift;
        a.head = a.chain->head;
    }
    {   // this workgroup's image: from here on `a` describes that image alone
        const int img = blockIdx.x / a.wpi;
        a.sparse += (size_t)img * a.cstride * a.slot_words;
        a.clen += img * a.cstride;
        if (a.gsum) a.gsum += img * a.gsum_stride;
        a.out += img * a.out_stride;
        a.agg += img * a.agg_stride;
        a.grp += img * a.grp_stride;
        a.hyp += img * a.hyp_stride;
        if (STUFF) a.ticket += 2 * img;
        a.status += img;
        if (a.host_status) a.host_status += img;
    }
    const unsigned wg = blockIdx.x % a.wpi, nwg = a.wpi;   // position among the workgroups of the image
    extern __shared__ uint64_t sG[];     // [ngroups + 1] bits before each group of 64 chunks
    __shared__ uint64_t sPart[ST_WAVES];
    __shared__ int sT[TABLE + 1];          // chunk offsets relative to the segment's first bit
    __shared__ int sW[TABLE + 1];          // running count of the words wanted from each chunk
    __shared__ int sW0[TABLE];             // first word wanted from each chunk
    __shared__ uint32_t sD[DENSE_WORDS];   // the segment's bits in place
    __shared__ uint32_t s[ST_WAVES];
    __shared__ uint64_t sBefore[2];        // (in turn: two backs may follow each other with one barrier between them)
    __shared__ uint32_t sSeg;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Stuffing: segments are handed out by ticket counters, lowest first, so every segment with a lower
    // number than one a workgroup holds is held by a workgroup that is running (or has finished): the waits
    // below make progress however few workgroups are resident and whatever else runs on the device.  The
    // ticket of the following turn is drawn a turn ahead, by a thread of the last wave.
    const bool ticketer = tid == ST_THREADS - 1;
    uint32_t ticket = 0;
#ifdef JPG_TRACE
    unsigned long long stamps[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long last_stamp = wall_clock64();
#endif

    // ---- bits per group, then their running sums.  A single image: summed up by a kernel queued before this one
    // (k_group_sums, or k_tail for a stripe of a chain; a thousand workgroups adding up the same lengths took a fifth
    // of the kernel's time); a batch: the few workgroups of an image add up its lengths themselves.
    for (int g = tid; g < a.ngroups; g += ST_THREADS) {
        if (a.gsum) { sG[g] = a.gsum[g]; continue; }
        const long long c0 = (long long)g * 64;
        uint32_t sum = 0;
        if (c0 + 64 <= a.nchunks) {
            const uint4 *p = reinterpret_cast<const uint4 *>(a.clen + c0);
            uint4 v[8];                                          // two batches of loads: fewer registers
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = p[k];
#pragma unroll
            for (int k = 0; k < 8; k++) sum += v[k].x + v[k].y + v[k].z + v[k].w;
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = p[8 + k];
#pragma unroll
            for (int k = 0; k < 8; k++) sum += v[k].x + v[k].y + v[k].z + v[k].w;
        } else {
            for (long long c = c0; c < a.nchunks; c++) sum += a.clen[c];
        }
        sG[g] = sum;
    }
    lds_barrier();
    JPG_STAMP(0);
    const int per = (a.ngroups + ST_THREADS - 1) / ST_THREADS;
    const int lo = min(tid * per, a.ngroups), hi = min(lo + per, a.ngroups);
    uint64_t loc = 0;
    for (int g = lo; g < hi; g++) loc += sG[g];
    // inclusive scan of the partial sums (each < 2^26) as two 32-bit wave scans: low 16 bits, rest
    const uint32_t slo = wave_incl_scan((uint32_t)loc & 0xffffu), shi = wave_incl_scan((uint32_t)(loc >> 16));
    if (lane == 63) sPart[wave] = ((uint64_t)shi << 16) + slo;
    lds_barrier();
    uint64_t incl = ((uint64_t)shi << 16) + slo, nbits = 0;
#pragma unroll
    for (int w = 0; w < ST_WAVES; w++) {
        const uint64_t p = sPart[w];
        if (w < wave) incl += p;
        nbits += p;
    }
    uint64_t run = incl - loc;
    for (int g = lo; g < hi; g++) {
        const uint64_t v = sG[g];
        sG[g] = run;
        run += v;
    }
    if (tid == 0) sG[a.ngroups] = nbits;
    lds_barrier();
    JPG_STAMP(0);

    const uint64_t tbits = nbits + a.shift;
    const uint64_t nbytes = (STUFF && !a.is_last) ? tbits >> 3 : (tbits + 7) >> 3;
    const uint64_t nseg = (nbytes + STUFF_SEG - 1) / STUFF_SEG;
    const uint64_t nloop = nseg ? nseg : 1;          // the tail of an empty stripe still has to be reported
    const Sums sums{a.agg, a.grp, a.hyp, a.epoch, (long long)nseg};

    // A segment goes through three steps, and a workgroup works on three segments at a time, a step of each in
    // turn, so that every wait for memory or for another workgroup has a step of another segment to hide behind:
    //   ahead: the group of chunks its first bit lies in is looked up, the lengths of the chunks from there on asked for
    //   front: the segment's bits are put in place (sD) and counted, the count published; waits for no other workgroup
    //   back:  the 0xFF bytes of all segments before it (asked for in the middle of the next segment's front,
    //          by which time they have mostly been published), then its bytes, stuffed, to their place
    struct Ahead {
        int g0;              // group of 64 chunks the segment's first bit lies in
        uint32_t v;          // this thread's chunk length from the start of that group on (first TABLE threads)
    };
    struct Held {
        uint32_t t[WPT];     // the thread's words of the segment
        uint32_t before_in_seg, total;   // 0xFF bytes: in the words of the threads before, in the segment
        uint64_t seg;
        bool valid;
    };

    // Tickets.  All workgroups of a device drawing from one counter get a number every 12 to 15 ns between them:
    // less than the 4 KiB segments of one large stream need.  TEAMS (single images): the workgroups draw in teams
    // of TEAM from a counter of their team; the first of every TEAM draws of a team (the leader of that round)
    // draws a window of TEAM consecutive segments from the counter of the image and publishes it in the team's
    // record for the other TEAM - 1, who find their segment in it by the number they drew.  Windows are handed
    // to teams, and their segments to workgroups, in the order of the draws, by workgroups that are running:
    // every segment below one that is held is held too, or is the next its team hands out.  The record keeps the
    // windows of TEAM_RING rounds: a workgroup looks its window up a turn after it drew, and none gets more than a
    // few turns ahead of the slowest (it would wait for that one's counts); a slot found overwritten is an error.
    uint32_t *const team_rec = a.ticket + TEAM_BASE + (wg / TEAM) * TEAM_WORDS;      // [0] the team's counter
    uint64_t *const team_win = reinterpret_cast<uint64_t *>(team_rec + 32);          // [TEAM_RING] {round + 1 : 32 | window : 32}
    uint32_t window = 0;
    int drawing = 0;                     // ticketer: 1 drawn from the team, 2 leader has drawn a window, 3 and published it
    auto draw_begin = [&]() {
        if (!ticketer) return;
        ticket = atomicAdd(TEAMS ? team_rec : &a.ticket[0], 1u);
        drawing = 1;
    };
    auto draw_window = [&]() {           // (waits for the team's number)
        if (!TEAMS || !ticketer || drawing != 1) return;
        if ((ticket & (TEAM - 1)) == 0) window = atomicAdd(&a.ticket[0], 1u);
        drawing = 2;
    };
    auto draw_publish = [&]() {          // (waits for the window)
        if (!TEAMS || !ticketer || drawing != 2) return;
        if ((ticket & (TEAM - 1)) == 0)
            hw_store(&team_win[(ticket / TEAM) % TEAM_RING], ((uint64_t)(ticket / TEAM + 1) << 32) | window);
        drawing = 3;
    };

    auto ahead = [&](const uint64_t seg) {
        const long long P0 = (long long)seg * SEG_BITS - (long long)a.shift;   // stream position of the segment's first bit
        const uint64_t p0 = P0 > 0 ? (uint64_t)P0 : 0;
        // the last group that starts at or before p0 (sG ascends from 0): two rounds of 64 probes, one per lane
        const int step = (a.ngroups + 63) >> 6;
        const int i1 = lane * step;
        const int c1 = __popcll(__ballot(i1 < a.ngroups && sG[i1] <= p0)) - 1;    // >= 0: sG[0] = 0
        int l = c1 * step - 1;
        for (int o = lane; o - lane < step; o += 64) {           // (step <= 64 up to 4096 groups)
            const int i2 = c1 * step + o;
            l += __popcll(__ballot(o < step && i2 < a.ngroups && sG[i2] <= p0));
        }
        const long long cbase = (long long)l * 64;
        return Ahead{l, (tid < TABLE && cbase + tid < a.nchunks) ? a.clen[cbase + tid] : 0u};
    };

    // the counts of the segments before `seg`, asked for (not waited for)
    auto ask = [&](const uint64_t seg) {
        Probe pr{0, 0, 0};
        if (nseg && wave == 0) pr = sums_probe(sums, (long long)seg, lane);
        return pr;
    };

    // asked: the counts of the segments before ask_seg (whose back follows this front) are asked for half-way, if wanted
    auto front = [&](const uint64_t seg, const Ahead ah, Held &hd, const bool wanted, const uint64_t ask_seg, Probe &asked) {
        hd.seg = seg;
        hd.valid = true;
        // ---- offsets of the chunks under this segment (first TABLE threads)
        const long long P0 = (long long)seg * SEG_BITS - (long long)a.shift;
        const int g0 = ah.g0;
        const long long cbase = (long long)g0 * 64;
        const uint32_t v = ah.v;
        const uint32_t vs = wave_incl_scan(v);
        if (lane == 63) s[wave] = vs;
        lds_barrier();
        if (tid < TABLE) {
            uint32_t ex = vs - v;
            for (int w = 0; w < wave; w++) ex += s[w];
            const long long rel0 = (long long)sG[g0] - P0;
            sT[tid] = (int)(rel0 + ex);
            if (tid == TABLE - 1) sT[TABLE] = (int)(rel0 + ex + v);
        }
        lds_barrier();
        JPG_STAMP(2);

        // ---- gather: every word of a chunk slot that overlaps the segment is OR-ed into place
        const uint32_t *slots = a.sparse + (size_t)cbase * a.slot_words;
        {
            int n = 0;
            if (tid < TABLE) {
                const int Ti = sT[tid], L = sT[tid + 1] - Ti;
                const int b0 = max(0, -Ti), b1 = min(L, DENSE_WORDS * 32 - Ti);   // bits of the chunk inside the window
                const int w0 = b0 >> 5;
                n = b1 > b0 ? ((b1 + 31) >> 5) - w0 : 0;
                sW0[tid] = w0;
            }
            const uint32_t ns = wave_incl_scan((uint32_t)n);
            if (lane == 63) s[wave] = ns;                        // (every thread read s[] before the last barrier)
            lds_barrier();
            if (tid < TABLE) {
                uint32_t nex = ns - n;
                for (int w = 0; w < wave; w++) nex += s[w];
                sW[tid] = (int)nex;
                if (tid == TABLE - 1) sW[TABLE] = (int)(nex + n);
            }
        }
        for (int k = tid; k < DENSE_WORDS; k += ST_THREADS) sD[k] = 0;
        lds_barrier();
        JPG_STAMP(3);
        if (STUFF) draw_window();
        {
            // thread t takes wanted words t * rounds .. t * rounds + rounds - 1: one search, then a walk
            const int nwant = sW[TABLE];
            const int rounds = (nwant + ST_THREADS - 1) / ST_THREADS;   // <= GATHER_ROUNDS
            uint32_t wv[GATHER_ROUNDS];
            int pos[GATHER_ROUNDS];
            int f = tid * rounds;
            int i = f < nwant ? find_last_le(sW, f) : TABLE - 1;  // the last of equal entries is the non-empty one
#pragma unroll
            for (int m = 0; m < GATHER_ROUNDS; m++) {
                wv[m] = 0;
                pos[m] = 0;
                if (m < rounds && f < nwant) {
                    while (sW[i + 1] <= f) i++;
                    const int j = sW0[i] + (f - sW[i]);
                    wv[m] = slots[(uint32_t)i * (uint32_t)a.slot_words + (uint32_t)j];   // (256 slots: 32 bits do)
                    pos[m] = sT[i] + 32 * j;                      // where the word's first bit goes (>= -31)
                }
                f++;
            }
#pragma unroll
            for (int m = 0; m < GATHER_ROUNDS; m++) {
                if (wv[m]) {
                    const int k = pos[m] >> 5, sh = pos[m] & 31;
                    if (k >= 0 && k < DENSE_WORDS) atomicOr(&sD[k], wv[m] >> sh);
                    if (sh && k + 1 >= 0 && k + 1 < DENSE_WORDS) atomicOr(&sD[k + 1], wv[m] << (32 - sh));
                }
            }
        }
        if (STUFF) draw_publish();                               // (the words gathered have arrived, and the window before them)
        // the counts the back that follows needs: asked for here, behind the phase that needs the most registers,
        // they arrive while the segment is counted
        if (STUFF && wanted) asked = ask(ask_seg);
        lds_barrier();
        JPG_STAMP(4);

        // ---- this thread's WPT words
        const uint64_t q = seg * SEG_WORDS + (uint64_t)WPT * tid;  // first word of the stream
#pragma unroll
        for (int k = 0; k < WPT; k++) {
            hd.t[k] = 0;
            if (4 * (q + k) < nbytes) {
                hd.t[k] = sD[WPT * tid + k];
                if (q + k == 0 && a.shift) hd.t[k] |= a.head << 24;
                // a stripe that is not the last one ends at a byte boundary: drop what follows
                const uint64_t rem = nbytes - 4 * (q + k);
                if (rem < 4) hd.t[k] &= ~0u << (32 - 8 * (int)rem);
            }
        }
        if (!STUFF) {
#pragma unroll
            for (int k = 0; k < WPT; k++)
                if (4 * (q + k) < nbytes) reinterpret_cast<uint32_t *>(a.out)[q + k] = __builtin_bswap32(hd.t[k]);   // padded buffer
            if (seg == nloop - 1 && tid == 0) {
                a.status->nbits = nbits;
                if (a.host_status) *a.host_status = *a.status;
            }
            lds_barrier();
            return;
        }
        uint32_t c = 0;
#pragma unroll
        for (int k = 0; k < WPT; k++) c += count_ff(hd.t[k]);
        const uint32_t sc = wave_incl_scan(c);
        if (lane == 63) s[wave] = sc;                            // (s[] was last read before the previous barrier)
        lds_barrier();
        JPG_STAMP(5);
        uint32_t total = 0, before_in_seg = sc - c;
#pragma unroll
        for (int w = 0; w < ST_WAVES; w++) {
            const uint32_t p = s[w];
            if (w < wave) before_in_seg += p;
            total += p;
        }
        total = (uint32_t)__builtin_amdgcn_readfirstlane((int)total);
        hd.before_in_seg = before_in_seg;
        hd.total = total;
        // ---- the segment's count, published; the last segment of a group of 64 adds up the group
        if (nseg && wave == 0) {
            if (lane == 0) hw_store(&a.agg[seg], hw_pack(a.epoch, total));
            if ((seg & 63) == 63) {
                bool timeout = false;
                sums_close<13>(sums, (long long)seg, total, lane, timeout);   // (<= STUFF_SEG per segment)
                if (timeout && lane == 0) raise_error(a.status, a.err_epoch, JPG_ERR_HIP);
            }
        }
    };

    int backs = 0;
    auto back = [&](const Held &hd, const Probe &probe) {
        const uint64_t seg = hd.seg;
        const uint32_t total = hd.total;
        bool timeout = false;
        backs ^= 1;
        if (wave == 0) {
            uint64_t before = 0;
            if (nseg) before = sums_prefix(sums, (long long)seg, probe, lane, timeout);
            if (lane == 0) sBefore[backs] = before;
        }
        if (__ballot(timeout) && lane == 0) raise_error(a.status, a.err_epoch, JPG_ERR_HIP);
        lds_barrier();
        JPG_STAMP(6);
        // (the same in every lane: the address of the segment's output stays in scalar registers)
        const uint64_t before_read = sBefore[backs];
        const uint64_t before_segs = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(before_read >> 32)) << 32) |
                                     (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)before_read);
        const uint64_t q = seg * SEG_WORDS + (uint64_t)WPT * tid;
        if (seg == nloop - 1 && tid == 0) {
            a.status->nbits = nbits;
            a.status->scan_bytes = nbytes + before_segs + total;
            // bits of the partial byte a non-final stripe leaves to its successor (no front follows the front of
            // the last segment: its bits are still in sD)
            uint32_t tail = 0;
            if (!a.is_last && (tbits & 7)) {
                const int r = (int)(8 * nbytes - seg * SEG_BITS);
                tail = (sD[r >> 5] >> (24 - (r & 31))) & 0xffu;
                if (nbytes == 0) tail |= a.head;
            }
            a.status->tail = tail;
            if (a.host_status) *a.host_status = *a.status;
        }
        // Every byte but the thread's last is stored as two: itself and a zero behind it.  The next byte lands on that
        // zero, or behind it when the byte was 0xFF (the stores of a lane are performed in order; two-byte stores
        // need no alignment): no branch, no mask, and the destination needs no alignment either.
        uint8_t *const ob = a.out + seg * STUFF_SEG + before_segs;
        uint32_t off = 4 * WPT * tid + hd.before_in_seg;
        if (4 * (q + WPT) <= nbytes) {
#pragma unroll
            for (int k = 0; k < 4 * WPT - 1; k++) {
                const uint32_t b = (hd.t[k >> 2] >> (24 - 8 * (k & 3))) & 0xffu;
                *reinterpret_cast<u16_unaligned *>(ob + off) = (uint16_t)b;
                off += 1 + (b == 0xffu);
            }
            const uint32_t b = hd.t[WPT - 1] & 0xffu;
            ob[off] = (uint8_t)b;
            if (b == 0xffu) ob[off + 1] = 0;
        } else {                                                 // the last bytes of the stream
            const uint32_t have = 4 * q < nbytes ? (uint32_t)(nbytes - 4 * q) : 0u;
#pragma unroll
            for (int k = 0; k < 4 * WPT; k++) {
                if ((uint32_t)k < have) {
                    const uint8_t b = (uint8_t)(hd.t[k >> 2] >> (24 - 8 * (k & 3)));
                    ob[off++] = b;
                    if (b == 0xff) ob[off++] = 0;
                }
            }
        }
        JPG_STAMP(7);
    };

    Probe asked{0, 0, 0};
    if (!STUFF) {
        Held hd;
        for (uint64_t seg = wg; seg < nloop; seg += nwg) front(seg, ahead(seg), hd, false, 0, asked);
        return;
    }
    auto next_segment = [&]() {
        if (ticketer) {
            uint32_t seg32 = ticket;
            if (TEAMS) {
                draw_window();
                draw_publish();
                const uint32_t round = ticket / TEAM, idx = ticket & (TEAM - 1);
                uint64_t w = ((uint64_t)(round + 1) << 32) | window;
                if (idx) {
                    const uint64_t *slot = &team_win[round % TEAM_RING];
                    uint32_t spin = 0;
                    for (;;) {
                        w = hw_load(slot);
                        if ((uint32_t)(w >> 32) == round + 1) break;
                        if (++spin >= SPIN_LIMIT || (uint32_t)(w >> 32) > round + 1) {   // (never: the leader is running)
                            raise_error(a.status, a.err_epoch, JPG_ERR_HIP);
                            w = 0xffffffffull / TEAM;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                const uint64_t seg = (w & 0xffffffffull) * TEAM + idx;
                seg32 = seg < 0xffffffffull ? (uint32_t)seg : 0xffffffffu;
            }
            sSeg = seg32;
        }
        lds_barrier();
        // (the same in every lane: kept in scalar registers, as is what follows from it)
        const uint64_t seg = (uint32_t)__builtin_amdgcn_readfirstlane((int)sSeg);
        lds_barrier();                                           // (sSeg may be written again)
        // the next draw: a turn before its result is needed
        if (seg < nloop) draw_begin();
        JPG_STAMP(1);
        return seg;
    };
    draw_begin();
    // segments put in place that wait for their back, oldest first
    Held ring[DEPTH];
#pragma unroll
    for (int j = 0; j < DEPTH; j++) ring[j].valid = false;
    uint64_t coming = next_segment();
    Ahead ac{0, 0u};
    if (coming < nloop) ac = ahead(coming);
    for (;;) {
        const bool have = coming < nloop;
        bool any = have;
#pragma unroll
        for (int j = 0; j < DEPTH; j++) any = any || ring[j].valid;
        if (!any) break;
        const Held out = ring[0];
#pragma unroll
        for (int j = 0; j + 1 < DEPTH; j++) ring[j] = ring[j + 1];
        ring[DEPTH - 1].valid = false;
        if (have) {
            const uint64_t seg = coming;
            const Ahead af = ac;
            coming = next_segment();
            if (coming < nloop) ac = ahead(coming);
            front(seg, af, ring[DEPTH - 1], out.valid, out.seg, asked);
        } else if (out.valid) {
            asked = ask(out.seg);
        }
        if (out.valid) back(out, asked);
    }
#ifdef JPG_TRACE
    if (tid == 0 && a.trace && blockIdx.x < 4096)
        for (int k = 0; k < 8; k++) a.trace[blockIdx.x * 8 + k] = stamps[k];
#endif
    // the last workgroup of the image to leave puts the counters back to 0 for the next launch (every
    // workgroup has drawn its last ticket before it counts itself out)
    if (ticketer) sSeg = atomicAdd(&a.ticket[1], 1u) == nwg - 1;
    lds_barrier();
    if (sSeg) {
        if (TEAMS)
            for (unsigned i = tid; i < ((nwg + TEAM - 1) / TEAM) * TEAM_WORDS; i += ST_THREADS)
                __hip_atomic_store(&a.ticket[TEAM_BASE + i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid < 2) __hip_atomic_store(&a.ticket[tid], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Length and last bits of the stream the chunks add up to, for a stripe: what the stripes after it need
// (where their bits start, and the bits that share a byte with their first ones) before anything is
// assembled.  One workgroup: a 64-bit sum of the chunk lengths, then one thread walks back from the last
// chunk until it has 32 bits.
// chain_slot: where a stripe of a chain publishes both for the stripes after it (host memory shared by the GPUs).
__global__ __launch_bounds__(1024) void k_tail(const uint32_t *__restrict__ sparse, const uint32_t *__restrict__ clen,
                                               long long nchunks, size_t slot_words, DevStatus *status, DevStatus *host_status,
                                               ChainMail *chain_slot, uint32_t chain_tag, uint32_t *__restrict__ gsum) {
    __shared__ uint64_t sSum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the lengths, 16 bytes per load (the array is padded to a multiple of 64 words; clen is 16-byte aligned)
    // (16 lanes in a row hold a group of 64 chunks: their sum goes to gsum for the stuffing kernel; the array is
    // padded with zeros to whole groups)
    uint64_t sum = 0;
    const long long quads = (nchunks + 63) / 64 * 16;
    const uint4 *q = reinterpret_cast<const uint4 *>(clen);
    constexpr int AHEAD = 4;                                     // loads in flight per thread: the kernel is one workgroup
    for (long long i0 = tid; i0 - tid < quads; i0 += 1024 * AHEAD) {   // (whole waves: the row sums below take all lanes)
        uint4 v[AHEAD];
#pragma unroll
        for (int k = 0; k < AHEAD; k++) {
            const long long i = i0 + 1024 * k;
            v[k] = uint4{0, 0, 0, 0};
            if (i < quads) v[k] = q[i];
        }
#pragma unroll
        for (int k = 0; k < AHEAD; k++) {
            const long long i = i0 + 1024 * k;
            const long long c = i * 4;                           // (what lies behind the last chunk is not defined)
            if (c + 3 >= nchunks) {
                if (c + 0 >= nchunks) v[k].x = 0;
                if (c + 1 >= nchunks) v[k].y = 0;
                if (c + 2 >= nchunks) v[k].z = 0;
                v[k].w = 0;
            }
            uint32_t part = v[k].x + v[k].y + v[k].z + v[k].w;
            sum += part;
            part += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)part, 0x111, 0xf, 0xf, false);   // row_shr:1
            part += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)part, 0x112, 0xf, 0xf, false);   // row_shr:2
            part += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)part, 0x114, 0xf, 0xf, false);   // row_shr:4
            part += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)part, 0x118, 0xf, 0xf, false);   // row_shr:8
            if (gsum && (tid & 15) == 15 && i < quads) gsum[i >> 4] = part;
        }
    }
    sum = wave_sum64(sum);
    if (lane == 0) sSum[wave] = sum;
    __syncthreads();
    if (tid != 0) return;
    uint64_t nbits = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) nbits += sSum[w];
    uint32_t acc = 0, got = 0;
    for (long long c = nchunks - 1; c >= 0 && got < 32; c--) {
        const uint32_t L = clen[c];
        if (L == 0) continue;
        const uint32_t take = min(32u - got, L), first = L - take;      // bits [first, L) of the chunk, MSB first
        const uint32_t *w = sparse + (size_t)c * slot_words + (first >> 5);
        const uint64_t two = ((uint64_t)w[0] << 32) | (((first & 31) + take > 32) ? w[1] : 0u);
        const uint32_t v = (uint32_t)((two << (first & 31)) >> (64 - take));
        acc |= got < 32 ? v << got : 0u;
        got += take;
    }
    status->nbits = nbits;
    status->last32 = acc;
    if (host_status) *host_status = *status;
    if (chain_slot) {
        __hip_atomic_store(&chain_slot->tail, ((uint64_t)chain_tag << 48) | acc, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&chain_slot->bits, ((uint64_t)chain_tag << 48) | (nbits & ((1ull << 48) - 1)), __ATOMIC_RELEASE,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

hipError_t launch_tail(const EntropyBufs &b, hipStream_t s) {
    if ((b.nimg > 0 ? b.nimg : 1) != 1 || b.slot_words <= 0) return hipErrorInvalidValue;   // single images only
    hipLaunchKernelGGL(k_tail, dim3(1), dim3(1024), 0, s, b.sparse, b.clen, b.nchunks, (size_t)b.slot_words, b.status,
                       b.host_status, b.chain_slot, b.chain_tag, b.gsum);
    return hipGetLastError();
}

hipError_t launch_entropy(const Geom &g, const int16_t *coef, const int16_t pred[3], const EntropyBufs &b,
                          hipStream_t s) {
    const long long nchunks = (g.nblocks + CHUNK_BLOCKS - 1) / CHUNK_BLOCKS;
    const int nimg = b.nimg > 0 ? b.nimg : 1;
    if (nchunks != b.nchunks || b.cstride < nchunks || b.slot_words != CHUNK_STRIDE_WORDS || (nimg > 1 && g.nblocks >= (1LL << 31))) return hipErrorInvalidValue;
    // as many workgroups as fit on the device at once (each loops over its share of the chunks)
    static std::atomic<long long> resident_cache[64] = {};   // device properties, not state
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    long long resident = dev >= 0 && dev < 64 ? resident_cache[dev].load(std::memory_order_relaxed) : 0;
    if (resident == 0) {
        int cus = 0, per_cu = 0;
        e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_entropy<false>, ENT_THREADS, 0);
        if (e != hipSuccess) return e;
        resident = (long long)cus * (per_cu > 0 ? per_cu : 1);
        if (dev >= 0 && dev < 64) resident_cache[dev].store(resident, std::memory_order_relaxed);
    }
    const unsigned grid = (unsigned)min((nchunks * nimg + ENT_WAVES - 1) / ENT_WAVES, resident);
    EntArgs a{coef, g.nblocks, nchunks, nimg, b.cstride, g.nb, g.h * g.v, {pred[0], pred[1], pred[2]}, b.epoch, b.clen,
              b.sparse, b.status};
    if (nimg > 1) hipLaunchKernelGGL(k_entropy<true>, dim3(grid), dim3(ENT_THREADS), 0, s, a);
    else hipLaunchKernelGGL(k_entropy<false>, dim3(grid), dim3(ENT_THREADS), 0, s, a);
    return hipGetLastError();
}

template <bool STUFF, int ST_THREADS, int DEPTH, bool TEAMS>
static hipError_t launch_stream_threads(const EntropyBufs &b, uint64_t bit_offset, uint8_t head_bits, int is_last,
                                        uint8_t *out, size_t out_stride, hipStream_t s) {
    const int nimg = b.nimg > 0 ? b.nimg : 1;
    const int ngroups = (int)((b.nchunks + 63) / 64);
    if (ngroups > MAX_GROUPS || !out || b.slot_words <= 0 || (STUFF && !b.sticket) || !b.gsum) return hipErrorInvalidValue;
    const size_t lds = (ngroups + 1) * sizeof(uint64_t);
    // As many workgroups as fit on an idle device at once (more would only queue up; fewer being resident,
    // because something else runs on the device, is harmless: segments are handed out in order by a ticket
    // counter).  The figure depends on the dynamic LDS size; it is looked up once per size class (the size
    // rounded up to 4 KiB).
    constexpr size_t LDS_CLASS = 4096;
    constexpr int NCLASS = (int)(((MAX_GROUPS + 1) * sizeof(uint64_t) + LDS_CLASS - 1) / LDS_CLASS) + 1;
    static std::atomic<long long> resident_cache[64][NCLASS] = {};   // device properties, not state
    const int cls = (int)((lds + LDS_CLASS - 1) / LDS_CLASS);
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    static std::atomic<int> cus_cache[64] = {};
    long long resident = dev >= 0 && dev < 64 ? resident_cache[dev][cls].load(std::memory_order_relaxed) : 0;
    int cus = dev >= 0 && dev < 64 ? cus_cache[dev].load(std::memory_order_relaxed) : 0;
    if (resident == 0 || cus == 0) {
        int per_cu = 0;
        e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (e == hipSuccess)
            e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_stream<STUFF, ST_THREADS, DEPTH, TEAMS>, ST_THREADS,
                                                             (size_t)cls * LDS_CLASS);
        if (e != hipSuccess) return e;
        if (cus < 1) cus = 1;
        resident = (long long)cus * (per_cu > 0 ? per_cu : 1);
        if (dev >= 0 && dev < 64) {
            resident_cache[dev][cls].store(resident, std::memory_order_relaxed);
            cus_cache[dev].store(cus, std::memory_order_relaxed);
        }
    }
    size_t cap = (size_t)resident;
    // A batch is stuffed while the coding kernel of the next batch runs: up to two workgroups per CU take nothing
    // from it; with more, the two kernels compete for the slots (7680x4320 batches: 865 K MP/s with two per CU,
    // 828 K with eight, 856 K with one for every second CU; one per CU: as two).
    if (ST_THREADS == ST_THREADS_BATCH && cap > 2 * (size_t)cus) cap = 2 * (size_t)cus;
    if (b.stuff_grid && b.stuff_grid < cap) cap = b.stuff_grid;
    // workgroups per image; at least one, however many images there are: more workgroups than are resident at once
    // only queue up (segments are handed out by ticket, so every wait is for a workgroup that is running or done)
    size_t wpi = cap / nimg > 0 ? cap / nimg : 1;
    if (wpi > b.nseg_cap) wpi = b.nseg_cap ? b.nseg_cap : 1;
    if (TEAMS && wpi > (size_t)MAX_TEAMS * TEAM) wpi = (size_t)MAX_TEAMS * TEAM;
    const size_t gstride = (size_t)(b.cstride / 64);
    const bool pre = nimg == 1;
    if (pre && !b.gsum_ready)
        hipLaunchKernelGGL(k_group_sums, dim3((unsigned)((ngroups + 15) / 16), 1u), dim3(1024), 0, s, b.clen, b.nchunks,
                           b.cstride, ngroups, b.gsum, gstride);
    StreamArgs a{b.sparse, b.clen, pre ? b.gsum : nullptr, gstride, b.nchunks, ngroups, (int)wpi, b.cstride, (size_t)b.slot_words, out_stride, b.nseg_cap, b.nseg_cap / 64 + 1,
                 b.nseg_cap / 4096 + 1, b.status, (uint32_t)(bit_offset & 7), head_bits, STUFF ? b.chain : nullptr, is_last, b.sagg, b.sgrp, b.shyp,
                 b.sticket, b.stuff_epoch, b.epoch, out, b.host_status, b.trace};
    hipLaunchKernelGGL((k_stream<STUFF, ST_THREADS, DEPTH, TEAMS>), dim3((unsigned)(wpi * nimg)), dim3(ST_THREADS), lds, s, a);
    return hipGetLastError();
}

template <bool STUFF>
static hipError_t launch_stream(const EntropyBufs &b, uint64_t bit_offset, uint8_t head_bits, int is_last, uint8_t *out,
                                size_t out_stride, hipStream_t s) {
    if (b.nimg > 1) return launch_stream_threads<STUFF, ST_THREADS_BATCH, 1, false>(b, bit_offset, head_bits, is_last, out, out_stride, s);
    return launch_stream_threads<STUFF, ST_THREADS_SINGLE, ST_DEPTH_SINGLE, STUFF>(b, bit_offset, head_bits, is_last, out, out_stride, s);
}

hipError_t launch_stuff(const EntropyBufs &b, uint64_t bit_offset, uint8_t head_bits, int is_last, hipStream_t s) {
    return launch_stream<true>(b, bit_offset, head_bits, is_last, b.scan, b.scan_stride, s);
}

hipError_t launch_dense(const EntropyBufs &b, hipStream_t s) {
    if ((b.nimg > 0 ? b.nimg : 1) != 1) return hipErrorInvalidValue;   // single images only
    return launch_stream<false>(b, 0, 0, 1, reinterpret_cast<uint8_t *>(b.bits), 0, s);
}

}  // namespace jpg
