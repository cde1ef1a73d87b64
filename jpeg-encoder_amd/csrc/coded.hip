// Fused encoder: BMP rows -> YCbCr -> chroma box average -> 8x8 DCT ->
// quantisation -> zigzag -> Huffman coding, one kernel; neither planes nor
// coefficients are ever written to memory.  Replaces, for whole encodes, the
// loop of JpegImage::from_bmp (src/jpeg/jpeg_image.rs:121-134),
// chrominance_downsampling (src/jpeg/sampling.rs:46-102), dct_and_quantization
// (src/jpeg/dct_quant.rs:15-61) and generate_entropy_encoded_data
// (src/jpeg/entropy_coding.rs:16-125) of the reference.
//
// One workgroup owns a tile of 512 (4:2:2: 1024) pixels x one MCU row, converts and transforms
// it (tile.hpp) and then codes the tile's blocks, which are consecutive in the
// scan: one thread per block measures its block, a scan over the workgroup
// places it, and the bits are assembled in LDS and stored to the tile's own
// fixed-size slot together with the tile's length.  The stream kernels of
// entropy.hip concatenate the slots.
//
// The only value a tile needs from outside is the DC of the last block of each
// component before it (the three predictor chains of
// src/jpeg/entropy_coding.rs:38-40,149-150).  Instead of waiting for the tile
// before it, the workgroup converts that one MCU again and computes the three DC
// coefficients (the fourth wave does it; except at 4:2:2 it has no block of the
// tile to transform), so that no workgroup depends on another.
#include <hip/hip_ext.h>

#include "chain.hpp"
#include "coding.hpp"
#include "tile.hpp"

namespace jpg {

// LDS words a tile's bits are assembled in per pass: 1536 (48 Kbit; tiles of photo-like content hold about 4 Kbit,
// noise about 45 Kbit).  (2048 words is 0.5 % slower than 1536: four workgroups then leave 10 KB of a CU's LDS free
// instead of 18, room for one instead of two workgroups of the stuffing kernel that runs beside this one.)
// At tiles of 1024 pixels (4:2:2, 256 blocks of 144 bytes in LDS) only a window of 256 words leaves room for a
// fourth workgroup per CU; their tiles hold about 4000 bits on photo-like content, one pass as a rule.
// SMALL: 512 words, with which a workgroup of 512-pixel tiles fits five times into a CU's 160 KB of LDS.  The f32
// kernels need 100 registers (four waves per SIMD) and gain nothing from it (1 % slower on photo-like content, 16 %
// on noise).  The integer transform runs in 82 registers since it was rewritten in f32 lifting steps, and a fifth
// workgroup per CU makes 3840x2160 4:4:4 10 % and 7680x4320 4:2:0 4 % faster on photo-like content -- and noise,
// whose tiles then take five passes instead of two, 28 % slower.  The caller chooses per launch
// (EntropyBufs::small_window) from the tile lengths of the encodes before.
template <int HS, int VS, bool SMALL>
constexpr int win_words() {
    return TileCfg<HS, VS>::TW == 1024 ? 256 : (SMALL ? 512 : 1536);
}
constexpr int SMALL_WINDOW_BITS = 512 * 32;

constexpr int MAX_LAUNCH_IMAGES = MAX_BATCH_IMAGES;   // images of one launch: their pixel pointers travel in the argument block

struct CodedArgs {
    // Pixel array of each image of the launch.  In the argument block rather than in memory: the pointer is
    // then one scalar load away at the start of the kernel instead of two dependent ones.
    const uint8_t *src[MAX_LAUNCH_IMAGES];
    Geom g;
    int mcu_row0;                 // first MCU row of this launch (0 unless the image is coded band by band)
    int16_t pred[3];              // DC predictors the first MCU of every image starts from
    const int16_t *pred_dev;      // ... read from here instead when not null (a stripe whose predictors a kernel queued before this one fetched)
    uint32_t epoch;               // tags errors (see raise_error)
    long long cstride;            // tile slots per image in clen / sparse
    int slot_words;               // words per tile slot
    uint32_t *clen;               // [nimg][cstride] out: bits of each tile
    uint32_t *sparse;             // [nimg][cstride][slot_words] out: the tile's bits, MSB first in each word
    DevStatus *status;            // [nimg]
#ifdef JPG_PHASE_STAMPS
    unsigned long long *stamps;   // [nimg][cstride][4 waves][STAMP_WORDS] out: clock readings of the instrumented build
#endif
};

#ifdef JPG_PHASE_STAMPS
// Instrumented build (tools/phase_stamps.py): every wave reads the shader clock at the phase boundaries and leaves
// the readings, with the id of the SIMD it ran on, in the buffer given to jpg_debug_phase_stamps.
constexpr int STAMP_WORDS = 10;
static unsigned long long *g_stamps = nullptr;
#define STAMP(i) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp[i]) : : "memory")
#else
#define STAMP(i)
#endif

template <int HS, int VS, int ALGO, bool SMALL>
__device__ __forceinline__ void coded_tile(const CodedArgs &a) {
    using C = TileCfg<HS, VS>;
    constexpr int HV = HS * VS;
    static_assert(C::BLOCKS % 64 == 0 && C::BLOCKS <= THREADS, "one thread per block");
    // Y, Cb, Cr planes of the tile; then its finished blocks
    __shared__ __attribute__((aligned(16))) uint8_t lds[C::LDS_BYTES];
    __shared__ __attribute__((aligned(16))) uint8_t sHalo[C::HALO_BYTES];   // planes of the MCU before the tile
    __shared__ int16_t sHaloDc[4];                                          // its last Y, Cb, Cr DC coefficients
    __shared__ uint32_t sLut[LUT_WORDS];
    constexpr int WIN_WORDS = win_words<HS, VS, SMALL>();
    __shared__ __attribute__((aligned(16))) uint32_t sWin[WIN_WORDS + 4];   // (a whole number of 16-byte pieces, for the clearing)
    __shared__ uint32_t sTot[THREADS / 64], sPass[THREADS / 64];
    uint8_t *sY = lds;
    uint8_t *sCb = lds + C::TILE_H * C::TW;
    uint8_t *sCr = sCb + 8 * C::CW;
#ifdef JPG_PHASE_STAMPS
    unsigned long long stamp[8];
    STAMP(0);
#endif
    // Every kernel argument the workgroup will use is fetched here, in one batch of scalar loads with one wait.  Left to
    // the compiler each is loaded where it is first used, and the start of a workgroup becomes a chain of scalar-load
    // round trips (kernel arguments, then the image's pixel pointer, then the geometry in two pieces: 1500 cycles from
    // the start of a wave to its first pixel load, of which this takes out a third), with more of them behind the
    // barriers of the coding phase.  The empty statements pin the values to registers at this point.
    Geom g = a.g;
    int k_row0 = a.mcu_row0, k_slot_words = a.slot_words;
    uint32_t k_epoch = a.epoch, k_cstride = (uint32_t)a.cstride;
    uintptr_t k_src = reinterpret_cast<uintptr_t>(a.src[blockIdx.z]), k_clen = reinterpret_cast<uintptr_t>(a.clen),
              k_sparse = reinterpret_cast<uintptr_t>(a.sparse);
    // (Not at tiles of 1024 pixels: 1920x1080 4:2:2 batches measure 2.7 % slower with it, 8K 4:2:0 and 4:4:4 0.5-0.7 % faster.)
    if constexpr (TileCfg<HS, VS>::TW != 1024) {
        asm volatile("" : "+s"(g.W), "+s"(g.H), "+s"(g.mcux), "+s"(g.stride), "+s"(g.bottom_up), "+s"(k_row0), "+s"(k_src));
        asm volatile("" : "+s"(k_slot_words), "+s"(k_epoch), "+s"(k_cstride), "+s"(k_clen), "+s"(k_sparse));
    }
#if defined(JPG_PHASE_STAMPS) && defined(JPG_STAMP_AFTER_ARGS)   // the first reading is taken when the arguments have arrived
    STAMP(0);
#endif
    typedef __attribute__((address_space(1))) uint32_t *global_words;
    const global_words clen_out = (global_words)k_clen, sparse_out = (global_words)k_sparse;
    // grid: x = tile column, y = MCU row (counted from the first row of the launch: a launch may cover a band of
    // rows, the rows above it being coded by earlier launches), z = image of a batch; no divisions to find the tile
    const int tile_x = blockIdx.x, tile_y = blockIdx.y + k_row0, img = blockIdx.z;
    const int tile = tile_y * (int)gridDim.x + tile_x;       // number of the tile in the scan of its image
    const uint8_t *src = reinterpret_cast<const uint8_t *>(k_src);
    const int mcu0 = tile_x * C::MCUS;                        // first MCU column of the tile
    const int mcus = min(C::MCUS, g.mcux - mcu0);             // MCUs really present
    const int x0 = mcu0 * 8 * HS;
    const int y0 = tile_y * C::TILE_H;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // the MCU before the tile in the scan: the last one of the row above for the first tile of a row
    const bool has_halo = tile > 0;
    const int hmx = mcu0 ? mcu0 - 1 : g.mcux - 1, hmy = mcu0 ? tile_y : tile_y - 1;

    // The Huffman table travels to LDS through registers: its loads are issued here, ahead of the pixel loads,
    // and stored after the conversion, so that nothing waits for them.
    constexpr int LUT_ROUNDS = (LUT_WORDS + THREADS - 1) / THREADS;
    uint32_t lut_words[LUT_ROUNDS];
#pragma unroll
    for (int k = 0; k < LUT_ROUNDS; k++) lut_words[k] = (&HUFF_DEV.e[0][0])[min(tid + k * THREADS, LUT_WORDS - 1)];

#ifdef JPG_PHASE_STAMPS
    stamp[1] = stamp[0];
    // (JPG_STAMP_LAST: the reading is taken when the last pixels have arrived instead of the first;
    // JPG_STAMP_ISSUED: when the loads have been issued)
    tile_convert<HS, VS>(src, g, x0, y0, mcus, sY, sCb, sCr, tid, [&](uint32_t first, uint32_t last) {
#ifdef JPG_STAMP_LAST
        first = last;
#endif
#ifdef JPG_STAMP_ISSUED   // the reading is taken when the loads have been issued, whether or not anything has arrived
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp[1]) : : "memory");
        return;
#endif
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp[1]) : "v"(first) : "memory");
    });
    STAMP(2);
#else
    tile_convert<HS, VS>(src, g, x0, y0, mcus, sY, sCb, sCr, tid);
#endif
#pragma unroll
    for (int k = 0; k < LUT_ROUNDS; k++)
        if ((k + 1) * THREADS <= LUT_WORDS || tid + k * THREADS < LUT_WORDS) sLut[tid + k * THREADS] = lut_words[k];
    constexpr int WIN_PIECES = (WIN_WORDS + 4) / 4;                    // (cleared after the loads, like the table)
#pragma unroll
    for (int r = 0; r < (WIN_PIECES + THREADS - 1) / THREADS; r++)
        if ((r + 1) * THREADS <= WIN_PIECES || tid + r * THREADS < WIN_PIECES)
            reinterpret_cast<uint4 *>(sWin)[tid + r * THREADS] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    STAMP(3);
#if JPG_ABLATE >= 2   // (timing experiments: the kernel without its later phases; see tools/build_variant.py)
    if (tid == 0) clen_out[(uint32_t)img * k_cstride + (uint32_t)tile] = sY[5] & 1;
    return;
#endif
    tile_transform<HS, VS, ALGO, (C::TW == 512)>(lds, mcus, tid, [&] {
        if (!has_halo) {
            // first tile of the image: the caller's predictors, left where the coders look for the MCU before the tile
            if (lane < 3) sHaloDc[lane] = a.pred_dev ? a.pred_dev[lane] : (lane == 0 ? a.pred[0] : (lane == 1 ? a.pred[1] : a.pred[2]));
            return;
        }
        HaloRegs<VS> halo;
        halo_load<HS, VS>(src, g, hmx * 8 * HS, hmy * 8 * VS, lane, halo);
        halo_convert<HS, VS>(src, g, hmx * 8 * HS, hmy * 8 * VS, sHalo, sHalo + C::HALO_Y, sHalo + C::HALO_Y + 64, lane, halo);
        __builtin_amdgcn_wave_barrier();   // (the LDS accesses of one wave execute in order)
        if constexpr (ALGO == ALGO_REAL) halo_dc_sums<HS, VS>(sHalo, sHalo + C::HALO_Y, sHalo + C::HALO_Y + 64, lane, sHaloDc);
        else halo_dc<HS, VS, ALGO>(sHalo, sHalo + C::HALO_Y, sHalo + C::HALO_Y + 64, lane, sHaloDc);
    });
    STAMP(4);
    __syncthreads();
    STAMP(5);
#if JPG_ABLATE == 1
    if (tid == 0) clen_out[(uint32_t)img * k_cstride + (uint32_t)tile] = lds[5] & 1;
    return;
#endif

    // ---------------- coding: thread `tid` codes block `tid` of the tile ----------------
    const int nblk = mcus * C::NBLK;
    const bool valid = tid < nblk;
    DevStatus *status = a.status + img;
    const uint16_t *slot = reinterpret_cast<const uint16_t *>(&lds[(valid ? tid : 0) * OUT_STRIDE]);
    uint32_t mlo = 0, mhi = 0, len = 0;
    int dc_diff = 0;
    const uint32_t *lut = sLut;
    AccWriter acc;                                            // blocks of up to ACC_BITS bits are kept whole
    if (valid) {
        uint4 q[8];
#pragma unroll
        for (int i = 0; i < 8; i++) q[i] = reinterpret_cast<const uint4 *>(slot)[i];
        nonzero_mask(q, mlo, mhi);
        mlo &= ~1u;                                           // the DC symbol is coded separately
        // DC difference against the previous block of the same component
        const int jb = tid % C::NBLK;
        const int comp = jb < HV ? 0 : jb - HV + 1;
        const int back = prev_distance(jb, C::NBLK, HV);
        // (one LDS read at a selected address: the block before it in the tile, or the MCU before the tile)
        const int16_t *prev = tid >= back ? reinterpret_cast<const int16_t *>(&lds[(tid - back) * OUT_STRIDE]) : &sHaloDc[comp];
        const int pdc = *prev;
        dc_diff = (int16_t)((int)(int16_t)(q[0].x & 0xffff) - pdc);
        lut = sLut + (comp ? jpgtab::HUFF_LUT_STRIDE : 0);
        len = code_block(slot, mlo, mhi, dc_diff, lut, status, k_epoch, acc);
    }
    STAMP(6);
    const uint32_t sc = wave_incl_scan(len);
    if (lane == 63) sTot[wave] = sc;
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < THREADS / 64; w++) {
        const uint32_t t = sTot[w];
        if (w < wave) before += t;
        total += t;
    }
    const uint32_t end = before + sc;                         // bits of the tile up to and including this block
    const uint32_t slot_index = (uint32_t)img * k_cstride + (uint32_t)tile;   // (the launcher checks the range)
    if (tid == 0) clen_out[slot_index] = total;

    // A tile too long for the LDS window goes in several passes, each starting in the word the
    // previous one ended in (that word is carried over and stored again).
    const global_words dst = sparse_out + (uint64_t)slot_index * (uint32_t)k_slot_words;
    constexpr uint32_t WINDOW = WIN_WORDS * 32;               // bits per pass
    uint32_t done = 0;                                        // bits of the tile already stored
    for (;;) {
        const uint32_t start = done & 31;
        // blocks that end within the window (at least one: a block is shorter than WINDOW - 31 bits)
        const bool mine = valid && end > done && end - done + start <= WINDOW;
        uint32_t pass_bits;
        if (total - done + start <= WINDOW) {
            pass_bits = total - done;
        } else {
            const uint32_t cand = __shfl(mine ? end - done : 0u, 63 - __clzll(__ballot(mine) | 1ull));
            if (lane == 0) sPass[wave] = cand;
            __syncthreads();
            pass_bits = 0;
#pragma unroll
            for (int w = 0; w < THREADS / 64; w++) pass_bits = max(pass_bits, sPass[w]);
        }
        const uint32_t nwords = (start + pass_bits + 31) >> 5;
        if (mine) {
            const uint32_t pos = end - len - done + start;
            if (len <= ACC_BITS) {
                acc.emit(sWin, pos, len);
            } else {                                          // long block: walk it again
                BitWriter w;
                w.init(sWin, pos);
                code_block(slot, mlo, mhi, dc_diff, lut, status, k_epoch, w);
            }
        }
        __syncthreads();
        const uint32_t endb = start + pass_bits;
        const uint32_t carry = (endb & 31) ? sWin[endb >> 5] : 0u;
#pragma unroll 1   // (a tile of photo-like content is one round; unrolled, the loop's set-up costs more than its body)
        for (uint32_t i = tid; i < nwords; i += THREADS) dst[(done >> 5) + i] = sWin[i];
        done += pass_bits;
        if (done >= total) break;
        __syncthreads();
#pragma unroll 1
        for (uint32_t i = tid; i < nwords + 1; i += THREADS) sWin[i] = i ? 0u : carry;
        __syncthreads();
    }
#ifdef JPG_PHASE_STAMPS
    STAMP(7);
    if (a.stamps && lane == 0) {
        unsigned long long *o = a.stamps + (slot_index * (THREADS / 64) + wave) * STAMP_WORDS;
#pragma unroll
        for (int i = 0; i < 8; i++) o[i] = stamp[i];
        uint32_t hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        o[8] = hw;
        o[9] = xcc;
    }
#endif
}

#ifdef JPG_PHASE_STAMPS
extern "C" void jpg_debug_phase_stamps(void *dev_buffer) { g_stamps = static_cast<unsigned long long *>(dev_buffer); }
#endif

template <int HS, int VS, int ALGO, bool SMALL>
__global__ __launch_bounds__(THREADS) void k_coded(CodedArgs a) {
    coded_tile<HS, VS, ALGO, SMALL>(a);
}

// The same kernel held to the 128 registers that let a fourth wave share a SIMD: the f32 transform at tiles of
// 1024 pixels (4:2:2) wants a few more.
template <int HS, int VS, int ALGO>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_coded_4waves(CodedArgs a) {
    coded_tile<HS, VS, ALGO, false>(a);
}

// The DC coefficients of the last Y, Cb, Cr blocks of MCU (mx, my), straight from the pixels: what a
// following stripe starts its predictors from (src/jpeg/entropy_coding.rs:149-150).  One wave.
// chain.mail != null: the first kernel of a pass of the stripe chain; the wave also publishes the three values for the
// stripe after this one and waits for those of the stripe before it (chain_exchange_dc).
template <int HS, int VS, int ALGO>
__global__ __launch_bounds__(64) void k_mcu_dc(const uint8_t *__restrict__ src, Geom g, int mx, int my,
                                               int16_t *__restrict__ dc3, ChainPass chain) {
    using C = TileCfg<HS, VS>;
    __shared__ __attribute__((aligned(16))) uint8_t sHalo[C::HALO_BYTES];
    __shared__ int16_t sDc[4];
    const int lane = threadIdx.x;
    HaloRegs<VS> regs;
    halo_load<HS, VS>(src, g, mx * 8 * HS, my * 8 * VS, lane, regs);
    halo_convert<HS, VS>(src, g, mx * 8 * HS, my * 8 * VS, sHalo, sHalo + C::HALO_Y, sHalo + C::HALO_Y + 64, lane, regs);
    __builtin_amdgcn_wave_barrier();   // (the LDS accesses of one wave execute in order)
    if constexpr (ALGO == ALGO_REAL) halo_dc_sums<HS, VS>(sHalo, sHalo + C::HALO_Y, sHalo + C::HALO_Y + 64, lane, sDc);
    else halo_dc<HS, VS, ALGO>(sHalo, sHalo + C::HALO_Y, sHalo + C::HALO_Y + 64, lane, sDc);
    __builtin_amdgcn_wave_barrier();
    if (lane < 3) dc3[lane] = sDc[lane];
    if (chain.mail) chain_exchange_dc(chain, sDc, lane);
}

template <int HS, int VS>
static hipError_t launch_mcu_dc_hv(const Geom &g, int algo, const uint8_t *src, int16_t *dc3, const ChainPass &cp, hipStream_t s) {
    const int mx = g.mcux - 1, my = g.mcuy - 1;
    switch (algo) {
        case ALGO_REAL: hipLaunchKernelGGL((k_mcu_dc<HS, VS, ALGO_REAL>), dim3(1), dim3(64), 0, s, src, g, mx, my, dc3, cp); break;
        case ALGO_BIN: hipLaunchKernelGGL((k_mcu_dc<HS, VS, ALGO_BIN>), dim3(1), dim3(64), 0, s, src, g, mx, my, dc3, cp); break;
        case ALGO_REAL_STRICT: hipLaunchKernelGGL((k_mcu_dc<HS, VS, ALGO_REAL_STRICT>), dim3(1), dim3(64), 0, s, src, g, mx, my, dc3, cp); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_last_dc_pixels(const Geom &g, int algo, const uint8_t *src, int16_t *dc3, hipStream_t s, const ChainPass *chain) {
    if ((reinterpret_cast<uintptr_t>(src) & 3) || (g.stride & 3)) return hipErrorInvalidValue;
    const ChainPass cp = chain ? *chain : ChainPass{};
    if (g.h == 2 && g.v == 2) return launch_mcu_dc_hv<2, 2>(g, algo, src, dc3, cp, s);
    if (g.h == 2 && g.v == 1) return launch_mcu_dc_hv<2, 1>(g, algo, src, dc3, cp, s);
    if (g.h == 1 && g.v == 1) return launch_mcu_dc_hv<1, 1>(g, algo, src, dc3, cp, s);
    return hipErrorInvalidValue;
}

template <int HS, int VS>
static hipError_t launch_coded_hv(const CodedArgs &a, int algo, int nimg, int mcu_rows, bool small_window, hipStream_t s,
                                  hipEvent_t e0, hipEvent_t e1) {
    using C = TileCfg<HS, VS>;
    const int tiles_x = (a.g.mcux + C::MCUS - 1) / C::MCUS;
    const long long tiles = (long long)tiles_x * a.g.mcuy;
    if (tiles <= 0 || tiles * nimg > 0x7fffffffLL || tiles > a.cstride || a.slot_words != coded_slot_words(a.g))
        return hipErrorInvalidValue;
    if (a.g.mcuy > 65535 || nimg > 65535 || a.cstride * nimg > 0xffffffffLL) return hipErrorInvalidValue;
    if (mcu_rows < 1 || a.mcu_row0 < 0 || a.mcu_row0 + mcu_rows > a.g.mcuy) return hipErrorInvalidValue;
    const dim3 grid((unsigned)tiles_x, (unsigned)mcu_rows, (unsigned)nimg), block(THREADS);
    switch (algo) {
        case ALGO_REAL:
            if constexpr (TileCfg<HS, VS>::TW == 1024) hipExtLaunchKernelGGL((k_coded_4waves<HS, VS, ALGO_REAL>), grid, block, 0, s, e0, e1, 0, a);
            else hipExtLaunchKernelGGL((k_coded<HS, VS, ALGO_REAL, false>), grid, block, 0, s, e0, e1, 0, a);
            break;
        case ALGO_BIN:
            if constexpr (TileCfg<HS, VS>::TW == 512) {
                if (small_window) {
                    hipExtLaunchKernelGGL((k_coded<HS, VS, ALGO_BIN, true>), grid, block, 0, s, e0, e1, 0, a);
                    break;
                }
            }
            hipExtLaunchKernelGGL((k_coded<HS, VS, ALGO_BIN, false>), grid, block, 0, s, e0, e1, 0, a);
            break;
        case ALGO_REAL_STRICT: hipExtLaunchKernelGGL((k_coded<HS, VS, ALGO_REAL_STRICT, false>), grid, block, 0, s, e0, e1, 0, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

static int tile_mcus(const Geom &g) { return tile_width(g.h, g.v) / (8 * g.h); }

bool coded_supported(const Geom &g) {
    // Every row of tiles must begin with a full tile: the stream kernels tabulate a bounded number of
    // slots per segment of the stream, which holds when no two consecutive slots are both nearly empty.
    return g.mcux >= tile_mcus(g);
}

// the coding kernel of this geometry and transform exists with the small bit window too
bool coded_small_window(const Geom &g, int algo) { return algo == ALGO_BIN && tile_width(g.h, g.v) == 512; }
int coded_small_window_bits() { return SMALL_WINDOW_BITS; }

long long coded_tiles(const Geom &g) { return (long long)((g.mcux + tile_mcus(g) - 1) / tile_mcus(g)) * g.mcuy; }

int coded_slot_words(const Geom &g) { return tile_mcus(g) * g.nb * SLOT_WORDS_PER_BLOCK; }

// srcs: host array of the nimg device pointers of a batch, or null for the single image `src`
hipError_t launch_coded(const Geom &g, int algo, const uint8_t *src, const uint8_t *const *srcs, int nimg,
                        const int16_t pred[3], const EntropyBufs &b, hipStream_t s, hipEvent_t e0, hipEvent_t e1,
                        int mcu_row0, int mcu_rows) {
    if (mcu_rows < 0) mcu_rows = g.mcuy - mcu_row0;
    // rows are read as dwords (the caller checks the pointers of a batch)
    if (!srcs) nimg = 1;
    if ((g.stride & 3) || nimg < 1 || nimg > MAX_LAUNCH_IMAGES || !coded_supported(g)) return hipErrorInvalidValue;
    if (nimg != (b.nimg > 0 ? b.nimg : 1) || b.nchunks != coded_tiles(g)) return hipErrorInvalidValue;
    CodedArgs a{{}, g, mcu_row0, {pred[0], pred[1], pred[2]}, b.chain ? b.chain->pred : nullptr, b.epoch, b.cstride, b.slot_words,
                b.clen, b.sparse, b.status};
#ifdef JPG_PHASE_STAMPS
    a.stamps = g_stamps;
#endif
    for (int i = 0; i < nimg; i++) {
        a.src[i] = srcs ? srcs[i] : src;
        if (!a.src[i] || (reinterpret_cast<uintptr_t>(a.src[i]) & 3)) return hipErrorInvalidValue;   // rows are read as dwords
    }
    if (g.h == 2 && g.v == 2) return launch_coded_hv<2, 2>(a, algo, nimg, mcu_rows, b.small_window != 0, s, e0, e1);
    if (g.h == 2 && g.v == 1) return launch_coded_hv<2, 1>(a, algo, nimg, mcu_rows, b.small_window != 0, s, e0, e1);
    if (g.h == 1 && g.v == 1) return launch_coded_hv<1, 1>(a, algo, nimg, mcu_rows, b.small_window != 0, s, e0, e1);
    return hipErrorInvalidValue;
}

}  // namespace jpg
