// Internal declarations shared by the translation units of libjpeg_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace jpg {

// Geometry of one image (or one MCU-row stripe of an image).
//
// Coefficient storage ("scan layout"): 8x8 blocks in the order the entropy
// coder visits them -- MCUs row-major, inside an MCU the h*v luma blocks
// row-major, then Cb, then Cr (src/jpeg/entropy_coding.rs:74-124) -- 64 int16
// per block in zigzag order (src/jpeg/huffman_tables.rs:97-101), 128 bytes per
// block.  The reference's row-major planes (src/pixel_matrix/block_iterator.rs:
// 78-90) are a fixed permutation of this, produced on request by unscan().
struct Geom {
    int W, H;        // image size in pixels
    int h, v;        // chroma subsampling factors (src/jpeg/sampling.rs:6-22)
    int Wp, Hp;      // luma plane size, padded to the MCU (src/jpeg/jpeg_image.rs:36-49)
    int Wc, Hc;      // chroma plane size (src/jpeg/sampling.rs:24-44); == Wp/h, Hp/v here
    int mcux, mcuy;  // MCUs per row / column
    int nb;          // blocks per MCU = h*v + 2
    uint32_t stride; // bytes between BMP rows
    int bottom_up;   // BMP row order (src/bmp/bmp_image.rs:65-70)
    long long nblocks;
};

enum { ALGO_REAL = 0, ALGO_BIN = 1, ALGO_REAL_STRICT = 2 };

// Entropy-stage geometry: one thread codes one block; one wavefront codes a
// "chunk" of 64 consecutive blocks, whose bits it assembles in LDS.
constexpr int CHUNK_BLOCKS = 64;
// worst-case coded size of one block: DC 9+11 bits, 63 AC x (16+10) bits
constexpr int MAX_BLOCK_BITS = 20 + 63 * 26;
// words a block gets in a slot of the sparse buffer: room for a block of maximum size
constexpr int SLOT_WORDS_PER_BLOCK = 52;
// words of a chunk's slot in the sparse buffer
constexpr int CHUNK_STRIDE_WORDS = CHUNK_BLOCKS * SLOT_WORDS_PER_BLOCK;
static_assert(SLOT_WORDS_PER_BLOCK * 32 >= MAX_BLOCK_BITS + 1, "chunk slot too small");
// byte-stuffing works on segments of this many unstuffed bytes
constexpr int STUFF_SEG = 4096;
// words of the ticket counters of the stuffing kernel for a single image (a batch: two per image)
constexpr int TICKET_WORDS_SINGLE = 64 + 64 * 64;
// images one launch of the coding and stream kernels can take (jpg_max_batch): the pixel pointers of a launch
// travel in the coding kernel's argument block (2 KiB of its 4)
constexpr int MAX_BATCH_IMAGES = 256;
// groups of 64 chunks a stream reader can tabulate in LDS (52 KiB of LDS; limits an image to 27M blocks)
constexpr int MAX_GROUPS = 6656;

// results a host needs after an encode, fetched with one copy
struct DevStatus {
    uint64_t nbits;       // bits of the unstuffed stream
    uint64_t scan_bytes;  // bytes of the stuffed stream
    int32_t error;        // sticky jpg_status raised by a kernel (0 = none)
    uint32_t tail;        // bits of the dropped partial byte (stripes)
    uint32_t last32;      // the last min(32, nbits) bits of the unstuffed stream, right-aligned (stripes)
    uint32_t reserved;
};

// ---- stripe chain: one image cut into MCU-row stripes, one per GPU, whose kernels hand each other the few
// scalars that link neighbouring stripes (the three DC predictors of src/jpeg/entropy_coding.rs:38-40,149-150,
// the bit counts that fix where a stripe's bits start, the bits that share a byte, the stuffed lengths that fix
// where a piece lies) through a mailbox in page-locked host memory that every GPU of the node reads and writes
// directly: no host thread takes part between the first and the last kernel of a pass.
constexpr int CHAIN_MAX_RANKS = 64;
// Passes whose records exist side by side.  A rank publishes its DC values before it waits for anything, so with two
// passes of every rank in flight a fast rank may publish for pass f + 3 while the slowest still reads the values of
// pass f + 1 (rank 0 queues pass q only after the whole of pass q - 2 is in place, and no rank completes a pass rank 0
// has not begun): three passes must not share a record, four do not.
constexpr int CHAIN_SLOTS = 4;
// One record per (pass number mod CHAIN_SLOTS, rank), written only by kernels of that rank.  Every word is
// {tag:16 | value:48} and stored whole; tag = 0x8000 | (pass & 0x7fff), so a zeroed mailbox and the words of
// an earlier pass in the same record never look valid.
struct ChainMail {
    uint64_t dc;     // value = y:16 | cb:16 | cr:16, the DC coefficients of the stripe's last Y, Cb, Cr blocks
    uint64_t bits;   // bits of the stripe's unstuffed stream
    uint64_t tail;   // its last min(32, bits) bits, right-aligned
    uint64_t len;    // bytes of its stuffed piece
    uint64_t done;   // the piece lies in the gather buffer: value = error:1 (bit 47) | bytes:47
    uint64_t total;  // written by the last rank with a stripe when every piece lies there: error:1 (bit 47) | bytes of the scan:47
    uint64_t pad[2];
};
static_assert(sizeof(ChainMail) == 64, "one record per 64 bytes");
constexpr size_t CHAIN_MAILBOX_BYTES = CHAIN_SLOTS * CHAIN_MAX_RANKS * sizeof(ChainMail);
// What the gate kernels of a context leave for the kernels queued behind them (device memory of the context).
struct ChainParams {
    int16_t pred[4];        // DC predictors the stripe starts from
    uint32_t shift, head;   // bit offset of the stripe modulo 8; the bits before it in its first byte, top-aligned
    uint64_t bit_offset;    // of the stripe in the whole scan
    uint64_t byte_offset;   // of its stuffed piece in the gather buffer
    uint64_t len;           // bytes the copy kernel moves (0 after a failure)
    uint64_t total;         // last rank with a stripe: bytes of the whole scan
    uint32_t failed;        // a wait ran out, or the gather buffer is too small
    uint32_t pad;
};

// A "chunk" is the unit whose bits one kernel assembles on its own: 64 consecutive blocks
// (k_entropy) or the blocks of one tile of the fused encoder (k_coded).
struct EntropyBufs {
    int nimg;              // images of the current launch (a batch of equally sized images), >= 1
    long long nchunks;     // chunks of one image
    long long cstride;     // chunk slots per image in clen / sparse (nchunks rounded up to 64)
    int slot_words;        // words per chunk slot (blocks per chunk x SLOT_WORDS_PER_BLOCK)
    int small_window;      // launch_coded: the variant with the small bit window (where coded_small_window() has one)
    uint32_t *clen;        // [nimg][cstride] bits of each chunk
    uint32_t *sparse;      // [nimg][cstride][slot_words] coded bits of each chunk, MSB first in each word
    uint32_t epoch;        // tags the error word of the current entropy launch (host side, 1..2^22)
    DevStatus *status;       // [nimg]
    DevStatus *host_status;  // [nimg] pinned host copy the last kernel fills (may be null)
    uint32_t *bits;        // dense unstuffed MSB-first stream in memory byte order (staged API only, may be null)
    size_t bits_cap;       // bytes
    // running sums of the 0xFF counts of the stuffing segments (see entropy.hip); each word is
    // {epoch:14 | valid:1 | value:49} written by one 8-byte store
    uint32_t *gsum;        // [cstride / 64] bits of each group of 64 chunks of a single image (working array of the stream kernels)
    bool gsum_ready;       // ... as left by the tail kernel queued before the stream kernel
    uint64_t *sagg;        // [nimg][nseg_cap] 0xFF bytes of the segment
    uint64_t *sgrp;        // [nimg][nseg_cap/64 + 1] ... of a group of 64 segments
    uint64_t *shyp;        // [nimg][nseg_cap/4096 + 1] ... of a hyper-group of 64 groups
    uint32_t stuff_epoch;  // their generation (1..16383)
    uint32_t *sticket;     // [nimg][2], at least TICKET_WORDS_SINGLE words: ticket counters of the stuffing kernel; zero between launches
    size_t stuff_grid;     // workgroups of the stream kernels (0 = as many as fit on the device)
    uint8_t *scan;         // [nimg][scan_stride] stuffed streams
    size_t scan_stride;    // bytes per image
    size_t nseg_cap;       // stuffing segments an image can have
    unsigned long long *trace;  // debug builds only (JPG_TRACE): phase stamps of the stream kernel
    // stripe chain (all null / 0 otherwise)
    const ChainParams *chain;   // launch_coded takes the DC predictors, launch_stuff the shift and head bits from here
    ChainMail *chain_slot;      // launch_tail publishes the stripe's bit count and last bits here, tagged chain_tag
    uint32_t chain_tag;
};

// One pass of a stripe through the chain, as the launchers below need it
struct ChainPass {
    ChainMail *mail;            // [CHAIN_SLOTS][CHAIN_MAX_RANKS], host memory visible to every GPU
    ChainParams *params;        // this context's
    DevStatus *status;          // this context's status record of the pass (errors are raised here)
    uint32_t err_epoch;         // ... tagged with this
    uint32_t tag;               // 0x8000 | (pass & 0x7fff)
    int slot;                   // pass % CHAIN_SLOTS
    int rank;
    int pred_rank;              // last rank before this one with a stripe of its own, -1: none
    uint64_t active;            // bit k: rank k has a stripe
    uint64_t timeout_ticks;     // of the 100 MHz clock: how long a gate waits for another rank
};
// (the first kernel of a pass is launch_last_dc_pixels with the pass: it clears params->failed, publishes the DC
// coefficients of the stripe's last blocks and waits for the predecessor's -> params->pred)
// gate: wait for bit count and last bits of every earlier stripe -> params->shift / head / bit_offset
hipError_t launch_chain_place(const ChainPass &c, hipStream_t s);
// publishes the piece's length (status->scan_bytes), waits for the earlier pieces' -> params->byte_offset / len
hipError_t launch_chain_offset(const ChainPass &c, size_t gather_cap, hipStream_t s);
// copies params->len bytes of `scan` to gather + params->byte_offset (any alignment; peer or host memory)
hipError_t launch_chain_copy(const ChainPass &c, const uint8_t *scan, uint8_t *gather, hipStream_t s);
// publishes "piece in place"; the last rank with a stripe (collect) also waits for the same word of every rank before
// it and publishes the length of the whole scan in its record's `total`, with the error bit when any rank failed.
// (No kernel of the chain ever waits for a rank above its own: whoever needs the total -- the owner of the gather
// buffer -- reads that word from the host.)
hipError_t launch_chain_done(const ChainPass &c, bool collect, hipStream_t s);


// e0/e1: optional events that receive the kernel's own start and end time stamps.
// srcs/nimg: a batch of nimg equally sized images (device array of device pointers) in one launch;
// their blocks follow each other in `coef`.
hipError_t launch_front(const Geom &g, int algo, const uint8_t *src, int16_t *coef, hipStream_t s,
                        hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr, const uint8_t *const *srcs = nullptr,
                        int nimg = 1);
hipError_t launch_color(const Geom &g, const uint8_t *src, uint8_t *y, uint8_t *cb, uint8_t *cr, hipStream_t s);
hipError_t launch_downsample(const Geom &g, const uint8_t *cb, const uint8_t *cr, uint8_t *cb_out, uint8_t *cr_out,
                             hipStream_t s);
hipError_t launch_dct_planes(const Geom &g, int algo, const uint8_t *y, const uint8_t *cb, const uint8_t *cr,
                             int16_t *coef, hipStream_t s);
hipError_t launch_unscan(const Geom &g, const int16_t *coef, int comp, int16_t *plane, hipStream_t s);
hipError_t launch_last_dc(const Geom &g, const int16_t *coef, int16_t *dc3, hipStream_t s);

// pred: the 3 DC predictors (Y, Cb, Cr) the first blocks start from
hipError_t launch_entropy(const Geom &g, const int16_t *coef, const int16_t pred[3], const EntropyBufs &b,
                          hipStream_t s);
// The fused encoder: pixels to coded tiles (chunks = tiles; b describes coded_tiles(g) slots of
// coded_slot_words(g) words).  Only for geometries coded_supported() accepts.
bool coded_supported(const Geom &g);
long long coded_tiles(const Geom &g);
// Is there a variant of the coding kernel with a small bit window (a fifth workgroup per CU: faster while the tiles
// fit the window, slower on content whose tiles are much longer), and how many bits does that window hold?
bool coded_small_window(const Geom &g, int algo);
int coded_small_window_bits();
int coded_slot_words(const Geom &g);
// mcu_row0 / mcu_rows: the band of MCU rows this launch codes (default: all); the bands of an image are
// launched in order on one stream, each reading the pixel rows of its own MCU rows and of the one above.
hipError_t launch_coded(const Geom &g, int algo, const uint8_t *src, const uint8_t *const *srcs, int nimg,
                        const int16_t pred[3], const EntropyBufs &b, hipStream_t s, hipEvent_t e0 = nullptr,
                        hipEvent_t e1 = nullptr, int mcu_row0 = 0, int mcu_rows = -1);
// the DC coefficients of the last Y, Cb, Cr blocks of the image, computed from its pixels
// chain: the launch is the first kernel of a pass of the stripe chain: the values are also published for the stripe
// after this one and those of the stripe before it are waited for and left in chain->params->pred
hipError_t launch_last_dc_pixels(const Geom &g, int algo, const uint8_t *src, int16_t *dc3, hipStream_t s,
                                 const ChainPass *chain = nullptr);
hipError_t launch_stuff(const EntropyBufs &b, uint64_t bit_offset, uint8_t head_bits, int is_last, hipStream_t s);
// writes the dense stream to b.bits
hipError_t launch_dense(const EntropyBufs &b, hipStream_t s);
// sums the chunk lengths and picks the last 32 bits of the stream out of the last chunks: status->nbits,
// status->last32 (and the host copy), without assembling the stream (single images)
hipError_t launch_tail(const EntropyBufs &b, hipStream_t s);

// JpegImage::new's dimension arithmetic (src/jpeg/jpeg_image.rs:36-56) for chroma factors (h, v): fills g and
// returns a jpg_status (bad sizes, or the sizes at which the reference skews its chroma plane)
int make_geom(Geom &g, int32_t W, int32_t H, uint32_t stride, int bottom_up, int h, int v);
// (4,4,4) -> (1,1), (4,2,0) -> (2,2), (4,2,2) -> (2,1) (src/jpeg/sampling.rs:6-22); false for any other ratio
bool ratio_factors(int ra, int rb, int rc, int &h, int &v);

// worst case of the unstuffed stream in bytes: every block at its maximum coded size
inline size_t worst_bits_bytes(size_t nblk) {
    const size_t b = (nblk * MAX_BLOCK_BITS + 7) / 8 + 64;
    return (b + 15) / 16 * 16;
}
// bytes a stuffed scan can take at worst (every byte 0xFF), rounded up to 256
inline size_t worst_scan_bytes(size_t nblk) { return (2 * worst_bits_bytes(nblk) + 16 + 255) / 256 * 256; }
// stuffing segments an image of nblk blocks can have
inline size_t worst_segments(size_t nblk) { return worst_bits_bytes(nblk) / STUFF_SEG + 2; }

}  // namespace jpg
