// C ABI of libjpeg_hip.so (declared in include/jpeg_hip.h): context and buffer
// management around the kernels of front.hip, staged.hip and entropy.hip.
#include "../../include/jpeg_hip.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>

#include "internal.hpp"

using namespace jpg;

enum Stage { ST_NONE = 0, ST_PLANES = 1, ST_DOWN = 2, ST_COEF = 3, ST_BITS = 4, ST_SCAN = 5 };

struct jpg_ctx {
    int device = 0;
    int algo = 0;
    int h = 1, v = 1;
    hipStream_t stream = nullptr;
    int last_hip = 0;
    int stage = ST_NONE;
    bool have_planes = false;   // staged u8 planes are valid
    bool down_done = false;
    Geom g{};

    uint8_t *d_src_own = nullptr; size_t src_cap = 0;
    const uint8_t *d_src = nullptr;
    uint8_t *d_plane[3] = {nullptr, nullptr, nullptr}; size_t plane_cap = 0;   // Wp x Hp each
    uint8_t *d_down[2] = {nullptr, nullptr}; size_t down_cap = 0;              // Wc x Hc each
    int16_t *d_coef = nullptr; size_t coef_cap = 0;                            // in blocks
    bool coef_valid = false;    // d_coef holds the coefficients of the current image
    bool two_kernels = false;   // JPG_TWO_KERNELS=1: whole encodes never use the fused coding kernel
    // The coding kernel's bit window (coded_small_window): 0 chosen per launch from the tile lengths of the encode
    // finished last, 1 always small, 2 always large (JPG_CODED_WINDOW=small|large)
    int window_mode = 0;
    double tile_bits = -1.0;    // mean coded bits per tile of the encode finished last (-1: none yet)
    int bands_override = 0;     // JPG_BANDS: bands a host-row encode is cut into (0: chosen per image)
    int16_t *d_tmp16 = nullptr; size_t tmp16_cap = 0;
    int16_t *d_dc = nullptr;     // last DCs out (3 values)
    static constexpr int NSLOT = 3;
    // Sets of back-end buffers: while the stuffing kernel of one image runs on stream_b, the
    // front and entropy kernels of the next image run on `stream` into the other set.
    struct Slot {
        EntropyBufs eb{};
        const uint8_t **d_srcs = nullptr; size_t cap_srcs = 0;   // device array of a batch's pixel pointers
        size_t cap_clen = 0, cap_sparse = 0, cap_scan = 0, cap_status = 0, cap_sagg = 0, cap_sgrp = 0, cap_shyp = 0, cap_gsum = 0;
        DevStatus *h_status = nullptr;  // [cap_status] pinned copy of eb.status
        DevStatus *h_status_alt = nullptr;   // not owned: where the kernels of the launches being queued put the copy instead
        hipEvent_t e_done = nullptr;    // entropy kernel finished (recorded on stream)
        hipEvent_t s_done = nullptr;    // stuffing kernel finished (recorded on the stream it ran on)
        bool busy = false;              // s_done has been recorded at least once
        bool dense_valid = false;       // eb.bits holds the dense stream of the last entropy launch
        bool tiles = false;             // the chunks are tiles of the fused coding kernel
        bool tail_known = false;        // h_status->last32 is that of the current chunks (k_tail has run on them)
        size_t nblocks = 0;             // blocks of the image the slot's chunks were coded from (sizes the dense stream)
        size_t cap_ticket = 0;
        uint64_t nbits = 0;
        size_t scan_len = 0;
    } slot[NSLOT];
    int nimg = 1;                // images of the current geometry (a batch), set with the geometry
    int cur = 0;                 // slot of the most recent entropy launch
    int view = 0;                // slot whose results the getters show
    int fifo[NSLOT] = {};        // slots of queued fused encodes that jpg_finish_next has not handed out yet
    int npend = 0;
    hipStream_t stream_b = nullptr;
    // whole encodes of host rows go band by band: the rows of a band travel on stream_c while the bands before it
    // are coded on `stream`
    hipStream_t stream_c = nullptr;
    static constexpr int MAX_BANDS = 16;
    hipEvent_t band_ev[MAX_BANDS] = {};
    hipEvent_t src_free = nullptr;   // everything queued on `stream` that reads d_src_own has finished
    size_t last_scan_len = 0;        // size of the previous stuffed scan: the first guess for the next output buffer
    uint8_t *scan_host = nullptr;    // output buffer of a begun jpg_encode_scan (page-locked), handed to the caller by _end
    size_t scan_host_cap = 0;
    bool scan_begun = false;
    // two page-locked bounce buffers for rows that arrive in pageable host memory
    uint8_t *h_bounce[2] = {nullptr, nullptr};
    hipEvent_t bounce_done[2] = {nullptr, nullptr};
    bool bounce_used[2] = {false, false};   // bounce_done[k] has been recorded
    // optional timing of the front kernel: a ring of event pairs, harvested by jpg_front_time
    int timing = 0;             // 0 = off, n = bracket every n-th front launch
    uint64_t t_seen = 0;
    static constexpr int TIMING_RING = 256;
    hipEvent_t t_ev[2 * TIMING_RING] = {};
    int t_pending = 0;
    double t_total_ms = 0.0;
    uint64_t t_count = 0, t_images = 0;
    int t_imgs[TIMING_RING] = {};   // images of each pending launch
    static constexpr int TIMING_KEEP = 4096;
    float t_sample[TIMING_KEEP] = {};   // duration of each harvested launch, in milliseconds (the first TIMING_KEEP)
    int16_t *h_dc = nullptr;        // pinned
    bool rows_lent = false;         // d_src points at rows the caller keeps alive until the stripe is coded (jpg_stripe_device)
    bool dc_pending = false;        // jpg_last_dc_async has queued the copy into h_dc
    bool code_pending = false;      // jpg_entropy_stripe_async has queued the coding of the stripe
    bool stitch_pending = false;    // jpg_stitch_async has queued the stuffing of the stripe
    // stripe chain (jpg_chain_setup): the mailbox all ranks share, where the pieces are gathered, this context's
    // hand-over record between its gate kernels and the kernels behind them, and the pass counter
    struct Chain {
        ChainMail *mail = nullptr;          // device-side address of the mailbox (host memory)
        const ChainMail *mail_host = nullptr;   // the address the host reads it by (rank 0 only needs it)
        uint8_t *gather = nullptr; size_t gather_cap = 0;
        int rank = 0, world = 0;
        ChainParams *d_params = nullptr;    // device, [2]: one per pass in flight (parity of the pass number)
        // Two passes may be queued before the first is taken: the page-locked records a pass reports through exist
        // twice, indexed by the parity of the pass number
        ChainParams *h_params = nullptr;    // [2] copies of d_params, made at the end of a pass
        int last_of[2] = {0, 0};            // the last rank with a stripe in the pass: it publishes the scan's length
        int slot_of[2] = {0, 0};            // the context's set of back-end buffers the pass used
        int queue_err[2] = {0, 0};          // status of the call that queued the pass
        DevStatus *h_status = nullptr;      // [2] status records of the passes
        hipEvent_t ev[2] = {nullptr, nullptr};   // end of a pass
        uint32_t epoch_of[2] = {0, 0};      // epoch the pass's errors are tagged with
        long long nchunks_of[2] = {0, 0};
        uint64_t pass = 0;                  // passes queued
        uint64_t taken = 0;                 // passes whose results have been handed out
        uint64_t timeout_ticks = 0;
    } chain;
};

static inline jpg_ctx::Slot &S(jpg_ctx *c) { return c->slot[c->cur]; }          // being written
static inline const jpg_ctx::Slot &S(const jpg_ctx *c) { return c->slot[c->cur]; }
static inline jpg_ctx::Slot &V(jpg_ctx *c) { return c->slot[c->view]; }         // being read

#define HIP_TRY(ctx, expr)                   \
    do {                                     \
        hipError_t _e = (expr);              \
        if (_e != hipSuccess) {              \
            (ctx)->last_hip = (int)_e;       \
            (void)hipGetLastError();         \
            return JPG_ERR_HIP;              \
        }                                    \
    } while (0)

extern "C" const char *jpg_strerror(int st) {
    switch (st) {
        case JPG_OK: return "ok";
        case JPG_ERR_INVALID_RATIO: return "Invalid chrominance downsampling ratio!";
        case JPG_ERR_DC_RANGE: return "DC coefficient bit length greater than 11!";
        case JPG_ERR_AC_RANGE: return "AC coefficient bit length greater than 10!";
        case JPG_ERR_HUFF_DC: return "DC Huffman Code Not Found!";
        case JPG_ERR_HUFF_AC: return "AC Huffman Code Not Found!";
        case JPG_ERR_BAD_ARG: return "bad argument";
        case JPG_ERR_HIP: return "HIP runtime error";
        case JPG_ERR_STATE: return "stage called out of order";
        case JPG_ERR_UNSUPPORTED_DIMS:
            return "unsupported dimensions: the reference skews the chroma plane when width % 16 == 1 (h=2) or "
                   "height % 16 == 1 (v=2)";
        case JPG_ERR_NOMEM: return "out of memory";
        case JPG_ERR_CAPACITY: return "output buffer too small";
        case JPG_ERR_TOO_LARGE: return "image too large for this path";
        case JPG_ERR_CHAIN: return "stripe chain: another stripe's values did not arrive in time, or the gather buffer is too small";
    }
    return "unknown status";
}

extern "C" int jpg_last_hip_error(const jpg_ctx *ctx) { return ctx ? ctx->last_hip : 0; }

static int round_up(int x, int m) { return (x + m - 1) / m * m; }

// Entry points make the context's device current and put the caller's device back when they return.
struct DeviceGuard {
    int prev = -1;
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

static int use_device(jpg_ctx *c, DeviceGuard &g) {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) {
        (void)hipGetLastError();
        cur = -1;
    }
    if (cur != c->device) {
        HIP_TRY(c, hipSetDevice(c->device));
        g.prev = cur;
    }
    return JPG_OK;
}

static int sync_all(jpg_ctx *c) {
    if (c->stream_c) HIP_TRY(c, hipStreamSynchronize(c->stream_c));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->stream_b) HIP_TRY(c, hipStreamSynchronize(c->stream_b));
    return JPG_OK;
}

template <class T>
static int grow(jpg_ctx *c, T *&p, size_t &cap, size_t need_elems) {
    if (need_elems <= cap && p) return JPG_OK;
    if (p) {
        if (int e = sync_all(c)) return e;
        HIP_TRY(c, hipFree(p));
        p = nullptr;
        cap = 0;
    }
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, need_elems * sizeof(T));
    if (e != hipSuccess) {
        c->last_hip = (int)e;
        (void)hipGetLastError();
        return e == hipErrorOutOfMemory ? JPG_ERR_NOMEM : JPG_ERR_HIP;
    }
    p = static_cast<T *>(q);
    cap = need_elems;
    return JPG_OK;
}

extern "C" int jpg_create(jpg_ctx **out, int device, int dct_algo, int ra, int rb, int rc) {
    if (!out) return JPG_ERR_BAD_ARG;
    *out = nullptr;
    int h, v;
    if (!ratio_factors(ra, rb, rc, h, v)) return JPG_ERR_INVALID_RATIO;
    if (dct_algo != JPG_DCT_REAL && dct_algo != JPG_DCT_BIN && dct_algo != JPG_DCT_REAL_STRICT) return JPG_ERR_BAD_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        (void)hipGetLastError();
        return JPG_ERR_HIP;
    }
    jpg_ctx *c = new (std::nothrow) jpg_ctx();
    if (!c) return JPG_ERR_NOMEM;
    c->device = device; c->algo = dct_algo; c->h = h; c->v = v;
    DeviceGuard dg;
    if (hipGetDevice(&dg.prev) != hipSuccess) { (void)hipGetLastError(); dg.prev = -1; }
    // (stream priorities were tried for the two streams and made no difference)
    bool ok = hipSetDevice(device) == hipSuccess &&
              hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&c->stream_b, hipStreamNonBlocking) == hipSuccess &&
              hipHostMalloc((void **)&c->h_dc, 4 * sizeof(int16_t), hipHostMallocDefault) == hipSuccess &&
              hipMalloc((void **)&c->d_dc, 4 * sizeof(int16_t)) == hipSuccess;
    for (jpg_ctx::Slot &sl : c->slot) {
        ok = ok && hipEventCreateWithFlags(&sl.e_done, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&sl.s_done, hipEventDisableTiming) == hipSuccess;
#ifdef JPG_TRACE
        ok = ok && hipMalloc((void **)&sl.eb.trace, 4096 * 8 * sizeof(unsigned long long)) == hipSuccess &&
             hipMemset(sl.eb.trace, 0, 4096 * 8 * sizeof(unsigned long long)) == hipSuccess;
#endif
        if (const char *ev = getenv("JPG_STUFF_GRID")) sl.eb.stuff_grid = (size_t)(atoi(ev) > 0 ? atoi(ev) : 0);
    }
    if (const char *ev = getenv("JPG_TWO_KERNELS")) c->two_kernels = atoi(ev) != 0;
    if (const char *ev = getenv("JPG_CODED_WINDOW")) c->window_mode = !strcmp(ev, "small") ? 1 : (!strcmp(ev, "large") ? 2 : 0);
    if (const char *ev = getenv("JPG_BANDS")) c->bands_override = atoi(ev) > 0 ? atoi(ev) : 0;
    if (!ok) {
        (void)hipGetLastError();
        jpg_destroy(c);
        return JPG_ERR_HIP;
    }
    *out = c;
    return JPG_OK;
}

extern "C" void jpg_destroy(jpg_ctx *c) {
    if (!c) return;
    DeviceGuard dg;
    if (hipGetDevice(&dg.prev) != hipSuccess) { (void)hipGetLastError(); dg.prev = -1; }
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->stream_b) (void)hipStreamSynchronize(c->stream_b);
    void *dev[] = {c->d_src_own, c->d_plane[0], c->d_plane[1], c->d_plane[2], c->d_down[0], c->d_down[1], c->d_coef,
                   c->d_tmp16, c->d_dc};
    for (void *p : dev)
        if (p) (void)hipFree(p);
    for (jpg_ctx::Slot &sl : c->slot) {
        void *sdev[] = {sl.d_srcs, sl.eb.clen, sl.eb.sparse, sl.eb.status, sl.eb.bits, sl.eb.gsum, sl.eb.sagg, sl.eb.sgrp, sl.eb.shyp,
                        sl.eb.sticket, sl.eb.scan, sl.eb.trace};
        for (void *p : sdev)
            if (p) (void)hipFree(p);
        if (sl.h_status) (void)hipHostFree(sl.h_status);
        if (sl.e_done) (void)hipEventDestroy(sl.e_done);
        if (sl.s_done) (void)hipEventDestroy(sl.s_done);
    }
    for (hipEvent_t e : c->t_ev)
        if (e) (void)hipEventDestroy(e);
    for (int k = 0; k < 2; k++) {
        if (c->h_bounce[k]) (void)hipHostFree(c->h_bounce[k]);
        if (c->bounce_done[k]) (void)hipEventDestroy(c->bounce_done[k]);
    }
    if (c->h_dc) (void)hipHostFree(c->h_dc);
    if (c->chain.d_params) (void)hipFree(c->chain.d_params);
    if (c->chain.h_params) (void)hipHostFree(c->chain.h_params);
    if (c->chain.h_status) (void)hipHostFree(c->chain.h_status);
    for (hipEvent_t e : c->chain.ev)
        if (e) (void)hipEventDestroy(e);
    if (c->scan_host) (void)hipHostFree(c->scan_host);
    for (hipEvent_t e : c->band_ev)
        if (e) (void)hipEventDestroy(e);
    if (c->src_free) (void)hipEventDestroy(c->src_free);
    if (c->stream_c) { (void)hipStreamSynchronize(c->stream_c); (void)hipStreamDestroy(c->stream_c); }

    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->stream_b) (void)hipStreamDestroy(c->stream_b);
    delete c;
}

extern "C" void *jpg_stream(jpg_ctx *c) { return c ? (void *)c->stream : nullptr; }

extern "C" int jpg_sync(jpg_ctx *c) {
    if (!c) return JPG_ERR_BAD_ARG;
    DeviceGuard dg;
    if (int e = use_device(c, dg)) return e;
    return sync_all(c);
}

namespace jpg {
bool ratio_factors(int ra, int rb, int rc, int &h, int &v) {
    // src/jpeg/sampling.rs:6-22
    if (ra == 4 && rb == 4 && rc == 4) { h = 1; v = 1; }
    else if (ra == 4 && rb == 2 && rc == 0) { h = 2; v = 2; }
    else if (ra == 4 && rb == 2 && rc == 2) { h = 2; v = 1; }
    else return false;
    return true;
}

int make_geom(Geom &g, int32_t W, int32_t H, uint32_t stride, int bottom_up, int h, int v) {
    if (W <= 0 || H <= 0 || (uint64_t)stride < (uint64_t)3 * (uint64_t)W) return JPG_ERR_BAD_ARG;
    // the kernels address the rows of a tile (at most 16) with 32-bit offsets from the tile's lowest row
    if ((uint64_t)stride * 16 + 4096 >= (1ull << 31)) return JPG_ERR_TOO_LARGE;
    g = Geom{};
    g.W = W; g.H = H; g.h = h; g.v = v;
    g.Wp = round_up(W, 8 * g.h); g.Hp = round_up(H, 8 * g.v);
    g.Wc = round_up(W / g.h, 8); g.Hc = round_up(H / g.v, 8);
    if (g.Wc != g.Wp / g.h || g.Hc != g.Hp / g.v) return JPG_ERR_UNSUPPORTED_DIMS;
    g.mcux = g.Wp / (8 * g.h); g.mcuy = g.Hp / (8 * g.v);
    g.nb = g.h * g.v + 2;
    g.stride = stride; g.bottom_up = bottom_up ? 1 : 0;
    g.nblocks = (long long)g.mcux * g.mcuy * g.nb;
    return JPG_OK;
}
}  // namespace jpg

static int set_geometry(jpg_ctx *c, int32_t W, int32_t H, uint32_t stride, int bottom_up) {
    Geom g{};
    if (int e = make_geom(g, W, H, stride, bottom_up, c->h, c->v)) return e;
    c->g = g;
    c->nimg = 1;
    c->stage = ST_NONE; c->have_planes = false; c->down_done = false; c->coef_valid = false;
    c->rows_lent = false; c->dc_pending = false; c->code_pending = false; c->stitch_pending = false;
    return JPG_OK;
}

static int ensure_coef(jpg_ctx *c) {
    const size_t nblk = (size_t)c->g.nblocks * (size_t)(c->nimg > 0 ? c->nimg : 1);
    return grow(c, c->d_coef, c->coef_cap, nblk * 64);
}

template <class T>
static int regrow(jpg_ctx *c, T *&p, size_t n) {
    size_t cap = 0;
    if (p) {
        if (int e = sync_all(c)) return e;
        (void)hipFree(p);
        p = nullptr;
    }
    return grow(c, p, cap, n);
}

// back-end buffers of the slot being written, for c->nimg images of the current geometry;
// coded: the chunks are the tiles of the fused coding kernel instead of runs of 64 blocks
static int ensure_entropy(jpg_ctx *c, bool coded) {
    jpg_ctx::Slot &sl = S(c);
    const size_t nimg = (size_t)(c->nimg > 0 ? c->nimg : 1);
    const size_t nblk = (size_t)c->g.nblocks;
    const size_t nchunks = coded ? (size_t)coded_tiles(c->g) : (nblk + CHUNK_BLOCKS - 1) / CHUNK_BLOCKS;
    const size_t slot_words = coded ? (size_t)coded_slot_words(c->g) : (size_t)CHUNK_STRIDE_WORDS;
    if ((nchunks + 63) / 64 > (size_t)MAX_GROUPS) return JPG_ERR_TOO_LARGE;
    const size_t cstride = (nchunks + 63) / 64 * 64;
    const size_t scan_stride = worst_scan_bytes(nblk);
    const size_t nseg = worst_segments(nblk);
    sl.eb.nimg = (int)nimg;
    sl.eb.nchunks = (long long)nchunks;
    sl.eb.cstride = (long long)cstride;
    sl.eb.slot_words = (int)slot_words;
    sl.nblocks = nblk;
    int e;
    if ((e = grow(c, sl.eb.clen, sl.cap_clen, nimg * cstride))) return e;
    if ((e = grow(c, sl.eb.gsum, sl.cap_gsum, nimg * (cstride / 64)))) return e;
    if ((e = grow(c, sl.eb.sparse, sl.cap_sparse, nimg * cstride * slot_words))) return e;
    if ((e = grow(c, sl.eb.scan, sl.cap_scan, nimg * scan_stride))) return e;
    sl.eb.scan_stride = scan_stride;
    if (nimg > sl.cap_status) {
        if ((e = sync_all(c))) return e;
        if (sl.eb.status) (void)hipFree(sl.eb.status);
        if (sl.h_status) (void)hipHostFree(sl.h_status);
        sl.eb.status = nullptr; sl.h_status = nullptr; sl.cap_status = 0;
        HIP_TRY(c, hipMalloc((void **)&sl.eb.status, nimg * sizeof(DevStatus)));
        // (cleared on the context's stream and waited for: the streams of a context do not wait for the default
        // stream, on which a plain hipMemset may still be pending when the kernels that use the memory start)
        HIP_TRY(c, hipMemsetAsync(sl.eb.status, 0, nimg * sizeof(DevStatus), c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        HIP_TRY(c, hipHostMalloc((void **)&sl.h_status, nimg * sizeof(DevStatus), hipHostMallocDefault));
        memset(sl.h_status, 0, nimg * sizeof(DevStatus));
        sl.cap_status = nimg;
        sl.eb.epoch = 0;
    }
    if (std::max<size_t>(2 * nimg, TICKET_WORDS_SINGLE) > sl.cap_ticket) {
        if ((e = grow(c, sl.eb.sticket, sl.cap_ticket, std::max<size_t>(2 * nimg, TICKET_WORDS_SINGLE)))) return e;
        HIP_TRY(c, hipMemsetAsync(sl.eb.sticket, 0, sl.cap_ticket * sizeof(uint32_t), c->stream));   // the kernel leaves them at 0
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    // the hand-off words are laid out per image with the stride nseg: a change of either relocates them
    if (nseg != sl.eb.nseg_cap || nimg * nseg > sl.cap_sagg) {
        if ((e = grow(c, sl.eb.sagg, sl.cap_sagg, nimg * nseg))) return e;
        if ((e = grow(c, sl.eb.sgrp, sl.cap_sgrp, nimg * (nseg / 64 + 1)))) return e;
        if ((e = grow(c, sl.eb.shyp, sl.cap_shyp, nimg * (nseg / 4096 + 1)))) return e;
        sl.eb.nseg_cap = nseg;
        sl.eb.stuff_epoch = 0;   // forces the hand-off words to be cleared before first use
    }
    return JPG_OK;
}

// the dense copy of the stream is only kept for the staged calls
static int ensure_dense(jpg_ctx *c) {
    jpg_ctx::Slot &sl = V(c);
    // sized for the image whose chunks the viewed slot holds, which need not be the one queued last
    const size_t need = worst_bits_bytes(sl.nblocks);
    if (sl.eb.bits && sl.eb.bits_cap >= need) return JPG_OK;
    if (int e = regrow(c, sl.eb.bits, need / 4)) return e;
    sl.eb.bits_cap = need;
    return JPG_OK;
}

constexpr size_t BOUNCE_BYTES = 8u << 20;

// is `p` host memory the device can read directly (page-locked or registered)?
static bool is_pinned_host(const void *p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();   // an ordinary malloc pointer is reported as an error by some runtimes
        return false;
    }
    return at.type == hipMemoryTypeHost;
}

// pageable rows -> device, through the two bounce buffers: while one chunk travels by DMA the
// next is packed by the CPU (a pageable hipMemcpy2D runs at a fraction of the link speed)
static int copy_rows_staged(jpg_ctx *c, uint8_t *dst, uint32_t pitch, const uint8_t *src, uint32_t stride,
                            size_t row_bytes, int32_t rows, hipStream_t st) {
    for (int k = 0; k < 2; k++) {
        if (!c->h_bounce[k]) HIP_TRY(c, hipHostMalloc((void **)&c->h_bounce[k], BOUNCE_BYTES, hipHostMallocDefault));
        if (!c->bounce_done[k]) HIP_TRY(c, hipEventCreateWithFlags(&c->bounce_done[k], hipEventDisableTiming));
    }
    const int32_t per = (int32_t)(BOUNCE_BYTES / pitch);
    if (per < 1) {   // a row longer than a bounce buffer: let the runtime stage it
        HIP_TRY(c, hipMemcpy2DAsync(dst, pitch, src, stride, row_bytes, (size_t)rows, hipMemcpyHostToDevice, st));
        return JPG_OK;
    }
    int k = 0;
    // (a buffer may still be travelling for an earlier call: its event, once recorded, is waited for)
    bool used[2] = {c->bounce_used[0], c->bounce_used[1]};
    for (int32_t r0 = 0; r0 < rows; r0 += per, k ^= 1) {
        const int32_t n = rows - r0 < per ? rows - r0 : per;
        if (used[k]) HIP_TRY(c, hipEventSynchronize(c->bounce_done[k]));
        uint8_t *b = c->h_bounce[k];
        if (stride == pitch) {
            memcpy(b, src + (size_t)r0 * stride, (size_t)n * pitch - (pitch - row_bytes));
        } else {
            for (int32_t r = 0; r < n; r++) memcpy(b + (size_t)r * pitch, src + (size_t)(r0 + r) * stride, row_bytes);
        }
        HIP_TRY(c, hipMemcpyAsync(dst + (size_t)r0 * pitch, b, (size_t)n * pitch, hipMemcpyHostToDevice, st));
        HIP_TRY(c, hipEventRecord(c->bounce_done[k], st));
        used[k] = true;
        c->bounce_used[k] = true;
    }
    return JPG_OK;
}

// Make `pixels` usable by the kernels: device-resident, 4-byte aligned base
// and stride.  Host memory is always copied; device memory only if misaligned.
static int stage_input(jpg_ctx *c, const uint8_t *pixels, bool on_host) {
    Geom &g = c->g;
    const bool aligned = ((g.stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(pixels) & 3) == 0);
    if (!on_host && aligned) {
        c->d_src = pixels;
        return JPG_OK;
    }
    const uint32_t pitch = (g.stride & 3) == 0 ? g.stride : (uint32_t)round_up(3 * g.W, 4);
    if (int e = grow(c, c->d_src_own, c->src_cap, (size_t)pitch * g.H + 16)) return e;
    if (on_host && !is_pinned_host(pixels)) {
        if (int e = copy_rows_staged(c, c->d_src_own, pitch, pixels, g.stride, (size_t)3 * g.W, g.H, c->stream)) return e;
    } else {
        HIP_TRY(c, hipMemcpy2DAsync(c->d_src_own, pitch, pixels, g.stride, (size_t)3 * g.W, (size_t)g.H,
                                    on_host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, c->stream));
    }
    g.stride = pitch;
    c->d_src = c->d_src_own;
    return JPG_OK;
}

static int load_common(jpg_ctx *c, const uint8_t *pixels, int32_t W, int32_t H, uint32_t stride, int bottom_up,
                       bool on_host) {
    if (!c || !pixels) return JPG_ERR_BAD_ARG;
    DeviceGuard dg;
    if (int e = use_device(c, dg)) return e;
    if (int e = set_geometry(c, W, H, stride, bottom_up)) return e;
    if (int e = stage_input(c, pixels, on_host)) return e;
    const size_t n = (size_t)c->g.Wp * c->g.Hp;
    if (n > c->plane_cap || !c->d_plane[0]) {
        for (int k = 0; k < 3; k++) {
            size_t cap = 0;
            if (c->d_plane[k]) { (void)hipStreamSynchronize(c->stream); (void)hipFree(c->d_plane[k]); c->d_plane[k] = nullptr; }
            if (int e = grow(c, c->d_plane[k], cap, n)) return e;
        }
        c->plane_cap = n;
    }
    HIP_TRY(c, launch_color(c->g, c->d_src, c->d_plane[0], c->d_plane[1], c->d_plane[2], c->stream));
    if (on_host) HIP_TRY(c, hipStreamSynchronize(c->stream));   // caller may reuse `pixels`
    c->stage = ST_PLANES; c->have_planes = true; c->down_done = false;
    return JPG_OK;
}

extern "C" int jpg_load_bgr(jpg_ctx *c, const uint8_t *pixels, int32_t W, int32_t H, uint32_t stride, int bottom_up) {
    return load_common(c, pixels, W, H, stride, bottom_up, true);
}
extern "C" int jpg_load_bgr_device(jpg_ctx *c, const uint8_t *pixels, int32_t W, int32_t H, uint32_t stride,
                                   int bottom_up) {
    return load_common(c, pixels, W, H, stride, bottom_up, false);
}

extern "C" int jpg_downsample(jpg_ctx *c) {
    if (!c) return JPG_ERR_BAD_ARG;
    if (c->stage != ST_PLANES) return JPG_ERR_STATE;
    DeviceGuard dg;
    if (int e = use_device(c, dg)) return e;
    if (c->h != 1 || c->v != 1) {
        const size_t n = (size_t)c->g.Wc * c->g.Hc;
        if (n > c->down_cap || !c->d_down[0]) {
            for (int k = 0; k < 2; k++) {
                size_t cap = 0;
                if (c->d_down[k]) { (void)hipStreamSynchronize(c->stream); (void)hipFree(c->d_down[k]); c->d_down[k] = nullptr; }
                if (int e = grow(c, c->d_down[k], cap, n)) return e;
            }
            c->down_cap = n;
        }
        HIP_TRY(c, launch_downsample(c->g, c->d_plane[1], c->d_plane[2], c->d_down[0], c->d_down[1], c->stream));
    }
    c->stage = ST_DOWN; c->down_done = true;
    return JPG_OK;
}

extern "C" int jpg_dct_quant(jpg_ctx *c) {
    if (!c) return JPG_ERR_BAD_ARG;
    if (c->stage != ST_DOWN) return JPG_ERR_STATE;
    DeviceGuard dg;
    if (int e = use_device(c, dg)) return e;
    if (int e = ensure_coef(c)) return e;
    const bool sub = c->h != 1 || c->v != 1;
    HIP_TRY(c, launch_dct_planes(c->g, c->algo, c->d_plane[0], sub ? c->d_down[0] : c->d_plane[1],
                                 sub ? c->d_down[1] : c->d_plane[2], c->d_coef, c->stream));
    c->coef_valid = true;
    c->stage = ST_COEF;
    return JPG_OK;
}

// add the elapsed times of the recorded event pairs to the totals (the stream must be idle)
static int harvest_timing(jpg_ctx *c) {
    for (int k = 0; k < c->t_pending; k++) {
        float ms = 0.0f;
        HIP_TRY(c, hipEventElapsedTime(&ms, c->t_ev[2 * k], c->t_ev[2 * k + 1]));
        c->t_total_ms += ms;
        if (c->t_count < (uint64_t)jpg_ctx::TIMING_KEEP) c->t_sample[c->t_count] = ms;
        c->t_count++;
        c->t_images += (uint64_t)c->t_imgs[k];
    }
    c->t_pending = 0;
    return JPG_OK;
}

// Launch a kernel that takes a pair of events for its own start and end time stamps (both null when
// this launch is not timed): launch(e0, e1) -> hipError_t.  n = images of the launch.
template <class F>
static int launch_timed(jpg_ctx *c, int n, F launch) {
    if (!c->timing || (c->t_seen++ % (uint64_t)c->timing) != 0) {
        HIP_TRY(c, launch(nullptr, nullptr));
        return JPG_OK;
    }
    if (c->t_pending == jpg_ctx::TIMING_RING) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (int e = harvest_timing(c)) return e;
    }
    hipEvent_t &e0 = c->t_ev[2 * c->t_pending], &e1 = c->t_ev[2 * c->t_pending + 1];
    if (!e0) HIP_TRY(c, hipEventCreate(&e0));
    if (!e1) HIP_TRY(c, hipEventCreate(&e1));
    // the events take the dispatch's own time stamps: no extra packets in the stream
    HIP_TRY(c, launch(e0, e1));
    c->t_imgs[c->t_pending] = n;
    c->t_pending++;
    return JPG_OK;
}

// the fused front kernel (for one image, or for the batch srcs[0..n-1])
static int queue_front(jpg_ctx *c, const uint8_t *const *srcs = nullptr, int n = 1) {
    if (int e = ensure_coef(c)) return e;
    if (int e = launch_timed(c, n, [&](hipEvent_t e0, hipEvent_t e1) {
            return launch_front(c->g, c->algo, c->d_src, c->d_coef, c->stream, e0, e1, srcs, n);
        }))
        return e;
    c->coef_valid = true;
    return JPG_OK;
}

extern "C" int jpg_set_timing(jpg_ctx *c, int every) {
    if (!c) return JPG_ERR_BAD_ARG;
    DeviceGuard dg;
    if (int e = use_device(c, dg)) return e;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->t_pending = 0;
    c->t_total_ms = 0.0;
    c->t_count = 0;
    c->t_images = 0;
    c->t_seen = 0;
    c->timing = every > 0 ? every : 0;
    return JPG_OK;
}

extern "C" int jpg_front_samples(jpg_ctx *c, float *ms, int32_t cap, int32_t *n) {
    if (!c || !n || (cap > 0 && !ms)) return JPG_ERR_BAD_ARG;
    DeviceGuard dg;
    if (int e = use_device(c, dg)) return e;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (int e = harvest_timing(c)) return e;
    const uint64_t have = c->t_count < (uint64_t)jpg_ctx::TIMING_KEEP ? c->t_count : (uint64_t)jpg_ctx::TIMING_KEEP;
    *n = (int32_t)(have < (uint64_t)(cap > 0 ? cap : 0) ? have : (uint64_t)(cap > 0 ? cap : 0));
    for (int32_t k = 0; k < *n; k++) ms[k] = c->t_sample[k];
    return JPG_OK;
}

extern "C" int jpg_front_time(jpg_ctx *c, double *total_ms, uint64_t *launches, uint64_t *images) {
    if (!c) return JPG_ERR_BAD_ARG;
    DeviceGuard dg;
    if (int e = use_device(c, dg)) return e;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (int e = harvest_timing(c)) return e;
    if (total_ms) *total_ms = c->t_total_ms;
    if (launches) *launches = c->t_count;
    if (images) *images = c->t_images;
    return JPG_OK;
}

// Get the slot c->cur ready for a coding kernel on `stream`.  The slot's previous stuffing kernel
// (possibly on stream_b) must have finished with the chunks first.
static int prepare_slot(jpg_ctx *c, bool coded) {
    if (int e = ensure_entropy(c, coded)) return e;
    jpg_ctx::Slot &sl = S(c);
    if (sl.busy) HIP_TRY(c, hipStreamWaitEvent(c->stream, sl.s_done, 0));
    // errors are tagged with a launch number; clear the word before the number repeats
    if (sl.eb.epoch >= (1u << 22)) {
        HIP_TRY(c, hipMemsetAsync(sl.eb.status, 0, sl.cap_status * sizeof(DevStatus), c->stream));
        sl.eb.epoch = 0;
    }
    sl.eb.epoch++;
    sl.eb.host_status = sl.h_status_alt ? sl.h_status_alt : sl.h_status;
    sl.dense_valid = false;
    sl.tail_known = false;
    sl.eb.gsum_ready = false;
    sl.tiles = coded;
    // The small bit window while the tiles of the encode before fitted it with room to spare (half of it on average)
    sl.eb.small_window = coded && coded_small_window(c->g, c->algo) &&
                         (c->window_mode == 1 || (c->window_mode == 0 && c->tile_bits <= 0.5 * coded_small_window_bits()));
    c->view = c->cur;
    return JPG_OK;
}

// what the next launch learns from a finished one: the mean length of its tiles
static void note_tile_bits(jpg_ctx *c, const jpg_ctx::Slot &sl) {
    if (sl.tiles && sl.eb.nchunks > 0) c->tile_bits = (double)sl.h_status->nbits / (double)sl.eb.nchunks;
}

static const int16_t ZERO_PRED[3] = {0, 0, 0};   // src/jpeg/entropy_coding.rs:38-40

static int queue_entropy(jpg_ctx *c, const int16_t pred[3]) {
    if (int e = prepare_slot(c, false)) return e;
    HIP_TRY(c, launch_entropy(c->g, c->d_coef, pred ? pred : ZERO_PRED, S(c).eb, c->stream));
    return JPG_OK;
}

// Queue the stuffing kernel of slot c->cur on stream `st` (new generation of its hand-off words per
// launch).  On stream_b it starts when the slot's entropy kernel is done and leaves `stream` free
// for the next image.
static int queue_stuff(jpg_ctx *c, uint64_t bit_offset, uint8_t head_bits, int is_last, hipStream_t st) {
    jpg_ctx::Slot &sl = S(c);
    if (st != c->stream) {
        HIP_TRY(c, hipEventRecord(sl.e_done, c->stream));
        HIP_TRY(c, hipStreamWaitEvent(st, sl.e_done, 0));
    }
    if (sl.eb.stuff_epoch == 0 || sl.eb.stuff_epoch >= 16383) {
        HIP_TRY(c, hipMemsetAsync(sl.eb.sagg, 0, sl.cap_sagg * sizeof(uint64_t), st));
        HIP_TRY(c, hipMemsetAsync(sl.eb.sgrp, 0, sl.cap_sgrp * sizeof(uint64_t), st));
        HIP_TRY(c, hipMemsetAsync(sl.eb.shyp, 0, sl.cap_shyp * sizeof(uint64_t), st));
        sl.eb.stuff_epoch = 0;
    }
    sl.eb.stuff_epoch++;
    sl.eb.host_status = sl.h_status_alt ? sl.h_status_alt : sl.h_status;
    sl.eb.gsum_ready = sl.tail_known;    // (the tail kernel leaves the sums of the groups of chunks behind)
    HIP_TRY(c, launch_stuff(sl.eb, bit_offset, head_bits, is_last, st));
    HIP_TRY(c, hipEventRecord(sl.s_done, st));
    sl.busy = true;
    return JPG_OK;
}

// status word of a slot's last launch: 0 or the jpg_status a kernel raised
static int kernel_error(const jpg_ctx::Slot &sl) {
    for (int i = 0; i < (sl.eb.nimg > 0 ? sl.eb.nimg : 1); i++) {
        const int v = -sl.h_status[i].error;
        if (v > 0 && (uint32_t)(v >> 8) == sl.eb.epoch) return -(v & 0xff);
    }
    return JPG_OK;
}

// gather the chunks of the viewed slot into the dense stream (also reports nbits)
static int make_dense(jpg_ctx *c) {
    jpg_ctx::Slot &sl = V(c);
    if (sl.dense_valid) return JPG_OK;
    if (int e = ensure_dense(c)) return e;
    sl.eb.host_status = sl.h_status;
    sl.eb.gsum_ready = sl.tail_known;
    HIP_TRY(c, launch_dense(sl.eb, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    sl.dense_valid = true;
    return JPG_OK;
}

// The fused coding kernel leaves no coefficients behind.  When they are asked for afterwards they are
// computed from the context's own copy of the rows; rows the caller owns may be gone by then.
static bool own_rows(const jpg_ctx *c) {
    return c->nimg == 1 && c->d_src && (c->d_src == c->d_src_own || c->rows_lent);
}

static int need_coef(jpg_ctx *c) {
    if (c->coef_valid) return JPG_OK;
    if (!own_rows(c)) return JPG_ERR_STATE;
    return queue_front(c);
}

// may the current image be coded straight from the context's copy of its rows?
static bool code_from_rows(const jpg_ctx *c) {
    return !c->coef_valid && own_rows(c) && !c->two_kernels && coded_supported(c->g);
}

// Queue the coding of the current image (or stripe, from the predictors `pred`) and the kernel that sums up
// its length and picks out its last bits; nothing is waited for.
static int queue_coding(jpg_ctx *c, const int16_t *pred) {
    c->npend = 0;
    if (code_from_rows(c)) {
        // no coefficients were asked for so far: one kernel from the rows to the coded tiles
        if (int e = prepare_slot(c, true)) return e;
        if (int e = launch_timed(c, 1, [&](hipEvent_t e0, hipEvent_t e1) {
                return launch_coded(c->g, c->algo, c->d_src, nullptr, 1, pred ? pred : ZERO_PRED, S(c).eb, c->stream, e0, e1);
            }))
            return e;
    } else {
        if (int e = need_coef(c)) return e;
        if (int e = queue_entropy(c, pred)) return e;
    }
    HIP_TRY(c, launch_tail(S(c).eb, c->stream));
    S(c).tail_known = true;
    return JPG_OK;
}

static int finish_coding(jpg_ctx *c, uint64_t *nbits, uint32_t *last32) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->code_pending = false;
    jpg_ctx::Slot &sl = S(c);
    if (int ke = kernel_error(sl)) return ke;
    sl.nbits = sl.h_status->nbits;
    note_tile_bits(c, sl);
    if (nbits) *nbits = sl.nbits;
    if (last32) *last32 = sl.h_status->last32;
    c->stage = ST_BITS;
    return JPG_OK;
}

static int entropy_common(jpg_ctx *c, const int16_t *pred, uint64_t *nbits) {
    if (!c) return JPG_ERR_BAD_ARG;
    if (c->stage < ST_COEF) return JPG_ERR_STATE;
    DeviceGuard dg;
    if (int e = use_device(c, dg)) return e;
    if (int e = queue_coding(c, pred)) return e;
    return finish_coding(c, nbits, nullptr);
}

extern "C" int jpg_entropy(jpg_ctx *c, uint64_t *nbits) { return entropy_common(c, nullptr, nbits); }
extern "C" int jpg_entropy_stripe(jpg_ctx *c, const int16_t pred[3], uint64_t *nbits) {
    if (!pred) return JPG_ERR_BAD_ARG;
    return entropy_common(c, pred, nbits);
}

extern "C" int jpg_entropy_stripe_async(jpg_ctx *c, const int16_t pred[3]) {
    if (!c || !pred) return JPG_ERR_BAD_ARG;
    if (c->stage < ST_COEF) return JPG_ERR_STATE;
    DeviceGuard dg;
    if (int e = use_device(c, dg)) return e;
    if (int e = queue_coding(c, pred)) return e;
    c->code_pending = true;
    return JPG_OK;
}

extern "C" int jpg_entropy_stripe_finish(jpg_ctx *c, uint64_t *nbits, uint32_t *last32) {
    if (!c) return JPG_ERR_BAD_ARG;
    if (!c->code_pending) return JPG_ERR_STATE;
    DeviceGuard dg;
    if (int e = use_device(c, dg)) return e;
    return finish_coding(c, nbits, last32);
}

extern "C" int jpg_stream_tail(jpg_ctx *c, uint32_t *last32) {
    if (!c || !last32) return JPG_ERR_BAD_ARG;
    if (c->stage < ST_BITS || c->npend || c->code_pending || c->stitch_pending) return JPG_ERR_STATE;
    jpg_ctx::Slot &sl = V(c);
    if ((sl.eb.nimg > 0 ? sl.eb.nimg : 1) != 1) return JPG_ERR_STATE;   // single images only
    if (!sl.tail_known) {
        // a whole encode (jpg_encode_device_async, jpg_encode_scan) never measured the tail: do it now, on the
        // chunks the slot still holds (the stuffing kernel only read them)
        DeviceGuard dg;
        if (int e = use_device(c, dg)) return e;
        if (int e = sync_all(c)) return e;
        sl.eb.host_status = sl.h_status;
        HIP_TRY(c, launch_tail(sl.eb, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        sl.tail_known = true;
    }
    *last32 = sl.h_status->last32;   // left there by the kernel that measured the stream
    return JPG_OK;
}

static int queue_last_dc(jpg_ctx *c) {
    if (code_from_rows(c)) {
        HIP_TRY(c, launch_last_dc_pixels(c->g, c->algo, c->d_src, c->d_dc, c->stream));
    } else {
        if (int e = need_coef(c)) return e;
        HIP_TRY(c, launch_last_dc(c->g, c->d_coef, c->d_dc, c->stream));
    }
    HIP_TRY(c, hipMemcpyAsync(c->h_dc, c->d_dc, 3 * sizeof(int16_t), hipMemcpyDeviceToHost, c->stream));
    c->dc_pending = true;
    return JPG_OK;
}

extern "C" int jpg_last_dc_async(jpg_ctx *c) {
    if (!c) return JPG_ERR_BAD_ARG;
    if (c->stage < ST_COEF) return JPG_ERR_STATE;
    DeviceGuard dg;
    if (int e = use_device(c, dg)) return e;
    return queue_last_dc(c);
}

extern "C" int jpg_last_dc(jpg_ctx *c, int16_t dc[3]) {
    if (!c || !dc) return JPG_ERR_BAD_ARG;
    if (c->stage < ST_COEF) return JPG_ERR_STATE;
    DeviceGuard dg;
    if (int e = use_device(c, dg)) return e;
    if (!c->dc_pending)
        if (int e = queue_last_dc(c)) return e;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->dc_pending = false;
    dc[0] = c->h_dc[0]; dc[1] = c->h_dc[1]; dc[2] = c->h_dc[2];
    return JPG_OK;
}

static int finish_stitch(jpg_ctx *c, size_t *len, uint8_t *tail) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->stitch_pending = false;
    jpg_ctx::Slot &sl = S(c);
    if (int ke = kernel_error(sl)) return ke;
    sl.scan_len = (size_t)sl.h_status->scan_bytes;
    if (len) *len = sl.scan_len;
    if (tail) *tail = (uint8_t)sl.h_status->tail;
    c->stage = ST_SCAN;
    return JPG_OK;
}

extern "C" int jpg_stitch_async(jpg_ctx *c, uint64_t bit_offset, uint8_t head_bits, int is_last) {
    if (!c) return JPG_ERR_BAD_ARG;
    if (c->stage < ST_BITS || c->npend || c->code_pending) return JPG_ERR_STATE;
    DeviceGuard dg;
    if (int e = use_device(c, dg)) return e;
    if (int e = queue_stuff(c, bit_offset, head_bits, is_last, c->stream)) return e;
    c->stitch_pending = true;
    return JPG_OK;
}

extern "C" int jpg_stitch_finish(jpg_ctx *c, size_t *len, uint8_t *tail) {
    if (!c) return JPG_ERR_BAD_ARG;
    if (!c->stitch_pending) return JPG_ERR_STATE;
    DeviceGuard dg;
    if (int e = use_device(c, dg)) return e;
    return finish_stitch(c, len, tail);
}

extern "C" int jpg_stitch(jpg_ctx *c, uint64_t bit_offset, uint8_t head_bits, int is_last, size_t *len, uint8_t *tail) {
    if (int e = jpg_stitch_async(c, bit_offset, head_bits, is_last)) return e;
    return jpg_stitch_finish(c, len, tail);
}

// ---- gathering the pieces of a striped scan in one device buffer

extern "C" int jpg_scan_copy_to_device(jpg_ctx *c, void *dst, int dst_device, size_t cap) {
    if (!c || !dst) return JPG_ERR_BAD_ARG;
    if (c->stage < ST_SCAN || c->stitch_pending) return JPG_ERR_STATE;
    const jpg_ctx::Slot &sl = V(c);
    if (cap < sl.scan_len) return JPG_ERR_CAPACITY;
    if (!sl.scan_len) return JPG_OK;
    DeviceGuard dg;
    if (int e = use_device(c, dg)) return e;
    if (dst_device < 0 || dst_device == c->device) {
        // the same device, memory of another process opened with jpg_ipc_open, or page-locked host memory (unified
        // addressing finds out which)
        HIP_TRY(c, hipMemcpyAsync(dst, sl.eb.scan, sl.scan_len, hipMemcpyDefault, c->stream));
    } else {
        HIP_TRY(c, hipMemcpyPeerAsync(dst, dst_device, sl.eb.scan, c->device, sl.scan_len, c->stream));
    }
    return JPG_OK;
}

extern "C" void *jpg_device_alloc(int device, size_t bytes) {
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; }
    void *p = nullptr;
    if (hipSetDevice(device) != hipSuccess || hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) {
        (void)hipGetLastError();
        p = nullptr;
    }
    if (prev >= 0) (void)hipSetDevice(prev);
    return p;
}

extern "C" void jpg_device_free(int device, void *p) {
    if (!p) return;
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; }
    if (hipSetDevice(device) == hipSuccess) (void)hipFree(p);
    (void)hipGetLastError();
    if (prev >= 0) (void)hipSetDevice(prev);
}

extern "C" int jpg_device_to_host(int device, void *dst, const void *src_dev, size_t bytes) {
    if (!dst || !src_dev) return JPG_ERR_BAD_ARG;
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; }
    bool ok = hipSetDevice(device) == hipSuccess && hipDeviceSynchronize() == hipSuccess &&
              hipMemcpy(dst, src_dev, bytes, hipMemcpyDeviceToHost) == hipSuccess;
    if (!ok) (void)hipGetLastError();
    if (prev >= 0) (void)hipSetDevice(prev);
    return ok ? JPG_OK : JPG_ERR_HIP;
}

static_assert(sizeof(hipIpcMemHandle_t) == JPG_IPC_HANDLE_BYTES, "handle size of the C ABI");

extern "C" int jpg_ipc_export(const void *dev_ptr, uint8_t handle[JPG_IPC_HANDLE_BYTES]) {
    if (!dev_ptr || !handle) return JPG_ERR_BAD_ARG;
    hipIpcMemHandle_t h;
    if (hipIpcGetMemHandle(&h, const_cast<void *>(dev_ptr)) != hipSuccess) {
        (void)hipGetLastError();
        return JPG_ERR_HIP;
    }
    memcpy(handle, &h, sizeof h);
    return JPG_OK;
}

extern "C" int jpg_ipc_open(int device, const uint8_t handle[JPG_IPC_HANDLE_BYTES], void **dev_ptr) {
    if (!handle || !dev_ptr) return JPG_ERR_BAD_ARG;
    *dev_ptr = nullptr;
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; }
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof h);
    const bool ok = hipSetDevice(device) == hipSuccess &&
                    hipIpcOpenMemHandle(dev_ptr, h, hipIpcMemLazyEnablePeerAccess) == hipSuccess;
    if (!ok) (void)hipGetLastError();
    if (prev >= 0) (void)hipSetDevice(prev);
    return ok ? JPG_OK : JPG_ERR_HIP;
}

extern "C" int jpg_ipc_close(void *dev_ptr) {
    if (!dev_ptr) return JPG_ERR_BAD_ARG;
    if (hipIpcCloseMemHandle(dev_ptr) != hipSuccess) {
        (void)hipGetLastError();
        return JPG_ERR_HIP;
    }
    return JPG_OK;
}


// ---- stripe chain: a pass of one stripe with every hand-over between the ranks done by kernels

extern "C" size_t jpg_chain_mailbox_bytes(void) { return CHAIN_MAILBOX_BYTES; }

extern "C" void *jpg_chain_mailbox_alloc(void) {
    void *p = nullptr;
    if (hipHostMalloc(&p, CHAIN_MAILBOX_BYTES, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    memset(p, 0, CHAIN_MAILBOX_BYTES);
    return p;
}

extern "C" int jpg_peer_access(int device, int peer) {
    if (device == peer) return 1;
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, device, peer) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    if (!can) return 0;
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; }
    bool ok = hipSetDevice(device) == hipSuccess;
    if (ok) {
        const hipError_t e = hipDeviceEnablePeerAccess(peer, 0);
        ok = e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled;
    }
    (void)hipGetLastError();
    if (prev >= 0) (void)hipSetDevice(prev);
    return ok ? 1 : 0;
}

// the address the context's device uses for page-locked or registered host memory
static int device_address_of_host(jpg_ctx *c, void *host, void **dev) {
    *dev = nullptr;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, host) != hipSuccess) {
        (void)hipGetLastError();
        *dev = host;   // not known to the runtime as host memory: taken for an address the device reaches as it is
        return JPG_OK;
    }
    if (at.type != hipMemoryTypeHost) { *dev = host; return JPG_OK; }   // device memory (own, a peer's, another process's)
    if (hipHostGetDevicePointer(dev, host, 0) != hipSuccess || !*dev) {
        (void)hipGetLastError();
        *dev = host;   // (memory of another process opened through a handle: one address for host and devices)
    }
    (void)c;
    return JPG_OK;
}

extern "C" int jpg_chain_setup(jpg_ctx *c, int rank, int world, void *mailbox, void *gather, size_t gather_cap) {
    if (!c || !mailbox || !gather || world < 1 || world > CHAIN_MAX_RANKS || rank < 0 || rank >= world) return JPG_ERR_BAD_ARG;
    DeviceGuard dg;
    if (int e = use_device(c, dg)) return e;
    if (int e = sync_all(c)) return e;
    jpg_ctx::Chain &ch = c->chain;
    void *mail_dev = nullptr, *gather_dev = nullptr;
    if (int e = device_address_of_host(c, mailbox, &mail_dev)) return e;
    if (int e = device_address_of_host(c, gather, &gather_dev)) return e;
    if (!ch.d_params) HIP_TRY(c, hipMalloc((void **)&ch.d_params, 2 * sizeof(ChainParams)));
    if (!ch.h_params) HIP_TRY(c, hipHostMalloc((void **)&ch.h_params, 2 * sizeof(ChainParams), hipHostMallocDefault));
    if (!ch.h_status) HIP_TRY(c, hipHostMalloc((void **)&ch.h_status, 2 * sizeof(DevStatus), hipHostMallocDefault));
    for (hipEvent_t &e : ch.ev)
        if (!e) HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIP_TRY(c, hipMemsetAsync(ch.d_params, 0, 2 * sizeof(ChainParams), c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    memset(ch.h_params, 0, 2 * sizeof(ChainParams));
    memset(ch.h_status, 0, 2 * sizeof(DevStatus));
    ch.mail = static_cast<ChainMail *>(mail_dev);
    ch.mail_host = static_cast<const ChainMail *>(mailbox);
    ch.gather = static_cast<uint8_t *>(gather_dev);
    ch.gather_cap = gather_cap;
    ch.rank = rank; ch.world = world;
    ch.pass = 0;
    ch.taken = 0;
    // the pass numbers start again: what earlier passes left in the records must not be taken for theirs
    if (rank == 0) memset(mailbox, 0, CHAIN_MAILBOX_BYTES);
    // how long a gate waits for another rank before the pass fails (wall clock of the device, 100 MHz)
    long ms = 10000;
    if (const char *ev = getenv("JPG_CHAIN_TIMEOUT_MS")) ms = atol(ev) > 0 ? atol(ev) : ms;
    ch.timeout_ticks = (uint64_t)ms * 100000ull;
    return JPG_OK;
}

extern "C" int jpg_chain_pass_async(jpg_ctx *c, uint64_t active_mask) {
    if (!c) return JPG_ERR_BAD_ARG;
    jpg_ctx::Chain &ch = c->chain;
    if (!ch.mail) return JPG_ERR_STATE;
    if (c->stage < ST_COEF || !code_from_rows(c)) return JPG_ERR_STATE;   // a loaded stripe the coding kernel takes
    if (ch.pass - ch.taken >= 2) return JPG_ERR_STATE;                    // two passes are queued already: take one first
    if (!((active_mask >> ch.rank) & 1)) return JPG_ERR_BAD_ARG;
    if (!(active_mask & 1)) return JPG_ERR_BAD_ARG;    // rank 0 takes part in every pass: it paces the others
    if (ch.world < 64 && (active_mask >> ch.world)) return JPG_ERR_BAD_ARG;
    DeviceGuard dg;
    if (int e = use_device(c, dg)) return e;
    c->npend = 0;
    // Two passes may be in flight: each has its own hand-over record, its own half of the gather buffer and its own
    // set of back-end buffers (slots 0 and 1 of the context), all picked by the parity of the pass number.
    const int par = (int)(ch.pass & 1);
    c->cur = par;
    S(c).h_status_alt = &ch.h_status[par];
    if (int e = prepare_slot(c, true)) { S(c).h_status_alt = nullptr; return e; }
    jpg_ctx::Slot &sl = S(c);
    ch.epoch_of[par] = sl.eb.epoch;
    ch.nchunks_of[par] = sl.eb.nchunks;
    ch.slot_of[par] = c->cur;
    int pred_rank = -1, last_rank = ch.rank;
    for (int k = 0; k < ch.world; k++) {
        if (!((active_mask >> k) & 1)) continue;
        if (k < ch.rank) pred_rank = k;
        last_rank = k > last_rank ? k : last_rank;
    }
    const uint64_t pass = ch.pass++;
    ChainParams *params = ch.d_params + par;
    const ChainPass cp{ch.mail, params, sl.eb.status, sl.eb.epoch, 0x8000u | (uint32_t)(pass & 0x7fff),
                       (int)(pass % CHAIN_SLOTS), ch.rank, pred_rank, active_mask, ch.timeout_ticks};
    const size_t half = ch.gather_cap / 2;
    // (One stream: putting the second half of a pass on the context's second stream, beside the coding of the next
    // pass, was measured and gains nothing -- the coding kernel slows down by what the stuffing takes.)
    const hipStream_t sa = c->stream, sb = c->stream;
    // the stripe's last DC coefficients (straight from its last MCU's pixels), published; the predecessor's fetched
    hipError_t he = launch_last_dc_pixels(c->g, c->algo, c->d_src, c->d_dc, sa, &cp);
    sl.eb.chain = params;
    sl.eb.chain_slot = ch.mail + cp.slot * CHAIN_MAX_RANKS + ch.rank;
    sl.eb.chain_tag = cp.tag;
    int e = JPG_OK;
    if (he == hipSuccess)
        e = launch_timed(c, 1, [&](hipEvent_t e0, hipEvent_t e1) {
            return launch_coded(c->g, c->algo, c->d_src, nullptr, 1, ZERO_PRED, sl.eb, sa, e0, e1);
        });
    if (!e && he == hipSuccess) {
        he = launch_tail(sl.eb, sa);                           // bit count and last bits, published
        sl.tail_known = he == hipSuccess;
        if (he == hipSuccess) he = launch_chain_place(cp, sb);
    }
    if (!e && he == hipSuccess) e = queue_stuff(c, 0, 0, ch.rank == last_rank, sb);   // shift and head bits from the gate
    sl.eb.chain = nullptr; sl.eb.chain_slot = nullptr; sl.eb.chain_tag = 0;
    sl.h_status_alt = nullptr;
    // (a pass that could not be queued whole still counts: the ranks number their passes alike, and its gates run dry)
    if (!e && he == hipSuccess) he = launch_chain_offset(cp, half, sb);
    if (!e && he == hipSuccess) he = launch_chain_copy(cp, sl.eb.scan, ch.gather + (size_t)par * half, sb);
    ch.last_of[par] = last_rank;
    if (!e && he == hipSuccess) he = launch_chain_done(cp, ch.rank == last_rank, sb);
    if (!e && he == hipSuccess) he = hipMemcpyAsync(&ch.h_params[par], params, sizeof(ChainParams), hipMemcpyDeviceToHost, sb);
    // the set of buffers is free again when the piece has been copied out of it (prepare_slot waits for this event)
    (void)hipEventRecord(sl.s_done, sb);
    sl.busy = true;
    (void)hipEventRecord(ch.ev[par], sb);
    // (what jpg_chain_pass_finish reports for a pass that was not queued whole)
    ch.queue_err[par] = e ? e : (he != hipSuccess ? JPG_ERR_HIP : JPG_OK);
    if (e) return e;
    HIP_TRY(c, he);
    return JPG_OK;
}

extern "C" int jpg_chain_pass_finish(jpg_ctx *c, uint64_t *nbits, size_t *piece_len, uint64_t *bit_offset, size_t *byte_offset,
                                     size_t *total) {
    if (!c) return JPG_ERR_BAD_ARG;
    jpg_ctx::Chain &ch = c->chain;
    if (ch.pass == ch.taken) return JPG_ERR_STATE;
    DeviceGuard dg;
    if (int e = use_device(c, dg)) return e;
    const int par = (int)(ch.taken & 1);                         // the oldest pass not handed out yet
    HIP_TRY(c, hipEventSynchronize(ch.ev[par]));
    ch.taken++;
    jpg_ctx::Slot &sl = c->slot[ch.slot_of[par]];
    c->view = ch.slot_of[par];
    if (ch.queue_err[par]) return ch.queue_err[par];
    const DevStatus &hs = ch.h_status[par];
    {
        const int v = -hs.error;
        if (v > 0 && (uint32_t)(v >> 8) == ch.epoch_of[par]) return -(v & 0xff);
    }
    sl.nbits = hs.nbits;
    sl.scan_len = (size_t)hs.scan_bytes;
    if (ch.nchunks_of[par] > 0) c->tile_bits = (double)hs.nbits / (double)ch.nchunks_of[par];
    c->stage = ST_SCAN;
    const ChainParams &hp = ch.h_params[par];
    if (hp.failed) return JPG_ERR_CHAIN;
    // Rank 0 (the owner of the gather buffer) reports the whole scan: the last rank with a stripe publishes its length
    // when every piece has landed; the host reads that word (the mailbox is host memory), for as long as a gate would wait.
    uint64_t whole = 0;
    if (ch.rank == 0) {
        const uint64_t tag = 0x8000u | (uint32_t)((ch.taken - 1) & 0x7fff);
        const uint64_t *word = &ch.mail_host[((ch.taken - 1) % CHAIN_SLOTS) * CHAIN_MAX_RANKS + ch.last_of[par]].total;
        const auto t0 = std::chrono::steady_clock::now();
        uint64_t w = 0;
        for (;;) {
            w = __atomic_load_n(word, __ATOMIC_ACQUIRE);
            if ((w >> 48) == tag) break;
            const double ticks = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e8;
            if (ticks > (double)ch.timeout_ticks) return JPG_ERR_CHAIN;
            std::this_thread::yield();
        }
        if (w & (1ull << 47)) return JPG_ERR_CHAIN;
        whole = w & ((1ull << 47) - 1);
    }
    if (nbits) *nbits = sl.nbits;
    if (piece_len) *piece_len = sl.scan_len;
    if (bit_offset) *bit_offset = hp.bit_offset;
    if (byte_offset) *byte_offset = (size_t)par * (ch.gather_cap / 2) + (size_t)hp.byte_offset;
    if (total) *total = (size_t)whole;
    return JPG_OK;
}

extern "C" int jpg_stuff(jpg_ctx *c, size_t *len) { return jpg_stitch(c, 0, 0, 1, len, nullptr); }

extern "C" int jpg_front_device(jpg_ctx *c, const uint8_t *dev_pixels, int32_t W, int32_t H, uint32_t stride,
                                int bottom_up) {
    if (!c || !dev_pixels) return JPG_ERR_BAD_ARG;
    DeviceGuard dg;
    if (int e = use_device(c, dg)) return e;
    if (int e = set_geometry(c, W, H, stride, bottom_up)) return e;
    if (int e = stage_input(c, dev_pixels, false)) return e;
    if (int e = queue_front(c)) return e;
    c->stage = ST_COEF;
    return JPG_OK;
}

extern "C" int jpg_stripe_device(jpg_ctx *c, const uint8_t *dev_pixels, int32_t W, int32_t H, uint32_t stride,
                                 int bottom_up) {
    if (!c || !dev_pixels) return JPG_ERR_BAD_ARG;
    DeviceGuard dg;
    if (int e = use_device(c, dg)) return e;
    if (int e = set_geometry(c, W, H, stride, bottom_up)) return e;
    if (int e = stage_input(c, dev_pixels, false)) return e;
    // Rows the caller keeps alive until the stripe is coded: where the single coding kernel applies nothing is
    // computed yet (jpg_last_dc transforms the last MCU, jpg_entropy_stripe codes straight from the rows)
    c->rows_lent = true;
    if (!c->two_kernels && coded_supported(c->g)) c->coef_valid = false;
    else if (int e = queue_front(c)) return e;
    c->stage = ST_COEF;
    return JPG_OK;
}

// rows from host memory into the context's own device copy; wait: until the caller may reuse `pixels`
static int front_host_common(jpg_ctx *c, const uint8_t *pixels, int32_t W, int32_t H, uint32_t stride, int bottom_up,
                             bool wait) {
    if (!c || !pixels) return JPG_ERR_BAD_ARG;
    DeviceGuard dg;
    if (int e = use_device(c, dg)) return e;
    if (int e = set_geometry(c, W, H, stride, bottom_up)) return e;
    if (int e = stage_input(c, pixels, true)) return e;
    if (wait) HIP_TRY(c, hipStreamSynchronize(c->stream));
    // The context now holds its own copy of the rows.  Where the single coding kernel applies, the
    // transform waits for the call that tells what is wanted: coefficients (jpg_get_coeffs: front kernel)
    // or bits (jpg_entropy, jpg_entropy_stripe: coding kernel, no coefficients in memory).
    if (!c->two_kernels && coded_supported(c->g)) c->coef_valid = false;
    else if (int e = queue_front(c)) return e;
    c->stage = ST_COEF;
    return JPG_OK;
}

extern "C" int jpg_front_host(jpg_ctx *c, const uint8_t *pixels, int32_t W, int32_t H, uint32_t stride, int bottom_up) {
    return front_host_common(c, pixels, W, H, stride, bottom_up, true);
}

extern "C" int jpg_stripe_host_async(jpg_ctx *c, const uint8_t *pixels, int32_t W, int32_t H, uint32_t stride,
                                     int bottom_up) {
    return front_host_common(c, pixels, W, H, stride, bottom_up, false);
}

// A whole encode of the staged rows (or of the batch srcs[0..n-1]): coding on `stream`, stuffing on
// stream_b, into the slot the previous encodes do not use.  The fused coding kernel where the
// geometry allows it, else the front kernel and the entropy kernel.
// srcs: host array of the batch's device pointers.
static int queue_encode(jpg_ctx *c, const uint8_t *const *srcs = nullptr, int n = 1) {
    c->cur = (c->cur + 1) % jpg_ctx::NSLOT;
    if (!c->two_kernels && coded_supported(c->g) && n <= MAX_BATCH_IMAGES) {
        if (int e = prepare_slot(c, true)) return e;
        if (int e = launch_timed(c, n, [&](hipEvent_t e0, hipEvent_t e1) {
                return launch_coded(c->g, c->algo, c->d_src, srcs, n, ZERO_PRED, S(c).eb, c->stream, e0, e1);
            }))
            return e;
        c->coef_valid = false;
    } else {
        const uint8_t *const *d_srcs = nullptr;
        if (srcs) {   // the front kernel reads the pointers from memory
            jpg_ctx::Slot &sl = S(c);
            if (int e = grow(c, sl.d_srcs, sl.cap_srcs, (size_t)n)) return e;
            HIP_TRY(c, hipMemcpyAsync(sl.d_srcs, srcs, (size_t)n * sizeof(uint8_t *), hipMemcpyHostToDevice, c->stream));
            d_srcs = sl.d_srcs;
        }
        if (int e = queue_front(c, d_srcs, n)) return e;
        if (int e = queue_entropy(c, nullptr)) return e;
    }
    c->stage = ST_COEF;
    if (int e = queue_stuff(c, 0, 0, 1, c->stream_b)) return e;
    if (c->npend == jpg_ctx::NSLOT) {   // the oldest queued result is overwritten unread, as documented
        for (int k = 1; k < c->npend; k++) c->fifo[k - 1] = c->fifo[k];
        c->npend--;
    }
    c->fifo[c->npend++] = c->cur;
    return JPG_OK;
}

extern "C" int jpg_encode_device_async(jpg_ctx *c, const uint8_t *dev_pixels, int32_t W, int32_t H, uint32_t stride,
                                       int bottom_up) {
    if (!c || !dev_pixels) return JPG_ERR_BAD_ARG;
    DeviceGuard dg;
    if (int e = use_device(c, dg)) return e;
    if (int e = set_geometry(c, W, H, stride, bottom_up)) return e;
    if (int e = stage_input(c, dev_pixels, false)) return e;
    return queue_encode(c);
}

// ---- batches of equally sized images: one launch of each kernel for all of them ----

extern "C" int jpg_max_batch(void) { return MAX_BATCH_IMAGES; }

extern "C" int jpg_encode_batch_device_async(jpg_ctx *c, const uint8_t *const *dev_pixels, int32_t n, int32_t W, int32_t H,
                                             uint32_t stride, int bottom_up) {
    if (!c || !dev_pixels || n < 1 || n > jpg_max_batch()) return JPG_ERR_BAD_ARG;
    DeviceGuard dg;
    if (int e = use_device(c, dg)) return e;
    if (int e = set_geometry(c, W, H, stride, bottom_up)) return e;
    if ((int64_t)c->g.nblocks * n >= (1LL << 31)) return JPG_ERR_TOO_LARGE;
    // a batch is read in place: rows must be dword-aligned (single images are re-pitched instead)
    if (stride & 3) return JPG_ERR_BAD_ARG;
    for (int i = 0; i < n; i++)
        if (!dev_pixels[i] || (reinterpret_cast<uintptr_t>(dev_pixels[i]) & 3)) return JPG_ERR_BAD_ARG;
    c->nimg = n;
    c->d_src = dev_pixels[0];
    return queue_encode(c, dev_pixels, n);
}

extern "C" int jpg_batch_result(jpg_ctx *c, int32_t i, uint64_t *nbits, size_t *stuffed_len, const uint8_t **scan_dev) {
    if (!c) return JPG_ERR_BAD_ARG;
    if (c->stage < ST_SCAN) return JPG_ERR_STATE;
    const jpg_ctx::Slot &sl = V(c);
    if (i < 0 || i >= (sl.eb.nimg > 0 ? sl.eb.nimg : 1)) return JPG_ERR_BAD_ARG;
    if (nbits) *nbits = sl.h_status[i].nbits;
    if (stuffed_len) *stuffed_len = (size_t)sl.h_status[i].scan_bytes;
    if (scan_dev) *scan_dev = sl.eb.scan + (size_t)i * sl.eb.scan_stride;
    return JPG_OK;
}

extern "C" int jpg_get_batch_scan(jpg_ctx *c, int32_t i, uint8_t *out, size_t cap) {
    if (!c || !out) return JPG_ERR_BAD_ARG;
    size_t len = 0;
    const uint8_t *src = nullptr;
    if (int e = jpg_batch_result(c, i, nullptr, &len, &src)) return e;
    if (cap < len) return JPG_ERR_CAPACITY;
    DeviceGuard dg;
    if (int e = use_device(c, dg)) return e;
    HIP_TRY(c, hipMemcpyAsync(out, src, len, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return JPG_OK;
}

#ifdef JPG_TRACE
static void print_trace(const jpg_ctx::Slot &sl) {
    if (!sl.eb.trace || !getenv("JPG_TRACE_PRINT")) return;
    static unsigned long long h[4096 * 8];
    (void)hipMemcpy(h, sl.eb.trace, sizeof h, hipMemcpyDeviceToHost);
    double sum[8] = {0}, life_max = 0;
    int n = 0;
    for (int b = 0; b < 4096; b++) {
        double life = 0;
        for (int k = 0; k < 8; k++) life += (double)h[b * 8 + k];
        if (life == 0) continue;
        n++;
        for (int k = 0; k < 8; k++) sum[k] += (double)h[b * 8 + k];
        life_max = life > life_max ? life : life_max;
    }
    fprintf(stderr, "stuffing kernel, %d workgroups, us per workgroup in phases 0..7:", n);
    double all = 0;
    for (int k = 0; k < 8; k++) { fprintf(stderr, " %7.2f", sum[k] / (n ? n : 1) / 100.0); all += sum[k]; }
    fprintf(stderr, "  life %7.2f (max %7.2f)\n", all / (n ? n : 1) / 100.0, life_max / 100.0);
    // the workgroups that spent the longest outside the wait for other workgroups (phase 6)
    for (int top = 0; top < 3; top++) {
        int best = -1;
        double best_busy = -1;
        for (int b = 0; b < 4096; b++) {
            double busy = 0;
            for (int k = 0; k < 8; k++) busy += k == 6 ? 0 : (double)h[b * 8 + k];
            if (busy > best_busy) { best_busy = busy; best = b; }
        }
        if (best < 0 || best_busy <= 0) break;
        fprintf(stderr, "  busiest workgroup %4d:", best);
        for (int k = 0; k < 8; k++) fprintf(stderr, " %7.2f", (double)h[best * 8 + k] / 100.0);
        fprintf(stderr, "\n");
        for (int k = 0; k < 8; k++) h[best * 8 + k] = 0;
    }
    (void)hipMemset(sl.eb.trace, 0, sizeof h);
}
#endif

// results of a finished slot
static int report(jpg_ctx *c, int k, uint64_t *nbits, size_t *stuffed_len) {
    jpg_ctx::Slot &sl = c->slot[k];
    c->view = k;
    if (int ke = kernel_error(sl)) return ke;
    sl.nbits = sl.h_status->nbits;
    sl.scan_len = (size_t)sl.h_status->scan_bytes;
    note_tile_bits(c, sl);
    if (nbits) *nbits = sl.nbits;
    if (stuffed_len) *stuffed_len = sl.scan_len;
    c->stage = ST_SCAN;
    return JPG_OK;
}

extern "C" int jpg_finish(jpg_ctx *c, uint64_t *nbits, size_t *stuffed_len) {
    if (!c) return JPG_ERR_BAD_ARG;
    if (!c->npend) return JPG_ERR_STATE;
    DeviceGuard dg;
    if (int e = use_device(c, dg)) return e;
    if (int e = sync_all(c)) return e;
    c->npend = 0;
#ifdef JPG_TRACE
    print_trace(S(c));
#endif
    return report(c, c->cur, nbits, stuffed_len);
}

extern "C" int jpg_finish_next(jpg_ctx *c, uint64_t *nbits, size_t *stuffed_len) {
    if (!c) return JPG_ERR_BAD_ARG;
    if (!c->npend) return JPG_ERR_STATE;
    DeviceGuard dg;
    if (int e = use_device(c, dg)) return e;
    const int k = c->fifo[0];
    HIP_TRY(c, hipEventSynchronize(c->slot[k].s_done));
    for (int j = 1; j < c->npend; j++) c->fifo[j - 1] = c->fifo[j];
    c->npend--;
    return report(c, k, nbits, stuffed_len);
}

// A band could not be queued: the bands before it are (copies and kernels that use the caller's rows and the
// slot), so everything queued is waited for before the error goes back to the caller.
static int band_failed(jpg_ctx *c, int status) {
    (void)hipGetLastError();
    (void)sync_all(c);
    return status;
}

// A whole encode of rows in host memory, band by band: the rows of a band of MCU rows are copied on stream_c
// and the band is coded on `stream` as soon as they have landed, while the following bands are still
// travelling (a tile needs its own rows and the MCU before it, which lies in its own band or the one above).
// Geometry set; only where the single coding kernel applies.
static int queue_encode_host_banded(jpg_ctx *c, const uint8_t *pixels, uint32_t host_stride) {
    Geom &g = c->g;
    const uint32_t pitch = (host_stride & 3) == 0 ? host_stride : (uint32_t)round_up(3 * g.W, 4);
    if (int e = grow(c, c->d_src_own, c->src_cap, (size_t)pitch * g.H + 16)) return e;
    g.stride = pitch;
    c->d_src = c->d_src_own;
    if (!c->stream_c) HIP_TRY(c, hipStreamCreateWithFlags(&c->stream_c, hipStreamNonBlocking));
    if (!c->src_free) HIP_TRY(c, hipEventCreateWithFlags(&c->src_free, hipEventDisableTiming));
    const bool pinned = is_pinned_host(pixels);
    c->cur = (c->cur + 1) % jpg_ctx::NSLOT;
    if (int e = prepare_slot(c, true)) return e;
    // the copies overwrite the context's rows: not before the kernels queued earlier have read them
    HIP_TRY(c, hipEventRecord(c->src_free, c->stream));
    HIP_TRY(c, hipStreamWaitEvent(c->stream_c, c->src_free, 0));
    const int mcu_h = 8 * g.v;
    // A few bands are enough: all but the last band's coding is hidden behind the copies that follow, and every
    // band costs an event, a wait and a launch (JPG_BANDS overrides, for experiments).  Measured on 7680x4320
    // rows in page-locked memory: 2.02-2.05 ms with 1, 2, 4 or 8 bands and with one or two copy streams -- the
    // copy itself (99.5 MB at about 50 GB/s) is all of it, the 45 us of coding disappear behind it either way.
    // Rows in pageable memory are packed into the staging buffers by the CPU, which is what their time is made
    // of (3.7 ms for the same image): cutting them into bands only adds waits between the chunks (4.1-4.4 ms).
    int nb = !pinned ? 1 : (g.mcuy / 32 < 4 ? g.mcuy / 32 : 4);
    if (c->bands_override) nb = c->bands_override;
    if (nb < 1) nb = 1;
    if (nb > jpg_ctx::MAX_BANDS) nb = jpg_ctx::MAX_BANDS;
    const int per = (g.mcuy + nb - 1) / nb;
    for (int b = 0, r0 = 0; r0 < g.mcuy; b++, r0 += per) {
        const int rows = r0 + per <= g.mcuy ? per : g.mcuy - r0;
        const int y0 = r0 * mcu_h, y1 = (r0 + rows) * mcu_h < g.H ? (r0 + rows) * mcu_h : g.H;   // image rows of the band
        const int f0 = g.bottom_up ? g.H - y1 : y0, n = y1 - y0;                                // ... in the caller's array
        const hipStream_t cs = c->stream_c;
        if (n > 0) {
            uint8_t *dst = c->d_src_own + (size_t)f0 * pitch;
            const uint8_t *src = pixels + (size_t)f0 * host_stride;
            if (pinned && pitch == host_stride) {   // whole rows, one after the other: a plain copy
                HIP_TRY(c, hipMemcpyAsync(dst, src, (size_t)(n - 1) * pitch + (size_t)3 * g.W, hipMemcpyHostToDevice, cs));
            } else if (pinned) {
                HIP_TRY(c, hipMemcpy2DAsync(dst, pitch, src, host_stride, (size_t)3 * g.W, (size_t)n, hipMemcpyHostToDevice,
                                            cs));
            } else if (int e = copy_rows_staged(c, dst, pitch, src, host_stride, (size_t)3 * g.W, n, cs)) {
                return band_failed(c, e);
            }
        }
        hipEvent_t &ev = c->band_ev[b];
        if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return band_failed(c, JPG_ERR_HIP);
        if (hipEventRecord(ev, cs) != hipSuccess || hipStreamWaitEvent(c->stream, ev, 0) != hipSuccess)
            return band_failed(c, JPG_ERR_HIP);
        // (with timing on, the image counts once: its last band carries it, the other launches count no image)
        const bool last_band = r0 + per >= g.mcuy;
        if (int e = launch_timed(c, last_band ? 1 : 0, [&](hipEvent_t e0, hipEvent_t e1) {
                return launch_coded(g, c->algo, c->d_src, nullptr, 1, ZERO_PRED, S(c).eb, c->stream, e0, e1, r0, rows);
            }))
            return band_failed(c, e);
    }
    c->coef_valid = false;
    c->stage = ST_COEF;
    if (int e = queue_stuff(c, 0, 0, 1, c->stream_b)) return e;
    if (c->npend == jpg_ctx::NSLOT) {
        for (int k = 1; k < c->npend; k++) c->fifo[k - 1] = c->fifo[k];
        c->npend--;
    }
    c->fifo[c->npend++] = c->cur;
    return JPG_OK;
}

extern "C" int jpg_encode_scan_begin(jpg_ctx *c, const uint8_t *pixels, int32_t W, int32_t H, uint32_t stride, int bottom_up) {
    if (!c || !pixels) return JPG_ERR_BAD_ARG;
    DeviceGuard dg;
    if (int e = use_device(c, dg)) return e;
    if (int e = set_geometry(c, W, H, stride, bottom_up)) return e;
    c->scan_begun = false;
    if (!c->two_kernels && coded_supported(c->g)) {
        if (int e = queue_encode_host_banded(c, pixels, stride)) return e;
    } else {
        if (int e = stage_input(c, pixels, true)) return e;
        if (int e = queue_encode(c)) return e;
    }
    // The output buffer is page-locked memory the caller will own; pinning pages takes a while, so it is
    // allocated now, while the device works, for the size of the previous scan plus a margin (first call: a
    // quarter byte per pixel); _end allocates again in the rare case that it is too small.
    const size_t guess = c->last_scan_len ? c->last_scan_len + c->last_scan_len / 4 + 4096 : (size_t)W * H / 4 + 4096;
    if (c->scan_host && c->scan_host_cap < guess) {
        (void)hipHostFree(c->scan_host);
        c->scan_host = nullptr;
        c->scan_host_cap = 0;
    }
    if (!c->scan_host) {
        void *host = nullptr;
        if (hipHostMalloc(&host, guess, hipHostMallocDefault) == hipSuccess) {
            c->scan_host = static_cast<uint8_t *>(host);
            c->scan_host_cap = guess;
        } else {
            (void)hipGetLastError();   // _end tries again with the exact size
        }
    }
    c->scan_begun = true;
    return JPG_OK;
}

extern "C" int jpg_encode_scan_end(jpg_ctx *c, uint8_t **stuffed, size_t *len) {
    if (!c || !stuffed || !len) return JPG_ERR_BAD_ARG;
    *stuffed = nullptr; *len = 0;
    if (!c->scan_begun) return JPG_ERR_STATE;
    c->scan_begun = false;
    DeviceGuard dg;
    if (int e = use_device(c, dg)) return e;
    // (the caller's rows have all been copied when the last band's kernel has run: jpg_finish waits for more)
    if (int e = jpg_finish(c, nullptr, nullptr)) return e;
    const jpg_ctx::Slot &sl = V(c);
    c->last_scan_len = sl.scan_len;
    if (c->scan_host && c->scan_host_cap < sl.scan_len) {
        (void)hipHostFree(c->scan_host);
        c->scan_host = nullptr;
        c->scan_host_cap = 0;
    }
    if (!c->scan_host) {
        void *host = nullptr;
        hipError_t he = hipHostMalloc(&host, sl.scan_len ? sl.scan_len : 1, hipHostMallocDefault);
        if (he != hipSuccess) { c->last_hip = (int)he; (void)hipGetLastError(); return JPG_ERR_NOMEM; }
        c->scan_host = static_cast<uint8_t *>(host);
        c->scan_host_cap = sl.scan_len ? sl.scan_len : 1;
    }
    hipError_t he = hipSuccess;
    if (sl.scan_len) {
        he = hipMemcpyAsync(c->scan_host, sl.eb.scan, sl.scan_len, hipMemcpyDeviceToHost, c->stream);
        if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
    }
    if (he != hipSuccess) { c->last_hip = (int)he; (void)hipGetLastError(); return JPG_ERR_HIP; }
    *stuffed = c->scan_host;       // the caller's from here on (jpg_free)
    *len = sl.scan_len;
    c->scan_host = nullptr;
    c->scan_host_cap = 0;
    return JPG_OK;
}

extern "C" int jpg_encode_scan(jpg_ctx *c, const uint8_t *pixels, int32_t W, int32_t H, uint32_t stride, int bottom_up,
                               uint8_t **stuffed, size_t *len) {
    if (!c || !pixels || !stuffed || !len) return JPG_ERR_BAD_ARG;
    *stuffed = nullptr; *len = 0;
    if (int e = jpg_encode_scan_begin(c, pixels, W, H, stride, bottom_up)) return e;
    return jpg_encode_scan_end(c, stuffed, len);
}

extern "C" void *jpg_host_alloc(size_t bytes) {
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}

extern "C" void jpg_free(void *p) {
    if (p) (void)hipHostFree(p);
}

extern "C" const uint8_t *jpg_bits_device(jpg_ctx *c) {
    if (!c || c->stage < ST_BITS || c->npend) return nullptr;
    DeviceGuard dg;
    if (use_device(c, dg) || make_dense(c)) return nullptr;
    return reinterpret_cast<const uint8_t *>(V(c).eb.bits);
}
extern "C" const uint8_t *jpg_scan_device(jpg_ctx *c) { return c && c->stage >= ST_SCAN ? V(c).eb.scan : nullptr; }

extern "C" int jpg_get_bits(jpg_ctx *c, uint8_t *out, size_t cap) {
    if (!c || !out) return JPG_ERR_BAD_ARG;
    if (c->stage < ST_BITS || c->npend) return JPG_ERR_STATE;   // (queued encodes still write the slots)
    const size_t n = (size_t)((V(c).nbits + 7) >> 3);
    if (cap < n) return JPG_ERR_CAPACITY;
    DeviceGuard dg;
    if (int e = use_device(c, dg)) return e;
    if (int e = make_dense(c)) return e;
    HIP_TRY(c, hipMemcpyAsync(out, V(c).eb.bits, n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return JPG_OK;
}

extern "C" int jpg_get_scan(jpg_ctx *c, uint8_t *out, size_t cap) {
    if (!c || !out) return JPG_ERR_BAD_ARG;
    if (c->stage < ST_SCAN) return JPG_ERR_STATE;
    const jpg_ctx::Slot &sl = V(c);
    if (cap < sl.scan_len) return JPG_ERR_CAPACITY;
    DeviceGuard dg;
    if (int e = use_device(c, dg)) return e;
    HIP_TRY(c, hipMemcpyAsync(out, sl.eb.scan, sl.scan_len, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return JPG_OK;
}

extern "C" int jpg_plane_dims(jpg_ctx *c, int comp, int32_t *w, int32_t *h) {
    if (!c || comp < 0 || comp > 2) return JPG_ERR_BAD_ARG;
    if (c->stage == ST_NONE) return JPG_ERR_STATE;
    const bool down = comp > 0 && c->down_done && (c->h != 1 || c->v != 1);
    if (w) *w = down ? c->g.Wc : c->g.Wp;
    if (h) *h = down ? c->g.Hc : c->g.Hp;
    return JPG_OK;
}

extern "C" int jpg_coeff_dims(jpg_ctx *c, int comp, int32_t *w, int32_t *h) {
    if (!c || comp < 0 || comp > 2) return JPG_ERR_BAD_ARG;
    if (c->stage == ST_NONE) return JPG_ERR_STATE;
    if (w) *w = comp ? c->g.Wc : c->g.Wp;
    if (h) *h = comp ? c->g.Hc : c->g.Hp;
    return JPG_OK;
}

extern "C" int jpg_get_plane(jpg_ctx *c, int comp, uint8_t *out, size_t cap) {
    if (!c || !out || comp < 0 || comp > 2) return JPG_ERR_BAD_ARG;
    if (!c->have_planes) return JPG_ERR_STATE;
    int32_t w, h;
    if (int e = jpg_plane_dims(c, comp, &w, &h)) return e;
    const size_t n = (size_t)w * h;
    if (cap < n) return JPG_ERR_CAPACITY;
    DeviceGuard dg;
    if (int e = use_device(c, dg)) return e;
    const bool down = comp > 0 && c->down_done && (c->h != 1 || c->v != 1);
    const uint8_t *src = down ? c->d_down[comp - 1] : c->d_plane[comp];
    HIP_TRY(c, hipMemcpyAsync(out, src, n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return JPG_OK;
}

extern "C" int jpg_get_coeffs(jpg_ctx *c, int comp, int16_t *out, size_t cap) {
    if (!c || !out || comp < 0 || comp > 2) return JPG_ERR_BAD_ARG;
    if (c->stage < ST_COEF) return JPG_ERR_STATE;
    const size_t n = comp ? (size_t)c->g.Wc * c->g.Hc : (size_t)c->g.Wp * c->g.Hp;
    if (cap < n * sizeof(int16_t)) return JPG_ERR_CAPACITY;
    DeviceGuard dg;
    if (int e = use_device(c, dg)) return e;
    if (int e = need_coef(c)) return e;
    if (int e = grow(c, c->d_tmp16, c->tmp16_cap, n)) return e;
    HIP_TRY(c, launch_unscan(c->g, c->d_coef, comp, c->d_tmp16, c->stream));
    HIP_TRY(c, hipMemcpyAsync(out, c->d_tmp16, n * sizeof(int16_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return JPG_OK;
}

extern "C" uint64_t jpg_front_bytes(const jpg_ctx *c) {
    if (!c || c->g.W == 0) return 0;
    return (uint64_t)c->g.stride * c->g.H + 2ull * 64ull * (uint64_t)c->g.nblocks;
}
extern "C" uint64_t jpg_num_blocks(const jpg_ctx *c) { return c ? (uint64_t)c->g.nblocks : 0; }
extern "C" int jpg_fused_coding(const jpg_ctx *c) {
    return c && c->g.W > 0 && !c->two_kernels && coded_supported(c->g) ? 1 : 0;
}
