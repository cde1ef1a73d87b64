// An encode that allocates nothing: jpg_encode_device_static (include/jpeg_hip.h).  The working memory the
// kernels need beyond the caller's rows and the caller's output -- the coded tiles, their lengths, the
// hand-off words and ticket counters of the stuffing kernel, the status record -- lives in one file-scope
// device array instead of buffers owned by a context.  Every GPU the library runs on holds its own copy of
// the array at its own address from the moment the code object is loaded until the process ends.
#include "../../include/jpeg_hip.h"

#include <atomic>
#include <cstddef>

#include "internal.hpp"

using namespace jpg;

static_assert(JPG_STATIC_SCRATCH_BYTES % 256 == 0, "whole 256-byte units");
static_assert(sizeof(jpg_static_result) == sizeof(DevStatus) && offsetof(jpg_static_result, error) == offsetof(DevStatus, error),
              "the result record is the kernels' status record");
static __device__ __align__(256) uint32_t scratch[JPG_STATIC_SCRATCH_BYTES / sizeof(uint32_t)];

// Address of the array on each device.  In host code the array's name only stands for the symbol: its
// memory is reached through the address hipGetSymbolAddress reports for the current device.
static constexpr int MAX_DEVICES = 64;
static std::atomic<uint32_t *> scratch_ptr[MAX_DEVICES] = {};

namespace {

struct DeviceScope {   // makes `device` current and puts the caller's device back
    int prev = -1;
    bool ok = false;
    explicit DeviceScope(int device) {
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); cur = -1; }
        if (cur == device) { ok = true; return; }
        if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return; }
        prev = cur;
        ok = true;
    }
    ~DeviceScope() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

// with `device` current
int scratch_of(int device, uint32_t **out) {
    if (device < 0 || device >= MAX_DEVICES) return JPG_ERR_BAD_ARG;
    uint32_t *p = scratch_ptr[device].load(std::memory_order_acquire);
    if (!p) {
        void *q = nullptr;
        if (hipGetSymbolAddress(&q, HIP_SYMBOL(scratch)) != hipSuccess || !q) {
            (void)hipGetLastError();
            return JPG_ERR_HIP;
        }
        p = static_cast<uint32_t *>(q);
        scratch_ptr[device].store(p, std::memory_order_release);
    }
    *out = p;
    return JPG_OK;
}

size_t up256(size_t n) { return (n + 255) / 256 * 256; }

// Where everything sits in the array for an image of geometry g (byte offsets, all multiples of 256).  The
// small records come first so that one memset clears what must start at zero.
struct Layout {
    size_t status, ticket, agg, grp, hyp, clen, clear_bytes, gsum, sparse, total;
    long long nchunks, cstride;
    int slot_words;
    size_t nseg;
};

Layout layout_of(const Geom &g) {
    Layout l{};
    l.nchunks = coded_tiles(g);
    l.cstride = (l.nchunks + 63) / 64 * 64;
    l.slot_words = coded_slot_words(g);
    l.nseg = worst_segments((size_t)g.nblocks);
    size_t at = 0;
    l.status = at; at += up256(sizeof(DevStatus));
    l.ticket = at; at += up256(TICKET_WORDS_SINGLE * sizeof(uint32_t));
    l.agg = at; at += up256(l.nseg * sizeof(uint64_t));
    l.grp = at; at += up256((l.nseg / 64 + 1) * sizeof(uint64_t));
    l.hyp = at; at += up256((l.nseg / 4096 + 1) * sizeof(uint64_t));
    l.clen = at; at += up256((size_t)l.cstride * sizeof(uint32_t));
    l.clear_bytes = at;
    l.gsum = at; at += up256((size_t)(l.cstride / 64) * sizeof(uint32_t));
    l.sparse = at; at += (size_t)l.cstride * (size_t)l.slot_words * sizeof(uint32_t);
    l.total = at;
    return l;
}

int geometry(Geom &g, int32_t W, int32_t H, uint32_t stride, int bottom_up, int ra, int rb, int rc) {
    int h = 1, v = 1;
    if (!ratio_factors(ra, rb, rc, h, v)) return JPG_ERR_INVALID_RATIO;
    if (int e = make_geom(g, W, H, stride, bottom_up, h, v)) return e;
    if (!coded_supported(g)) return JPG_ERR_BAD_ARG;               // narrower than a tile of the coding kernel
    return JPG_OK;
}

}  // namespace

extern "C" int jpg_static_init(int device) {
    DeviceScope ds(device);
    if (!ds.ok) return JPG_ERR_HIP;
    uint32_t *p = nullptr;
    return scratch_of(device, &p);
}

extern "C" size_t jpg_static_scan_capacity(int32_t W, int32_t H, int ra, int rb, int rc) {
    Geom g{};
    if (W <= 0 || geometry(g, W, H, (uint32_t)((3 * (int64_t)W + 3) / 4 * 4), 1, ra, rb, rc) != JPG_OK) return 0;
    return worst_scan_bytes((size_t)g.nblocks);
}

extern "C" int jpg_encode_device_static(int device, int dct_algo, int ra, int rb, int rc, const uint8_t *dev_pixels,
                                        int32_t W, int32_t H, uint32_t stride, int bottom_up, uint8_t *dev_scan,
                                        size_t scan_cap, jpg_static_result *dev_result, void *stream) {
    if (!dev_pixels || !dev_scan || !dev_result) return JPG_ERR_BAD_ARG;
    if (dct_algo != JPG_DCT_REAL && dct_algo != JPG_DCT_BIN && dct_algo != JPG_DCT_REAL_STRICT) return JPG_ERR_BAD_ARG;
    if ((stride & 3) || (reinterpret_cast<uintptr_t>(dev_pixels) & 3)) return JPG_ERR_BAD_ARG;   // rows are read as dwords
    Geom g{};
    if (int e = geometry(g, W, H, stride, bottom_up, ra, rb, rc)) return e;
    const Layout l = layout_of(g);
    if (l.total > (size_t)JPG_STATIC_SCRATCH_BYTES) return JPG_ERR_TOO_LARGE;
    if (scan_cap < worst_scan_bytes((size_t)g.nblocks)) return JPG_ERR_CAPACITY;
    DeviceScope ds(device);
    if (!ds.ok) return JPG_ERR_HIP;
    uint32_t *base = nullptr;
    if (int e = scratch_of(device, &base)) return e;
    uint8_t *b8 = reinterpret_cast<uint8_t *>(base);
    hipStream_t s = static_cast<hipStream_t>(stream);

    // The array keeps what the previous call left: status, ticket counters, hand-off words and tile lengths
    // start from zero on every call (the coded tiles themselves are overwritten as far as they are read).
    if (hipMemsetAsync(b8, 0, l.clear_bytes, s) != hipSuccess) { (void)hipGetLastError(); return JPG_ERR_HIP; }
    EntropyBufs eb{};
    eb.nimg = 1;
    eb.nchunks = l.nchunks;
    eb.cstride = l.cstride;
    eb.slot_words = l.slot_words;
    eb.clen = reinterpret_cast<uint32_t *>(b8 + l.clen);
    eb.gsum = reinterpret_cast<uint32_t *>(b8 + l.gsum);
    eb.sparse = reinterpret_cast<uint32_t *>(b8 + l.sparse);
    eb.epoch = 1;
    eb.status = reinterpret_cast<DevStatus *>(b8 + l.status);
    eb.host_status = reinterpret_cast<DevStatus *>(dev_result);   // the last kernel copies the record there
    eb.sagg = reinterpret_cast<uint64_t *>(b8 + l.agg);
    eb.sgrp = reinterpret_cast<uint64_t *>(b8 + l.grp);
    eb.shyp = reinterpret_cast<uint64_t *>(b8 + l.hyp);
    eb.stuff_epoch = 1;
    eb.sticket = reinterpret_cast<uint32_t *>(b8 + l.ticket);
    eb.scan = dev_scan;
    eb.scan_stride = scan_cap;
    eb.nseg_cap = l.nseg;
    const int16_t zero[3] = {0, 0, 0};   // src/jpeg/entropy_coding.rs:38-40
    hipError_t e = launch_coded(g, dct_algo, dev_pixels, nullptr, 1, zero, eb, s);
    if (e == hipSuccess) e = launch_stuff(eb, 0, 0, 1, s);
    if (e != hipSuccess) { (void)hipGetLastError(); return JPG_ERR_HIP; }
    return JPG_OK;
}

extern "C" int jpg_static_status(const jpg_static_result *r) {
    if (!r) return JPG_ERR_BAD_ARG;
    const int v = -r->error;
    return v > 0 && (v >> 8) == 1 ? -(v & 0xff) : JPG_OK;
}
