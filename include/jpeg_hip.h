/*
 * jpeg_hip.h -- C ABI of the MI355X (gfx950) JPEG encode path.
 *
 * The reference (uriGrif/jpeg-encoder) has no FFI of its own: its API is the
 * five `JpegImage` methods that `main` calls in order (src/main.rs:17,35,48,
 * 61,66).  Each entry point below replaces one of them (cited per function)
 * with plain pointers and sizes, so a Rust/C/C++/Python host can bind it.
 *
 * Threading: a context owns its device buffers and its HIP streams.  A context
 * is thread-compatible (one thread at a time), not thread-safe.  The library
 * keeps no global mutable state apart from per-device caches of device
 * properties (atomics).  Every entry point makes the context's device current
 * for its own duration and restores the caller's current device on return.
 *
 * Errors: every function returns 0 (JPG_OK) or a negative jpg_status.  The
 * library never aborts; jpg_strerror() gives the reference's panic text where
 * one exists.
 */
#ifndef JPEG_HIP_H
#define JPEG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum jpg_status {
    JPG_OK = 0,
    JPG_ERR_INVALID_RATIO = -1, /* "Invalid chrominance downsampling ratio!"     src/jpeg/sampling.rs:19 */
    JPG_ERR_DC_RANGE = -2,      /* "DC coefficient bit length greater than 11!"  src/jpeg/entropy_coding.rs:154 */
    JPG_ERR_AC_RANGE = -3,      /* "AC coefficient bit length greater than 10!"  src/jpeg/entropy_coding.rs:190 */
    JPG_ERR_HUFF_DC = -4,       /* "DC Huffman Code Not Found!"                  src/jpeg/entropy_coding.rs:212 */
    JPG_ERR_HUFF_AC = -5,       /* "AC Huffman Code Not Found!"                  src/jpeg/entropy_coding.rs:221 */
    JPG_ERR_BAD_ARG = -6,
    JPG_ERR_HIP = -7,           /* a HIP runtime call failed; see jpg_last_hip_error */
    JPG_ERR_STATE = -8,         /* stage called out of order */
    JPG_ERR_UNSUPPORTED_DIMS = -9, /* width mod 16 == 1 with h=2 or height mod 16 == 1 with v=2: the
                                      reference's chroma plane is skewed there (src/jpeg/sampling.rs:56-101
                                      pushes (Wp/h)*(Hp/v) samples into a plane declared Wc x Hc) */
    JPG_ERR_NOMEM = -10,
    JPG_ERR_CAPACITY = -11,     /* caller buffer too small */
    JPG_ERR_TOO_LARGE = -12,    /* more blocks than the call's kernels index (see the call) */
    JPG_ERR_CHAIN = -13         /* stripe chain: another stripe's values did not arrive in time, or the gather buffer
                                   is too small for the scan */
} jpg_status;

/* src/jpeg/dct_quant.rs:8-12.  JPG_DCT_REAL is the separable f32 transform
 * (same cosine table and the same final scale/divide/truncate expression as
 * the reference; quantised coefficients may differ by +-1 at truncation
 * boundaries).  JPG_DCT_REAL_STRICT evaluates the reference's 64-term sum in
 * its own order without FMA and is bit-exact. */
typedef enum jpg_dct {
    JPG_DCT_REAL = 0,
    JPG_DCT_BIN = 1,
    JPG_DCT_REAL_STRICT = 2
} jpg_dct;

typedef struct jpg_ctx jpg_ctx;

const char *jpg_strerror(int status);
/* hipError_t of the last failing HIP call made through this context (0 if none) */
int jpg_last_hip_error(const jpg_ctx *ctx);

/* JpegImage::new (src/jpeg/jpeg_image.rs:24-102): ratio_a:b:c is the
 * --subsampling-ratio triple (4:4:4, 4:2:2, 4:2:0; anything else ->
 * JPG_ERR_INVALID_RATIO).  Buffers are sized at the first load. */
int jpg_create(jpg_ctx **out, int device, int dct_algo, int ratio_a, int ratio_b, int ratio_c);
void jpg_destroy(jpg_ctx *ctx);
/* the context's HIP stream (hipStream_t), for callers that time or order work on it */
void *jpg_stream(jpg_ctx *ctx);
/* block until everything queued on the context's stream has finished */
int jpg_sync(jpg_ctx *ctx);

/* ---- staged path: one call per reference method ------------------------ */

/* JpegImage::from_bmp minus the file parsing (src/jpeg/jpeg_image.rs:104-137,
 * src/bmp/bmp_image.rs:57-90): `pixels` are BMP rows, 3 bytes B,G,R per
 * pixel, `row_stride` bytes apart, last image row first when bottom_up != 0.
 * Host memory; copied to the device before return.  Produces the three u8
 * planes (padded to the MCU size with zeros). */
int jpg_load_bgr(jpg_ctx *ctx, const uint8_t *pixels, int32_t width, int32_t height,
                 uint32_t row_stride, int bottom_up);
/* same with `pixels` already in device memory (not copied; must stay valid
 * until the context has been synchronised) */
int jpg_load_bgr_device(jpg_ctx *ctx, const uint8_t *dev_pixels, int32_t width, int32_t height,
                        uint32_t row_stride, int bottom_up);
/* JpegImage::chrominance_downsampling (src/jpeg/sampling.rs:46-102); no-op for 4:4:4 */
int jpg_downsample(jpg_ctx *ctx);
/* JpegImage::dct_and_quantization (src/jpeg/dct_quant.rs:15-61) */
int jpg_dct_quant(jpg_ctx *ctx);
/* JpegImage::generate_entropy_encoded_data (src/jpeg/entropy_coding.rs:16-125):
 * produces the MSB-first bit string (= entropy_coded_bits.as_raw_slice(), last
 * byte zero-padded, no 0xFF stuffing) in device memory; *nbits = its length */
int jpg_entropy(jpg_ctx *ctx, uint64_t *nbits);
/* the stuffing loop of write_image_data (src/jpeg/file.rs:92-103), on the
 * device: 0x00 after every 0xFF byte; *len = stuffed byte count */
int jpg_stuff(jpg_ctx *ctx, size_t *len);

/* ---- fused path --------------------------------------------------------- */

/* Colour conversion + subsampling + DCT + quantisation as ONE kernel from
 * device-resident BMP rows straight to quantised coefficients (the u8 planes
 * are never written to memory).  Asynchronous: returns after queueing. */
int jpg_front_device(jpg_ctx *ctx, const uint8_t *dev_pixels, int32_t width, int32_t height,
                     uint32_t row_stride, int bottom_up);
/* the same from host memory: the rows are copied to the device first (the copy
 * is complete when the call returns, so `pixels` may be reused).  For images at
 * least one tile wide the transform itself is deferred to the call that tells
 * what is wanted: jpg_get_coeffs runs the kernel above, jpg_entropy /
 * jpg_entropy_stripe code the rows in one kernel without writing coefficients,
 * jpg_last_dc transforms the last MCU only. */
int jpg_front_host(jpg_ctx *ctx, const uint8_t *pixels, int32_t width, int32_t height, uint32_t row_stride,
                   int bottom_up);
/* A whole encode from device-resident rows to the stuffed scan, queued without any host
 * synchronisation; jpg_finish() waits and reports.  Images at least one tile (512 pixels, 1024 at 4:2:2) wide go
 * through ONE kernel from pixels to Huffman-coded bits (no coefficient ever reaches memory) plus the
 * stuffing kernel; narrower ones through the front kernel, the entropy kernel and the stuffing
 * kernel (also for every image when the environment variable JPG_TWO_KERNELS=1 is set at
 * jpg_create).  The bytes are the same either way.  After the single-kernel path jpg_get_coeffs,
 * jpg_last_dc and jpg_entropy compute the coefficients on request when the context holds its own
 * copy of the rows (host input), else they return JPG_ERR_STATE: call jpg_front_device for them. */
int jpg_encode_device_async(jpg_ctx *ctx, const uint8_t *dev_pixels, int32_t width, int32_t height,
                            uint32_t row_stride, int bottom_up);
/* ---- batches (no counterpart in the reference, which encodes one file per run) ----
 * n (1..jpg_max_batch()) device-resident images of the same size, row stride and orientation go
 * through ONE launch of each kernel, which removes the per-image launch and latency costs that
 * dominate small images.  dev_pixels is a host array of n device pointers (rows dword-aligned).
 * Queued like jpg_encode_device_async and retired with jpg_finish / jpg_finish_next, after which
 * jpg_batch_result gives image i's bit count, stuffed length and the device address of its scan
 * (valid as long as the results of that encode are), and jpg_get_batch_scan copies it out. */
int jpg_max_batch(void);
int jpg_encode_batch_device_async(jpg_ctx *ctx, const uint8_t *const *dev_pixels, int32_t n, int32_t width, int32_t height,
                                  uint32_t row_stride, int bottom_up);
int jpg_batch_result(jpg_ctx *ctx, int32_t i, uint64_t *nbits, size_t *stuffed_len, const uint8_t **scan_dev);
int jpg_get_batch_scan(jpg_ctx *ctx, int32_t i, uint8_t *out, size_t cap);

/* Measurement aid: with every = n > 0, every n-th launch of the front kernel or of the fused coding
 * kernel (jpg_front_device/_host, jpg_encode_*) hands two events to the dispatch, which receive
 * the kernel's own start and end time stamps (0 switches it off); jpg_front_time waits for
 * the stream and returns the summed duration of those launches in milliseconds, their
 * number and the number of images they covered (a batch is one launch).  No counterpart in
 * the reference. */
int jpg_set_timing(jpg_ctx *ctx, int every);
int jpg_front_time(jpg_ctx *ctx, double *total_ms, uint64_t *launches, uint64_t *images);
/* the individual durations (milliseconds) of the first up to 4096 of those launches, in launch order: *n of
 * them are written to ms[0..cap) */
int jpg_front_samples(jpg_ctx *ctx, float *ms, int32_t cap, int32_t *n);
/* Page-locked host memory (release with jpg_free).  Rows handed to jpg_load_bgr,
 * jpg_front_host or jpg_encode_scan from such memory are copied by DMA at full
 * link speed; ordinary (pageable) memory is staged through the library's own
 * page-locked buffers.  No counterpart in the reference. */
void *jpg_host_alloc(size_t bytes);
/* wait for the queued encode; nbits/stuffed_len may be NULL.  Returns the
 * reference's range errors (JPG_ERR_DC_RANGE/JPG_ERR_AC_RANGE) if a kernel
 * met a coefficient the Huffman tables cannot code. */
int jpg_finish(jpg_ctx *ctx, uint64_t *nbits, size_t *stuffed_len);
/* Up to three fused encodes may be in flight on a context: the byte stuffing of one image runs
 * on a second stream while the front and entropy kernels of the next image run on the first
 * (three sets of output buffers).  jpg_finish_next waits for the OLDEST queued encode only and
 * reports it; its scan stays readable (jpg_scan_device, jpg_get_scan) until three more encodes
 * have been queued.  Queuing a fourth encode before taking the first overwrites the first's
 * result.  jpg_finish waits for everything and reports the newest.  No counterpart in the
 * reference. */
int jpg_finish_next(jpg_ctx *ctx, uint64_t *nbits, size_t *stuffed_len);
/* everything above from host pixels to a host copy of the stuffed scan bytes
 * (library-allocated pinned memory; release with jpg_free) */
int jpg_encode_scan(jpg_ctx *ctx, const uint8_t *pixels, int32_t width, int32_t height,
                    uint32_t row_stride, int bottom_up, uint8_t **stuffed, size_t *len);
/* The same in two steps, so that the host can do its own work (emit the file header: generate_file,
 * src/jpeg/file.rs:105-133, writes 623 bytes before the scan) while the device works.  _begin queues the encode
 * and returns: where the single coding kernel applies, the rows are copied and coded band by band (a band of MCU
 * rows is coded while the following bands are still on their way to the device), and the output buffer is
 * allocated meanwhile.  `pixels` must stay valid and unchanged until _end returns.  _end waits, copies the
 * stuffed scan to the host and hands the buffer over (jpg_free).  One begun encode per context at a time. */
int jpg_encode_scan_begin(jpg_ctx *ctx, const uint8_t *pixels, int32_t width, int32_t height, uint32_t row_stride,
                          int bottom_up);
int jpg_encode_scan_end(jpg_ctx *ctx, uint8_t **stuffed, size_t *len);
void jpg_free(void *p);

/* ---- MCU-row stripes of one large image (no counterpart in the reference,
 * which codes the whole image serially; the arithmetic is that of
 * src/jpeg/entropy_coding.rs:38-40,149-150 carried across the cut) -------- */

/* Entropy-code the coefficients currently in the context as a stripe whose
 * three DC predictors start at pred[0..2] (Y, Cb, Cr) instead of 0.  The bit
 * string starts at bit 0 of the context's buffer. */
int jpg_entropy_stripe(jpg_ctx *ctx, const int16_t pred[3], uint64_t *nbits);
/* The last min(32, nbits) bits of the stripe's unstuffed bit string, right-aligned
 * in *last32: what a following stripe needs to fill the byte the two share. */
int jpg_stream_tail(jpg_ctx *ctx, uint32_t *last32);
/* DC coefficient of the last Y, Cb, Cr block of the stripe in scan order */
int jpg_last_dc(jpg_ctx *ctx, int16_t dc[3]);
/* Place this stripe's bit string at global bit offset `bit_offset` of the
 * whole scan: shift right by bit_offset % 8, OR `head_bits` (the bits of the
 * previous stripe that share this stripe's first byte, already in place in
 * the top bits) into the first byte, then apply 0xFF stuffing.  When
 * !is_last the trailing partial byte is dropped (the next stripe owns it).
 * *len = stuffed byte count of the piece; *tail = this stripe's bits of the
 * dropped partial byte (top-aligned), 0 when byte-aligned or is_last. */
int jpg_stitch(jpg_ctx *ctx, uint64_t bit_offset, uint8_t head_bits, int is_last, size_t *len,
               uint8_t *tail);

/* A stripe from device-resident rows the caller keeps valid until the stripe has been coded (jpg_entropy_stripe
 * or jpg_entropy_stripe_finish has returned).  Where the single coding kernel applies (see
 * jpg_encode_device_async) nothing is computed by this call: jpg_last_dc transforms the stripe's last MCU only
 * and the coding call goes from the rows to the coded bits in one kernel; jpg_get_coeffs still works. */
int jpg_stripe_device(jpg_ctx *ctx, const uint8_t *dev_pixels, int32_t width, int32_t height, uint32_t row_stride,
                      int bottom_up);
/* jpg_front_host without the wait for the copy: `pixels` (host memory) must stay valid and unchanged until a
 * later call on the context has waited (jpg_last_dc, jpg_entropy_stripe*, jpg_sync).  Lets the host-to-device
 * copies of the stripes of several contexts run at the same time. */
int jpg_stripe_host_async(jpg_ctx *ctx, const uint8_t *pixels, int32_t width, int32_t height, uint32_t row_stride,
                          int bottom_up);
/* The three steps of a stripe without waiting in between, so that one host thread can keep the stripes of
 * several contexts (one per GPU) running at the same time: *_async queues on the context's stream, the
 * matching call waits for it and reports.
 *   jpg_last_dc_async           -> jpg_last_dc
 *   jpg_entropy_stripe_async    -> jpg_entropy_stripe_finish (also the stripe's last bits: jpg_stream_tail)
 *   jpg_stitch_async            -> jpg_stitch_finish */
int jpg_last_dc_async(jpg_ctx *ctx);
int jpg_entropy_stripe_async(jpg_ctx *ctx, const int16_t pred[3]);
int jpg_entropy_stripe_finish(jpg_ctx *ctx, uint64_t *nbits, uint32_t *last32);
int jpg_stitch_async(jpg_ctx *ctx, uint64_t bit_offset, uint8_t head_bits, int is_last);
int jpg_stitch_finish(jpg_ctx *ctx, size_t *len, uint8_t *tail);
/* Gathering the pieces: copies the context's stuffed piece (after jpg_stitch) to `dst` in device memory,
 * queued on the context's stream (jpg_sync waits).  dst_device >= 0 and different from the context's device:
 * a peer copy between the two GPUs (hipMemcpyPeerAsync, over xGMI where the GPUs are linked); dst_device < 0 or
 * the same device: `dst` is any address this process can write by unified addressing, e.g. a buffer of
 * another process opened with jpg_ipc_open, or page-locked host memory.  cap = bytes available at dst. */
int jpg_scan_copy_to_device(jpg_ctx *ctx, void *dst, int dst_device, size_t cap);
/* Plain device memory for the gathered scan, for hosts that do not link the HIP runtime themselves, a blocking
 * copy of it to the host (waits for everything queued on the device first), and its hand-over between
 * processes (one process per GPU): the owner exports a handle, the others open it on their own device and
 * copy their pieces in.  No counterpart in the reference. */
void *jpg_device_alloc(int device, size_t bytes);
void jpg_device_free(int device, void *p);
int jpg_device_to_host(int device, void *dst, const void *src_dev, size_t bytes);
#define JPG_IPC_HANDLE_BYTES 64
int jpg_ipc_export(const void *dev_ptr, uint8_t handle[JPG_IPC_HANDLE_BYTES]);
int jpg_ipc_open(int device, const uint8_t handle[JPG_IPC_HANDLE_BYTES], void **dev_ptr);
int jpg_ipc_close(void *dev_ptr);

/* ---- stripe chain: the stripes of one image with every hand-over done by kernels (no counterpart in the
 * reference; the arithmetic carried is that of src/jpeg/entropy_coding.rs:38-40,149-150, the stuffing after
 * the shift that of src/jpeg/file.rs:92-103).
 *
 * The calls above leave the host between the steps of a stripe (three DC values, a bit count and a few bits, a
 * stuffed length: a wait and an exchange each).  In a chain the kernels of the ranks publish those values
 * themselves into a MAILBOX in page-locked host memory that every GPU of the node maps, small one-wave "gate"
 * kernels queued between the heavy kernels pick them up, and a copy kernel writes the stuffed piece to its byte
 * offset in the GATHER buffer; the host queues a whole pass at once and waits once.  Every gate wait is bounded
 * (JPG_CHAIN_TIMEOUT_MS, default 10000) and ends in JPG_ERR_CHAIN, never in a hang.
 *
 *  - mailbox: jpg_chain_mailbox_alloc(): jpg_chain_mailbox_bytes() zeroed bytes of page-locked host memory the
 *    ranks share (release with jpg_free).  One process driving several GPUs passes the pointer to every
 *    context; with one process per GPU rank 0 allocates and exports it (jpg_ipc_export works on page-locked host
 *    memory of jpg_host_alloc / jpg_chain_mailbox_alloc as on device memory) and the others jpg_ipc_open it.
 *  - gather: where the whole stuffed scan is put together: device memory of the collector rank 0 (the other
 *    ranks pass the address they reach it by: the same pointer within a process once jpg_peer_access(device,
 *    owner) returned 1, the result of jpg_ipc_open between processes), or, where GPUs cannot reach each other's
 *    memory, page-locked host memory.  gather_cap: its size.  The buffer has two halves of gather_cap / 2 bytes:
 *    passes with an even number (counted from jpg_chain_setup) put their scan at the start of the first half,
 *    those with an odd number at the start of the second; a scan that does not fit its half fails the pass.
 *  - jpg_chain_setup binds both to the context and restarts the pass counter: every rank calls it before the
 *    first pass, with its rank (0 .. world-1, world <= 64), and all ranks run the same number of passes.  Rank 0's
 *    call clears the mailbox (through the host address it was given): no rank queues a pass before that call
 *    has returned, and none is still running one of an earlier set-up.
 *  - jpg_chain_pass_async: after jpg_stripe_device / jpg_stripe_host_async has named the stripe's rows (images
 *    at least one tile wide: JPG_ERR_STATE otherwise), queues the whole pass on the context's stream and returns.  active_mask: bit k set when rank k has a stripe (ranks without rows make no call; rank 0 always
 *    has one: JPG_ERR_BAD_ARG otherwise).
 *    One process driving several contexts queues the ranks in ascending order, all of them before it waits for
 *    any: a gate only waits for lower ranks, so nothing queued can wait for something behind it in a hardware queue.
 *    Two passes may be queued before the first is taken; a third is refused with JPG_ERR_STATE.  Each pass in
 *    flight has its own hand-over record in the mailbox (four records, used in turn: a rank that queues pass p
 *    has taken pass p-2, which the ranks below it had coded after rank 0 took pass p-4, so nobody reads the record
 *    any more), its own half of the gather buffer and its own set of the context's working buffers.  The rows of a
 *    stripe stay the caller's until its pass is taken.  Rank 0 has read, or given up, the scan of pass p-2 when it
 *    queues pass p: that pass fills the same half.
 *  - jpg_chain_pass_finish waits for the context's oldest queued pass: the stripe's bit count, the length of its stuffed
 *    piece, where its bits start in the scan and its piece in the gather buffer (the half's offset included), and
 *    on rank 0 the length of the whole scan (0 elsewhere): the last
 *    rank with a stripe publishes it in the mailbox when every piece lies in the gather buffer, and rank 0's call
 *    waits for that word on the host -- rank 0 therefore passes a mailbox address its host can read (its own
 *    allocation).  Any of the pointers may be NULL. */
size_t jpg_chain_mailbox_bytes(void);
void *jpg_chain_mailbox_alloc(void);
/* 1 when kernels on `device` can write memory of `peer` (peer access is switched on by the call), else 0 */
int jpg_peer_access(int device, int peer);
int jpg_chain_setup(jpg_ctx *ctx, int rank, int world, void *mailbox, void *gather, size_t gather_cap);
int jpg_chain_pass_async(jpg_ctx *ctx, uint64_t active_mask);
int jpg_chain_pass_finish(jpg_ctx *ctx, uint64_t *nbits, size_t *piece_len, uint64_t *bit_offset, size_t *byte_offset,
                          size_t *total);

/* ---- an encode that allocates nothing ---------------------------------------
 *
 * No counterpart in the reference.  For callers that cannot allocate device memory while they run: code under
 * HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  jpg_encode_device_static is
 * jpg_encode_device_async + jpg_finish without a context: the same two kernels (rows -> coded tiles, coded
 * tiles -> stuffed scan) are queued on the caller's stream, and the working memory a context would allocate
 * (coded tiles, their lengths, the stuffing kernel's hand-off words and counters, the status record) is a
 * fixed array of JPG_STATIC_SCRATCH_BYTES inside the library's code object, one copy per GPU, held until the
 * process ends.  The call makes no hipMalloc/hipFree and creates no stream or event; it only queues a memset
 * and two kernels and returns without waiting.
 *
 *  - dev_pixels: device-resident BMP rows, 4-byte aligned base and stride; width at least one tile of the
 *    coding kernel (512 pixels, 1024 at 4:2:2), else JPG_ERR_BAD_ARG.
 *  - limit: images whose working memory exceeds the array are refused with JPG_ERR_TOO_LARGE.  Fits e.g.
 *    1920x1080 at every ratio and 2048x2048 at 4:2:0.
 *  - dev_scan / scan_cap: the caller's output buffer in device memory; scan_cap must be at least
 *    jpg_static_scan_capacity(width, height, ratio) (the worst case: every byte stuffed), else JPG_ERR_CAPACITY.
 *  - dev_result: 32 bytes the device can write (device memory or page-locked host memory); filled by the last
 *    kernel.  After the stream has been waited for, jpg_static_status() of a host copy gives JPG_OK or the
 *    status a kernel raised (the reference's range panics), and nbits / scan_bytes the sizes.
 *  - stream: a hipStream_t of `device` (NULL = the default stream).
 *  - all calls on one GPU share that GPU's array: two calls whose kernels may overlap on one GPU (different
 *    streams, or different processes' contexts are fine -- each process has its own copy) race.  Queue them
 *    on one stream, or order them with events.
 * jpg_static_init(device) looks up the array's address on that device ahead of the first call (otherwise the
 * first call on a device does it). */
#define JPG_STATIC_SCRATCH_BYTES (24u << 20)
typedef struct jpg_static_result {
    uint64_t nbits;      /* bits of the unstuffed stream */
    uint64_t scan_bytes; /* bytes written to dev_scan */
    int32_t error;       /* raw status word: decode with jpg_static_status */
    uint32_t reserved[3];
} jpg_static_result;
int jpg_static_init(int device);
size_t jpg_static_scan_capacity(int32_t width, int32_t height, int ra, int rb, int rc);
int jpg_encode_device_static(int device, int dct_algo, int ra, int rb, int rc, const uint8_t *dev_pixels, int32_t width,
                             int32_t height, uint32_t row_stride, int bottom_up, uint8_t *dev_scan, size_t scan_cap,
                             jpg_static_result *dev_result, void *stream);
int jpg_static_status(const jpg_static_result *host_copy);

/* ---- results ------------------------------------------------------------ */

/* device pointers of the current results (valid until the next call that
 * re-encodes or destroys the context) */
const uint8_t *jpg_bits_device(jpg_ctx *ctx);
const uint8_t *jpg_scan_device(jpg_ctx *ctx);
/* copies to host memory; cap = capacity of `out` in bytes.  The bit string is assembled on the first request
 * (jpg_bits_device / jpg_get_bits); JPG_ERR_STATE while queued encodes have not been taken with jpg_finish*. */
int jpg_get_bits(jpg_ctx *ctx, uint8_t *out, size_t cap);
int jpg_get_scan(jpg_ctx *ctx, uint8_t *out, size_t cap);
/* public fields y_channel..cr_dct_coeffs of JpegImage (src/jpeg/jpeg_image.rs:
 * 14-19) in the reference's row-major plane layout; comp 0 Y, 1 Cb, 2 Cr.
 * Planes exist on the staged path only; coefficients on both. */
int jpg_plane_dims(jpg_ctx *ctx, int comp, int32_t *w, int32_t *h);
int jpg_coeff_dims(jpg_ctx *ctx, int comp, int32_t *w, int32_t *h);
int jpg_get_plane(jpg_ctx *ctx, int comp, uint8_t *out, size_t cap);
int jpg_get_coeffs(jpg_ctx *ctx, int comp, int16_t *out, size_t cap);

/* ---- introspection for benchmarks --------------------------------------- */

/* algorithmic bytes one fused front launch moves for the current image:
 * row_stride*H read + 2 bytes per quantised coefficient written */
uint64_t jpg_front_bytes(const jpg_ctx *ctx);
/* number of 8x8 blocks (all components) of the current image */
uint64_t jpg_num_blocks(const jpg_ctx *ctx);
/* 1 when whole encodes of the current image size run the single fused coding kernel (which reads
 * row_stride*H bytes and writes only the coded bits), 0 when they run front + entropy kernels.
 * No counterpart in the reference. */
int jpg_fused_coding(const jpg_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
