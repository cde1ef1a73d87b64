// Stripe chain: the kernels that carry the scalars linking the MCU-row stripes of one image from GPU to GPU.
//
// The reference codes an image serially (src/jpeg/entropy_coding.rs:16-125); cut between MCU rows, a stripe needs
// from the stripes above it the three DC predictors (src/jpeg/entropy_coding.rs:38-40,149-150), their bit counts,
// the bits that share its first byte, and -- because 0xFF stuffing (src/jpeg/file.rs:92-103) follows the shift --
// their stuffed lengths.  Each value is published by a kernel of the rank that computed it into a mailbox in
// page-locked host memory that all GPUs of the node map, and picked up by a small "gate" kernel queued in front of
// the kernel that needs it: a gate is one wave that polls, leaves the values in the context's ChainParams and ends,
// so the heavy kernels never wait themselves and a waiting rank holds one wave slot, not a grid.  Every wait is
// bounded (wall clock) and ends in an error, never in a hang.
#include "chain.hpp"

namespace jpg {

namespace {

// lane k looks after rank k
__global__ __launch_bounds__(64) void k_chain_place(ChainPass c) {
    __shared__ uint64_t sBits[CHAIN_MAX_RANKS];
    __shared__ uint32_t sTail[CHAIN_MAX_RANKS];
    const int lane = threadIdx.x;
    const ChainMail *box = c.mail + c.slot * CHAIN_MAX_RANKS;
    const bool need = lane < c.rank && ((c.active >> lane) & 1);
    bool timeout = false;
    sBits[lane] = mail_wait(&box[lane].bits, need, c.tag, c.timeout_ticks, timeout);
    sTail[lane] = (uint32_t)mail_wait(&box[lane].tail, need, c.tag, c.timeout_ticks, timeout);
    __syncthreads();
    if (lane == 0) {
        // where this stripe's bits start, and the last 32 bits of everything before it (stripes may be a few bits long)
        uint64_t total = 0;
        uint32_t last = 0;
        for (int k = 0; k < c.rank; k++) {
            const uint64_t n = sBits[k];
            if (!n) continue;
            last = n >= 32 ? sTail[k] : ((last << n) | sTail[k]);
            total += n;
        }
        const uint32_t r = (uint32_t)(total & 7);
        c.params->bit_offset = total;
        c.params->shift = r;
        c.params->head = r ? ((last & ((1u << r) - 1u)) << (8 - r)) & 0xffu : 0u;
    }
    if (__ballot(timeout) && lane == 0) chain_fail(c);
}

__global__ __launch_bounds__(64) void k_chain_offset(ChainPass c, unsigned long long gather_cap) {
    __shared__ uint64_t sLen[CHAIN_MAX_RANKS];
    const int lane = threadIdx.x;
    ChainMail *box = c.mail + c.slot * CHAIN_MAX_RANKS;
    const uint64_t own = c.status->scan_bytes;
    if (lane == 0) mail_store(&box[c.rank].len, mail_pack(c.tag, own));
    const bool need = lane < c.rank && ((c.active >> lane) & 1);
    bool timeout = false;
    sLen[lane] = mail_wait(&box[lane].len, need, c.tag, c.timeout_ticks, timeout);
    const bool timed_out = __ballot(timeout) != 0;
    __syncthreads();
    if (lane == 0) {
        uint64_t off = 0;
        for (int k = 0; k < c.rank; k++) off += sLen[k];
        const bool bad = timed_out || c.params->failed || off + own > gather_cap;
        c.params->byte_offset = off;
        c.params->len = bad ? 0 : own;
        if (bad) chain_fail(c);
    }
}

// dst may have any alignment (it follows the pieces before it): 16 bytes per thread and round on its aligned part, each
// put together from the source dwords around them; single bytes at both ends.  The stores are system-scope
// (write-through: the destination may be another GPU's memory or host memory), so that the piece is in place when the
// kernel has ended and the next kernel in the stream says so; no fence inside the kernel (a system-scope fence writes
// the whole L2 back, once per wave).
__device__ __forceinline__ void store_through(uint64_t *p, uint64_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(256) void k_chain_copy(const ChainParams *params, const uint8_t *__restrict__ src,
                                                     uint8_t *__restrict__ gather) {
    const uint64_t len = params->len;
    if (!len) return;
    uint8_t *dst = gather + params->byte_offset;
    const uint64_t head = min((uint64_t)((16 - (reinterpret_cast<uintptr_t>(dst) & 15)) & 15), len);
    const uint64_t body = (len - head) / 16;                     // 16-byte pieces
    const uint64_t tail0 = head + body * 16;
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (uint64_t)gridDim.x * blockDim.x;
    if (t < head) __hip_atomic_store(&dst[t], src[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (t < len - tail0) __hip_atomic_store(&dst[tail0 + t], src[tail0 + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // source of piece i starts at src + head + 16 i: inside two consecutive aligned 16-byte pieces of src (src itself
    // is 16-byte aligned), `sh` bytes into the first
    const uint4 *sv = reinterpret_cast<const uint4 *>(src);
    const uint32_t sh = (uint32_t)(head & 15);
    const uint32_t dw = sh >> 2, phase = (sh & 3) * 8;
    uint64_t *dv = reinterpret_cast<uint64_t *>(dst + head);
    for (uint64_t i = t; i < body; i += nt) {
        const uint4 A = sv[i], B = sv[i + 1];                    // (the buffer is padded beyond len: B stays inside)
        const uint32_t w[8] = {A.x, A.y, A.z, A.w, B.x, B.y, B.z, B.w};
        uint32_t a[5];
#pragma unroll
        for (int k = 0; k < 5; k++) a[k] = dw == 0 ? w[k] : (dw == 1 ? w[k + 1] : (dw == 2 ? w[k + 2] : w[k + 3]));
        uint32_t o[4];
#pragma unroll
        for (int k = 0; k < 4; k++) o[k] = phase ? __builtin_amdgcn_alignbit(a[k + 1], a[k], phase) : a[k];
        store_through(&dv[2 * i], ((uint64_t)o[1] << 32) | o[0]);
        store_through(&dv[2 * i + 1], ((uint64_t)o[3] << 32) | o[2]);
    }
}

__global__ __launch_bounds__(64) void k_chain_done(ChainPass c, int collect) {
    __shared__ uint64_t sDone[CHAIN_MAX_RANKS];
    const int lane = threadIdx.x;
    ChainMail *box = c.mail + c.slot * CHAIN_MAX_RANKS;
    const uint64_t own = (c.params->failed ? DONE_ERROR : 0) | c.params->len;
    if (lane == 0) mail_store(&box[c.rank].done, mail_pack(c.tag, own));
    if (!collect) return;
    // the last rank with a stripe: every piece has landed when every rank before it has said so
    const bool need = lane < c.rank && ((c.active >> lane) & 1) != 0;
    bool timeout = false;
    sDone[lane] = mail_wait(&box[lane].done, need, c.tag, c.timeout_ticks, timeout);
    bool bad = __ballot(timeout) != 0;
    __syncthreads();
    if (lane == 0) {
        uint64_t total = own & (DONE_ERROR - 1);
        bad = bad || (own & DONE_ERROR);
        for (int k = 0; k < c.rank; k++) {
            bad = bad || (sDone[k] & DONE_ERROR);
            total += sDone[k] & (DONE_ERROR - 1);
        }
        if (bad) chain_fail(c);
        c.params->total = total;
        mail_store(&box[c.rank].total, mail_pack(c.tag, (bad ? DONE_ERROR : 0) | total));
    }
}

bool pass_ok(const ChainPass &c) {
    return c.mail && c.params && c.status && c.rank >= 0 && c.rank < CHAIN_MAX_RANKS && c.pred_rank < c.rank &&
           c.slot >= 0 && c.slot < CHAIN_SLOTS && (c.tag & 0x8000u) && c.tag <= 0xffffu;
}

}  // namespace

hipError_t launch_chain_place(const ChainPass &c, hipStream_t s) {
    if (!pass_ok(c)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_chain_place, dim3(1), dim3(64), 0, s, c);
    return hipGetLastError();
}

hipError_t launch_chain_offset(const ChainPass &c, size_t gather_cap, hipStream_t s) {
    if (!pass_ok(c)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_chain_offset, dim3(1), dim3(64), 0, s, c, (unsigned long long)gather_cap);
    return hipGetLastError();
}

hipError_t launch_chain_copy(const ChainPass &c, const uint8_t *scan, uint8_t *gather, hipStream_t s) {
    if (!pass_ok(c) || !scan || !gather || (reinterpret_cast<uintptr_t>(scan) & 15)) return hipErrorInvalidValue;
    // enough workgroups to keep a few hundred 16-byte stores in flight per CU; the loop covers any length
    hipLaunchKernelGGL(k_chain_copy, dim3(2048), dim3(256), 0, s, c.params, scan, gather);
    return hipGetLastError();
}

hipError_t launch_chain_done(const ChainPass &c, bool collect, hipStream_t s) {
    if (!pass_ok(c)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_chain_done, dim3(1), dim3(64), 0, s, c, collect ? 1 : 0);
    return hipGetLastError();
}

}  // namespace jpg
