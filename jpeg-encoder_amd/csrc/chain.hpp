// Device-side helpers of the stripe chain (chain.hip): the mailbox words and the bounded wait on them.
#pragma once
#include "../../include/jpeg_hip.h"

#include "internal.hpp"

namespace jpg {

constexpr uint64_t VALUE_MASK = (1ull << 48) - 1;
constexpr uint64_t DONE_ERROR = 1ull << 47;

__device__ __forceinline__ uint64_t mail_pack(uint32_t tag, uint64_t value) { return ((uint64_t)tag << 48) | (value & VALUE_MASK); }

// the mailbox lies in host memory shared by several GPUs (and processes): system scope
__device__ __forceinline__ void mail_store(uint64_t *p, uint64_t w) {
    __hip_atomic_store(p, w, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ uint64_t mail_load(const uint64_t *p) {
    return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Poll *p until it carries `tag` (lanes with need == false do not poll); all lanes of the wave leave together.
// Returns the value, or 0 with `timeout` set when the clock ran out.
__device__ __forceinline__ uint64_t mail_wait(const uint64_t *p, bool need, uint32_t tag, uint64_t timeout_ticks, bool &timeout) {
    uint64_t w = 0;
    bool ok = !need;
    const uint64_t t0 = wall_clock64();
    for (uint32_t spin = 0;; spin++) {
        if (!ok) {
            w = mail_load(p);
            ok = (w >> 48) == tag;
        }
        if (__ballot(!ok) == 0) break;
        if ((spin & 15) == 15 && wall_clock64() - t0 > timeout_ticks) { timeout = true; break; }
        __builtin_amdgcn_s_sleep(8);
    }
    return need && ok ? (w & VALUE_MASK) : 0;
}

__device__ __forceinline__ void chain_fail(const ChainPass &c) {
    c.params->failed = 1;
    atomicMin(&c.status->error, -(int)((c.err_epoch << 8) | (uint32_t)(-JPG_ERR_CHAIN)));
}

// First step of a pass, called by one wave: clears params->failed, publishes this stripe's last DC coefficients
// (dc3, LDS or registers of the caller) and waits for the predecessor's -> params->pred.
__device__ __forceinline__ void chain_exchange_dc(const ChainPass &c, const int16_t *dc3, int lane) {
    ChainMail *box = c.mail + c.slot * CHAIN_MAX_RANKS;
    if (lane == 0) {
        c.params->failed = 0;
        const uint64_t v = ((uint64_t)(uint16_t)dc3[0] << 32) | ((uint64_t)(uint16_t)dc3[1] << 16) | (uint16_t)dc3[2];
        mail_store(&box[c.rank].dc, mail_pack(c.tag, v));
    }
    bool timeout = false;
    const bool need = lane == 0 && c.pred_rank >= 0;
    const uint64_t v = mail_wait(&box[need ? c.pred_rank : 0].dc, need, c.tag, c.timeout_ticks, timeout);
    if (lane == 0) {
        c.params->pred[0] = (int16_t)(v >> 32);       // src/jpeg/entropy_coding.rs:38-40: zero without a predecessor
        c.params->pred[1] = (int16_t)(v >> 16);
        c.params->pred[2] = (int16_t)v;
        c.params->pred[3] = 0;
    }
    if (__ballot(timeout) && lane == 0) chain_fail(c);
}

}  // namespace jpg
