// Are 2-byte stores to odd addresses of device memory performed (and in program order within a lane)?
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/unaligned_store tools/micro/unaligned_store.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

__global__ void k(uint8_t *out, int n) {
    // every lane writes n two-byte values {i, 0xEE} at consecutive byte addresses starting at an odd one: each
    // store's second byte is overwritten by the next store's first, the last byte stays 0xEE
    uint8_t *p = out + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * 64 + 1;
    for (int i = 0; i < n; i++) {
        const uint16_t v = (uint16_t)(0xEE00u | (uint32_t)(i + 1));
        *reinterpret_cast<volatile uint16_t *>(p + i) = v;
    }
}

int main() {
    const int threads = 256, blocks = 64, n = 40;
    uint8_t *d;
    if (hipMalloc(&d, (size_t)threads * blocks * 64) != hipSuccess) return 1;
    (void)hipMemset(d, 0x55, (size_t)threads * blocks * 64);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, n);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    std::vector<uint8_t> h((size_t)threads * blocks * 64);
    (void)hipMemcpy(h.data(), d, h.size(), hipMemcpyDeviceToHost);
    long bad = 0;
    for (int t = 0; t < threads * blocks; t++) {
        const uint8_t *p = &h[(size_t)t * 64];
        if (p[0] != 0x55) bad++;
        for (int i = 0; i < n; i++) bad += p[1 + i] != (uint8_t)(i + 1);
        bad += p[1 + n] != 0xEE;
        for (int i = n + 2; i < 64; i++) bad += p[i] != 0x55;
    }
    printf("unaligned 2-byte stores: %ld wrong bytes of %zu\n", bad, h.size());
    return bad != 0;
}
