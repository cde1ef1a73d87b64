// How fast can the workgroups of a kernel draw numbers from one counter?  Every workgroup's last thread draws K
// tickets one after the other (each waits for the one before); variants: the whole device on one counter, one
// counter per group of 8 workgroups, tickets drawn two or four at a time.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/ticket_rate tools/micro/ticket_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void draw(unsigned *counters, int ncounters, int k, int stride_words, unsigned *sink, int work) {
    __shared__ unsigned s;
    unsigned *c = counters + (size_t)(blockIdx.x % ncounters) * stride_words;
    unsigned acc = 0;
    for (int i = 0; i < k; i++) {
        if (threadIdx.x == blockDim.x - 1) s = atomicAdd(c, 1u);
        __syncthreads();
        acc += s;
        // some work between two draws (a dependent chain of LDS-free ALU ops)
        for (int w = 0; w < work; w++) acc = acc * 1664525u + 1013904223u;
        __syncthreads();
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char **argv) {
    unsigned *counters, *sink;
    const char *kind = argc > 1 ? argv[1] : "plain";
    hipError_t e = hipSuccess;
    if (kind[0] == 'f') e = hipExtMallocWithFlags((void **)&counters, 1 << 20, hipDeviceMallocFinegrained);
    else if (kind[0] == 'u') e = hipExtMallocWithFlags((void **)&counters, 1 << 20, hipDeviceMallocUncached);
    else if (kind[0] == 'h') e = hipHostMalloc((void **)&counters, 1 << 20, hipHostMallocMapped);
    else e = hipMalloc(&counters, 1 << 20);
    if (e != hipSuccess) { printf("allocation (%s) failed: %d\n", kind, (int)e); return 1; }
    printf("counters in %s memory\n", kind);
    hipMalloc(&sink, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int blocks = 1024, threads = 512;
    for (int work : {0}) {
        for (int nc : {1, 8, 64, 1024}) {
            for (int k : {1, 16}) {
                float best = 1e9f;
                for (int rep = 0; rep < 5; rep++) {
                    hipMemset(counters, 0, 1 << 20);
                    hipDeviceSynchronize();
                    hipEventRecord(e0);
                    hipLaunchKernelGGL(draw, dim3(blocks), dim3(threads), 0, 0, counters, nc, k, 64, sink, work);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    best = ms < best ? ms : best;
                }
                printf("work %5d  counters %5d  draws per workgroup %3d: %8.2f us, %7.2f ns per draw over the device\n", work, nc, k,
                       best * 1e3, best * 1e6 / (blocks * k));
            }
        }
    }
    return 0;
}
