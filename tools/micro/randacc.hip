// Random-access microbenchmark: how many independent 2-byte table probes per second
// does the memory system sustain, by footprint (L2 / Infinity Cache / HBM) and kind
// (read, write, read+write of the same slot)?   hipcc --offload-arch=gfx950 -O3 randacc.hip -o randacc
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

template <int MODE>  // 0 read, 1 write, 2 read then write
__global__ void k(uint16_t *t, uint64_t mask, int iters, uint32_t *sink) {
    uint32_t x = mix(blockIdx.x * blockDim.x + threadIdx.x + 1);
    uint32_t acc = 0;
    for (int i = 0; i < iters; i += 4) {
        uint64_t a[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            x = mix(x + 0x9e3779b9U);
            a[j] = (((uint64_t)x << 7) ^ mix(x)) & mask;
        }
        uint32_t v[4] = {0, 0, 0, 0};
        if (MODE != 1) {
#pragma unroll
            for (int j = 0; j < 4; j++) v[j] = t[a[j]];
        }
        if (MODE != 0) {
#pragma unroll
            for (int j = 0; j < 4; j++) t[a[j]] = (uint16_t)(i + v[j]);
        }
        acc += v[0] + v[1] + v[2] + v[3];
    }
    if (acc == 0x12345678) sink[0] = acc;
}

int main(int argc, char **argv) {
    const size_t cap = 8ull << 30;
    uint16_t *t; uint32_t *sink;
    hipMalloc(&t, cap); hipMalloc(&sink, 4);
    hipMemset(t, 0, cap);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int waves_per_cu[] = {2, 8, 16};
    for (size_t fp : {16ull << 20, 128ull << 20, 1ull << 30, 8ull << 30})
        for (int w : waves_per_cu)
            for (int mode = 0; mode < 3; mode++) {
                const uint64_t mask = fp / 2 - 1;
                const int grid = 256 * w, iters = 2048;
                for (int rep = 0; rep < 2; rep++) {
                    hipEventRecord(e0);
                    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(64), 0, 0, t, mask, iters, sink);
                    if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(64), 0, 0, t, mask, iters, sink);
                    if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(64), 0, 0, t, mask, iters, sink);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                }
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double acc = (double)grid * 64 * iters * (mode == 2 ? 2 : 1);
                printf("footprint %5zu MiB  waves/CU %2d  %s  %7.2f G accesses/s\n", fp >> 20, w,
                       mode == 0 ? "read " : mode == 1 ? "write" : "rd+wr", acc / ms / 1e6);
                fflush(stdout);
            }
    return 0;
}
