// Lone-wave cost of the XXH32 accumulator chain v <- rol(v + xp, 13) * P1 in three spellings:
//   0: add, alignbit, v_mul_lo_u32               (what the compiler makes of the definition)
//   1: alignbit, v_dot2_u32_u16, v_mad_u32_u16, v_lshl_add_u32   (16-bit pieces, all full rate)
// hipcc -O3 --offload-arch=gfx950 -o xxhchain xxhchain.hip && ./xxhchain
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
constexpr uint32_t P1 = 2654435761u;

__device__ __forceinline__ uint32_t step0(uint32_t v, uint32_t xp) {
    return __builtin_rotateleft32(v + xp, 13) * P1;
}
// t = v + xp carried instead of v: t' = rol(t,13)*P1 + xp'
__device__ __forceinline__ uint32_t step1(uint32_t t, uint32_t xp_next) {
    const uint32_t r = __builtin_rotateleft32(t, 13);
    const uint32_t K = ((P1 & 0xffffu) << 16) | (P1 >> 16);  // halves [p1, p0]
    const uint32_t s = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, r), __builtin_bit_cast(u16x2, K), 0u, false);
    uint32_t a;
    asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(a) : "v"(r), "s"(P1 & 0xffffu), "v"(xp_next));
    return (s << 16) + a;
}

// t' = rol(t,13)*P1 + xp' in one v_mad_u64_u32 (only the low half is used, so the high
// half of the 64-bit addend may hold anything)
__device__ __forceinline__ uint32_t step2(uint32_t t, uint64_t xp_next) {
    const uint32_t r = __builtin_rotateleft32(t, 13);
    uint64_t m;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(m) : "v"(r), "s"(P1), "v"(xp_next) : "vcc");
    return (uint32_t)m;
}

template <int MODE> __global__ __launch_bounds__(64) void k(const uint32_t *x, uint32_t *out, int n) {
    __shared__ uint32_t y[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) y[i] = x[i];
    __syncthreads();
    if (threadIdx.x >= 4) return;
    uint32_t v = threadIdx.x * 0x1234567u + 1;
    // 64 steps per iteration from a 64-word row, rows read as 16 x uint4 like the product kernel
    if (MODE == 2) {
        const uint64_t *y2 = reinterpret_cast<const uint64_t *>(y);  // pairs: low word is the addend
        uint32_t t = v + y[0];
        for (int i = 0; i < n; i += 16) {
            const ulonglong2 *row = reinterpret_cast<const ulonglong2 *>(&y2[(i * 2) & 2047]);
            ulonglong2 q[16];
#pragma unroll
            for (int k = 0; k < 16; k++) q[k] = row[k];
            const uint64_t nx = y2[(i * 2 + 32) & 2047];
#pragma unroll
            for (int k = 0; k < 16; k++) {
                t = step2(t, q[k].y);
                t = step2(t, k < 15 ? q[k + 1].x : nx);
            }
        }
        out[threadIdx.x] = t;
        return;
    }
    if (MODE == 0) {
        for (int i = 0; i < n; i += 16) {
            const uint4 *row = reinterpret_cast<const uint4 *>(&y[(i * 4) & 4095]);
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const uint4 q = row[k];
                v = step0(v, q.x); v = step0(v, q.y); v = step0(v, q.z); v = step0(v, q.w);
            }
        }
    } else {
        uint32_t t = v + y[0];
        for (int i = 0; i < n; i += 16) {
            const uint4 *row = reinterpret_cast<const uint4 *>(&y[(i * 4) & 4095]);
            const uint32_t nx = y[(i * 4 + 64) & 4095];
            uint4 q[16];
#pragma unroll
            for (int k = 0; k < 16; k++) q[k] = row[k];
#pragma unroll
            for (int k = 0; k < 16; k++) {
                t = step1(t, q[k].y); t = step1(t, q[k].z); t = step1(t, q[k].w);
                t = step1(t, k < 15 ? q[k + 1].x : nx);
            }
        }
        v = t - y[(n * 4) & 4095];
    }
    out[threadIdx.x] = v;
}

int main() {
    uint32_t *x, *o, h[4096], r0[4], r1[4];
    for (int i = 0; i < 4096; i++) h[i] = i * 2654435761u ^ (i << 7);
    hipMalloc(&x, sizeof h); hipMalloc(&o, 16);
    hipMemcpy(x, h, sizeof h, hipMemcpyHostToDevice);
    const int n = 1 << 22;  // x4 steps
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float ms;
    for (int mode = 0; mode < 3; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(a);
            if (mode == 0) hipLaunchKernelGGL(k<0>, 1, 64, 0, 0, x, o, n);
            else if (mode == 1) hipLaunchKernelGGL(k<1>, 1, 64, 0, 0, x, o, n);
            else hipLaunchKernelGGL(k<2>, 1, 64, 0, 0, x, o, n);  // 32 steps per 16 rows: half the steps
            hipEventRecord(b); hipEventSynchronize(b);
            hipEventElapsedTime(&ms, a, b);
        }
        hipMemcpy(mode ? r1 : r0, o, 16, hipMemcpyDeviceToHost);
        const double steps = mode == 2 ? 2.0 * n : 4.0 * n;
        printf("mode %d: %.2f ns/step = %.2f GB/s of stream\n", mode, ms * 1e6 / steps, 16.0 * steps / (ms * 1e6));
    }
    printf("match: %d\n", r0[0] == r1[0] && r0[1] == r1[1] && r0[2] == r1[2] && r0[3] == r1[3]);
    return 0;
}
