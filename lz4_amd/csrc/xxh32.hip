// xxh32.hip -- XXH32 seed 0 on gfx950.
//
// Restates xxh32.ChecksumZero (internal/xxh32/xxh32zero.go:142-184) for block
// checksums and XXHZero.Write/Sum32 (:58-139) for the content checksum.
// XXH32's four accumulators are each a serial chain over 16-byte stripes, so a
// span cannot be split across lanes.  Block checksums therefore run one span
// per lane (many spans in flight); the content checksum is one serial walk.
#include "common.h"

namespace lz4hip {

constexpr uint32_t P1 = 2654435761u, P2 = 2246822519u, P3 = 3266489917u, P4 = 668265263u,
                   P5 = 374761393u;

__device__ __forceinline__ uint32_t rol(uint32_t u, int k) {
    return __builtin_rotateleft32(u, k);
}
__device__ __forceinline__ uint32_t xround(uint32_t v, uint32_t x) {
    return rol(v + x * P2, 13) * P1;
}

// xxh32zero.go:166-181
__device__ __forceinline__ uint32_t xtail(uint32_t h32, const uint8_t *p, size_t n) {
    size_t i = 0;
    for (; i + 4 <= n; i += 4) {
        h32 += ld32(p + i) * P3;
        h32 = rol(h32, 17) * P4;
    }
    for (; i < n; i++) {
        h32 += (uint32_t)p[i] * P5;
        h32 = rol(h32, 11) * P1;
    }
    h32 ^= h32 >> 15;
    h32 *= P2;
    h32 ^= h32 >> 13;
    h32 *= P3;
    h32 ^= h32 >> 16;
    return h32;
}

// One span per thread.  Spans start 16-byte aligned whenever they are block cuts
// of an aligned buffer; unaligned spans (arbitrary d_off) take dword loads.
__device__ uint32_t xxh32_span(const uint8_t *p, size_t n, uint32_t len_word) {
    uint32_t h32 = len_word;
    size_t i = 0;
    if (n >= 16) {
        uint32_t v1 = P1 + P2, v2 = P2, v3 = 0, v4 = 0u - P1;
        if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
            const uint4 *q = reinterpret_cast<const uint4 *>(p);
            const size_t stripes = n / 16;
            size_t k = 0;
            for (; k + 4 <= stripes; k += 4) {  // 4 loads in flight per lane
                const uint4 a = q[k], b = q[k + 1], c = q[k + 2], e = q[k + 3];
                v1 = xround(v1, a.x); v2 = xround(v2, a.y); v3 = xround(v3, a.z); v4 = xround(v4, a.w);
                v1 = xround(v1, b.x); v2 = xround(v2, b.y); v3 = xround(v3, b.z); v4 = xround(v4, b.w);
                v1 = xround(v1, c.x); v2 = xround(v2, c.y); v3 = xround(v3, c.z); v4 = xround(v4, c.w);
                v1 = xround(v1, e.x); v2 = xround(v2, e.y); v3 = xround(v3, e.z); v4 = xround(v4, e.w);
            }
            for (; k < stripes; k++) {
                const uint4 a = q[k];
                v1 = xround(v1, a.x); v2 = xround(v2, a.y); v3 = xround(v3, a.z); v4 = xround(v4, a.w);
            }
            i = stripes * 16;
        } else {
            for (; i + 16 <= n; i += 16) {
                v1 = xround(v1, ld32(p + i));
                v2 = xround(v2, ld32(p + i + 4));
                v3 = xround(v3, ld32(p + i + 8));
                v4 = xround(v4, ld32(p + i + 12));
            }
        }
        h32 += rol(v1, 1) + rol(v2, 7) + rol(v3, 12) + rol(v4, 18);
    } else {
        h32 += P5;
    }
    return xtail(h32, p + i, n - i);
}

__global__ __launch_bounds__(64) void k_xxh32_blocks(const uint8_t *__restrict__ base, size_t n,
                                                     size_t stride,
                                                     const uint64_t *__restrict__ off,
                                                     const int32_t *__restrict__ len,
                                                     size_t n_spans, uint32_t *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_spans) return;
    const size_t o = off ? (size_t)off[i] : i * stride;
    size_t l;
    if (len) {
        l = len[i] < 0 ? 0 : (size_t)len[i];
    } else {
        const size_t left = n > o ? n - o : 0;
        l = left < stride ? left : stride;
    }
    out[i] = xxh32_span(base + o, l, (uint32_t)l);
}

// Content checksum: XXHZero over the whole stream.  Sum32 tests uint32(totalLen)
// >= 16 (xxh32zero.go:113-118), so a stream of 2^32*k + r bytes with r < 16
// takes the short branch and drops the accumulators; reproduced here.
__global__ __launch_bounds__(64) void k_xxh32_stream(const uint8_t *__restrict__ src, size_t n,
                                                     uint32_t *__restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const uint32_t len_word = (uint32_t)n;
    uint32_t h;
    if (n >= 16 && len_word < 16) {
        // accumulators ignored; the tail is still the last n % 16 buffered bytes
        const size_t i = n - (n & 15);
        h = xtail(len_word + P5, src + i, n - i);
    } else {
        h = xxh32_span(src, n, len_word);
    }
    out[0] = h;
}

// Chained form of the content checksum for a stream cut across GPUs: state =
// {v1, v2, v3, v4, totalLen lo, totalLen hi}.  Non-final pieces must be multiples
// of 16 bytes (block sizes are), so no carry buffer travels (xxh32zero.go:58-86).
__global__ __launch_bounds__(64) void k_xxh32_update(const uint8_t *__restrict__ src, size_t n,
                                                     uint32_t *__restrict__ state, int first,
                                                     int final_piece,
                                                     uint32_t *__restrict__ digest) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint32_t v1, v2, v3, v4;
    uint64_t total;
    if (first) {  // Reset(): xxh32zero.go:37-44
        v1 = P1 + P2; v2 = P2; v3 = 0; v4 = 0u - P1; total = 0;
    } else {
        v1 = state[0]; v2 = state[1]; v3 = state[2]; v4 = state[3];
        total = (uint64_t)state[4] | ((uint64_t)state[5] << 32);
    }
    const size_t stripes = n / 16;
    if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        const uint4 *q = reinterpret_cast<const uint4 *>(src);
        for (size_t k = 0; k < stripes; k++) {
            const uint4 a = q[k];
            v1 = xround(v1, a.x); v2 = xround(v2, a.y); v3 = xround(v3, a.z); v4 = xround(v4, a.w);
        }
    } else {
        for (size_t k = 0; k < stripes; k++) {
            const uint8_t *p = src + k * 16;
            v1 = xround(v1, ld32(p)); v2 = xround(v2, ld32(p + 4));
            v3 = xround(v3, ld32(p + 8)); v4 = xround(v4, ld32(p + 12));
        }
    }
    total += n;
    state[0] = v1; state[1] = v2; state[2] = v3; state[3] = v4;
    state[4] = (uint32_t)total; state[5] = (uint32_t)(total >> 32);
    if (final_piece) {  // Sum32(): xxh32zero.go:112-139 incl. the uint32 truncation
        uint32_t h = (uint32_t)total;
        if (h >= 16) h += rol(v1, 1) + rol(v2, 7) + rol(v3, 12) + rol(v4, 18);
        else h += P5;
        digest[0] = xtail(h, src + stripes * 16, n - stripes * 16);
    }
}

hipError_t launch_xxh32_update(const uint8_t *src, size_t n, uint32_t *state, int first,
                               int final_piece, uint32_t *digest, hipStream_t st) {
    hipLaunchKernelGGL(k_xxh32_update, dim3(1), dim3(kWave), 0, st, src, n, state, first,
                       final_piece, digest);
    return hipGetLastError();
}

hipError_t launch_xxh32_blocks(const uint8_t *base, size_t n, size_t stride, const uint64_t *off,
                               const int32_t *len, size_t n_spans, uint32_t *out,
                               hipStream_t st) {
    if (n_spans == 0) return hipSuccess;
    const unsigned grid = (unsigned)((n_spans + kWave - 1) / kWave);
    hipLaunchKernelGGL(k_xxh32_blocks, dim3(grid), dim3(kWave), 0, st, base, n, stride, off, len,
                       n_spans, out);
    return hipGetLastError();
}

hipError_t launch_xxh32_stream(const uint8_t *src, size_t n, uint32_t *out, hipStream_t st) {
    hipLaunchKernelGGL(k_xxh32_stream, dim3(1), dim3(kWave), 0, st, src, n, out);
    return hipGetLastError();
}

}  // namespace lz4hip
