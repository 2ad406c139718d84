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

// ---- content checksum: one stream, serial by construction ----
//
// Each accumulator is the chain v <- rol(v + x*P2, 13) * P1 over every 4th word of the
// stream (xxh32zero.go:66-86); rotation makes it non-linear, so the chain cannot be
// split or prefix-summed.  What can be taken off the chain is everything else, and it is
// taken off the chain's WAVE too: a lone wave issues one instruction every few cycles, so
// every load, multiply and LDS store it does itself is time the chain stands still.
// A block of four waves: waves 1..3 are producers -- each loads a 4 KiB chunk (256
// stripes) with coalesced 16-byte loads three chunks ahead, multiplies its words by P2
// and publishes them in LDS, one row per accumulator; wave 0's lanes 0..3 -- one
// accumulator each -- do nothing but read their row and run add / rotate / multiply.
// Two LDS buffers, one barrier per chunk.  tools/micro/xxhchain.hip measures the bare
// chain at 15.4 cycles per 16-byte stripe (2.5 GB/s); that is the bound of this kernel.
constexpr int kXBlock = 256;                   // threads: 1 consumer wave + 3 producer waves
constexpr int kChunkStripes = 256;             // 4 KiB per chunk
constexpr int kRow = kChunkStripes + 4;        // row stride in words: rows land in different banks
constexpr int kXBuf = 4 * kRow;                // words per buffer
constexpr int kXLds = 2 * kXBuf;               // words of LDS a caller provides

__device__ __forceinline__ uint4 ld128(const uint8_t *p) {
    uint4 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}

// v: accumulator `lane` in lanes 0..3 of wave 0 (threads 0..3); the result likewise.
// Every thread of the block must call it (barriers inside).  y: kXLds words of LDS.
__device__ __forceinline__ uint32_t block_accumulate(const uint8_t *__restrict__ p, size_t stripes,
                                                     uint32_t v, uint32_t *y) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t chunks = stripes / kChunkStripes;
    uint4 r[4];
    auto load = [&](size_t c) {
        const uint8_t *q = p + c * (size_t)(kChunkStripes * 16) + lane * 16;
#pragma unroll
        for (int j = 0; j < 4; j++) r[j] = ld128(q + j * 1024);
    };
    if (wave && (size_t)(wave - 1) < chunks) load(wave - 1);
    // step s: the producer of chunk s publishes it while the consumer runs chunk s-1
    for (size_t s = 0; s <= chunks; s++) {
        if (wave) {
            if (s < chunks && s % 3 == (size_t)(wave - 1)) {
                uint32_t *b = y + (s & 1) * kXBuf;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    b[0 * kRow + j * 64 + lane] = r[j].x * P2;
                    b[1 * kRow + j * 64 + lane] = r[j].y * P2;
                    b[2 * kRow + j * 64 + lane] = r[j].z * P2;
                    b[3 * kRow + j * 64 + lane] = r[j].w * P2;
                }
                if (s + 3 < chunks) load(s + 3);
            }
        } else if (s > 0 && lane < 4) {
            const uint4 *row = reinterpret_cast<const uint4 *>(y + ((s - 1) & 1) * kXBuf + lane * kRow);
            for (int g = 0; g < kChunkStripes / 64; g++) {
                uint4 t[16];
#pragma unroll
                for (int k = 0; k < 16; k++) t[k] = row[g * 16 + k];
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    v = rol(v + t[k].x, 13) * P1;
                    v = rol(v + t[k].y, 13) * P1;
                    v = rol(v + t[k].z, 13) * P1;
                    v = rol(v + t[k].w, 13) * P1;
                }
            }
        }
        __syncthreads();
    }
    if (wave == 0 && lane < 4)
        for (size_t k = chunks * kChunkStripes; k < stripes; k++) v = xround(v, ld32(p + k * 16 + lane * 4));
    return v;
}

__device__ __forceinline__ uint32_t merge_accumulators(uint32_t v) {
    return rol(__shfl(v, 0), 1) + rol(__shfl(v, 1), 7) + rol(__shfl(v, 2), 12) + rol(__shfl(v, 3), 18);
}

// Content checksum: XXHZero over the whole stream.  Sum32 tests uint32(totalLen)
// >= 16 (xxh32zero.go:113-118), so a stream of 2^32*k + r bytes with r < 16
// takes the short branch and drops the accumulators; reproduced here.
__global__ __launch_bounds__(kXBlock) void k_xxh32_stream(const uint8_t *__restrict__ src, size_t n,
                                                          uint32_t *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint32_t y[kXLds];
    if (blockIdx.x != 0) return;
    const int lane = threadIdx.x;
    const uint32_t len_word = (uint32_t)n;
    const size_t stripes = n / 16;
    uint32_t h = len_word;
    if (len_word >= 16) {
        const uint32_t init = lane == 0 ? P1 + P2 : lane == 1 ? P2 : lane == 2 ? 0u : 0u - P1;
        h += merge_accumulators(block_accumulate(src, stripes, init, y));
    } else {
        h += P5;  // n < 16, or the truncation quirk: accumulators ignored
    }
    // the tail is the last n % 16 buffered bytes either way
    if (lane == 0) out[0] = xtail(h, src + stripes * 16, n - stripes * 16);
}

// One BLOCK per span (few long spans: a frame of a handful of 4 MiB blocks, the
// single-block ChecksumZero entry point): the same cooperative accumulate.
__global__ __launch_bounds__(kXBlock) void k_xxh32_spans_wave(const uint8_t *__restrict__ base, size_t n,
                                                         size_t stride,
                                                         const uint64_t *__restrict__ off,
                                                         const int32_t *__restrict__ len,
                                                         size_t n_spans,
                                                         uint32_t *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint32_t y[kXLds];
    const size_t i = blockIdx.x;
    if (i >= n_spans) return;
    const int lane = threadIdx.x;
    const size_t o = off ? (size_t)off[i] : i * stride;
    size_t l;
    if (len) {
        l = len[i] < 0 ? 0 : (size_t)len[i];
    } else {
        const size_t left = n > o ? n - o : 0;
        l = left < stride ? left : stride;
    }
    const uint8_t *p = base + o;
    const size_t stripes = l / 16;
    uint32_t h = (uint32_t)l;
    if (l >= 16) {
        const uint32_t init = lane == 0 ? P1 + P2 : lane == 1 ? P2 : lane == 2 ? 0u : 0u - P1;
        h += merge_accumulators(block_accumulate(p, stripes, init, y));
    } else {
        h += P5;
    }
    if (lane == 0) out[i] = xtail(h, p + stripes * 16, l - stripes * 16);
}

// Chained form of the content checksum for a stream cut across GPUs: state =
// {v1, v2, v3, v4, totalLen lo, totalLen hi}.  Non-final pieces must be multiples
// of 16 bytes (block sizes are), so no carry buffer travels (xxh32zero.go:58-86).
__global__ __launch_bounds__(kXBlock) void k_xxh32_update(const uint8_t *__restrict__ src, size_t n,
                                                          uint32_t *__restrict__ state, int first,
                                                          int final_piece,
                                                          uint32_t *__restrict__ digest) {
    __shared__ __attribute__((aligned(16))) uint32_t y[kXLds];
    if (blockIdx.x != 0) return;
    const int lane = threadIdx.x;
    uint32_t v;
    uint64_t total;
    if (first) {  // Reset(): xxh32zero.go:37-44
        v = lane == 0 ? P1 + P2 : lane == 1 ? P2 : lane == 2 ? 0u : 0u - P1;
        total = 0;
    } else {
        v = state[lane & 3];
        total = (uint64_t)state[4] | ((uint64_t)state[5] << 32);
    }
    const size_t stripes = n / 16;
    v = block_accumulate(src, stripes, v, y);
    total += n;
    __syncthreads();  // every thread has read the incoming state
    if (lane < 4) state[lane] = v;
    if (lane == 0) {
        state[4] = (uint32_t)total;
        state[5] = (uint32_t)(total >> 32);
    }
    if (final_piece) {  // Sum32(): xxh32zero.go:112-139 incl. the uint32 truncation
        uint32_t h = (uint32_t)total;
        const uint32_t m = merge_accumulators(v);
        if (h >= 16) h += m;
        else h += P5;
        if (lane == 0) digest[0] = xtail(h, src + stripes * 16, n - stripes * 16);
    }
}

hipError_t launch_xxh32_update(const uint8_t *src, size_t n, uint32_t *state, int first,
                               int final_piece, uint32_t *digest, hipStream_t st) {
    hipLaunchKernelGGL(k_xxh32_update, dim3(1), dim3(kXBlock), 0, st, src, n, state, first,
                       final_piece, digest);
    return hipGetLastError();
}

hipError_t launch_xxh32_blocks(const uint8_t *base, size_t n, size_t stride, const uint64_t *off,
                               const int32_t *len, size_t n_spans, uint32_t *out,
                               hipStream_t st) {
    if (n_spans == 0) return hipSuccess;
    // Few spans: a lane per span leaves the chip empty and each lane latency-bound;
    // a wave per span streams its span at the accumulator chain's speed instead.
    if (n_spans < 16384) {
        hipLaunchKernelGGL(k_xxh32_spans_wave, dim3((unsigned)n_spans), dim3(kXBlock), 0, st, base, n,
                           stride, off, len, n_spans, out);
        return hipGetLastError();
    }
    const unsigned grid = (unsigned)((n_spans + kWave - 1) / kWave);
    hipLaunchKernelGGL(k_xxh32_blocks, dim3(grid), dim3(kWave), 0, st, base, n, stride, off, len,
                       n_spans, out);
    return hipGetLastError();
}

hipError_t launch_xxh32_stream(const uint8_t *src, size_t n, uint32_t *out, hipStream_t st) {
    hipLaunchKernelGGL(k_xxh32_stream, dim3(1), dim3(kXBlock), 0, st, src, n, out);
    return hipGetLastError();
}

}  // namespace lz4hip
