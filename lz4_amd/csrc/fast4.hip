// fast4.hip -- Fast compressor, generation 4: one LANE per block.
//
// Reference: internal/lz4block/block.go:98-295 ((*Compressor).CompressBlock).
//
// The window-parallel kernels (fast2/fast3) spend one 64-lane wave on one block and
// are bound by the number of 64-byte memory requests they issue: every position of
// every window is gathered from the hash table and from its candidate, although
// the greedy parse only ever probes the positions it steps on.  When a launch has
// many blocks the cheaper way to fill the machine is the other axis: each lane
// parses its own block with its own u16 table (128 KiB, in global memory,
// pre-zeroed: an all-zero table IS the reference's reset table, trap T1), and does
// exactly the probes the reference does.  64 lanes issue 64 independent requests
// per instruction, so the memory system sees the same parallelism with about half
// the requests per input byte.
//
// To keep lanes from waiting for each other, the parse is a per-lane state machine
// stepped by a wave-uniform loop: in each tick a lane does one bounded piece of
// work (one probe step; a few 8-byte match-extension strides; a few 8-byte literal
// copies) and a long match or a long literal run only occupies its own lane.
#include "common.h"

namespace lz4hip {

namespace {

__device__ __forceinline__ void st16(uint8_t *p, uint32_t v) {
    p[0] = (uint8_t)v;
    p[1] = (uint8_t)(v >> 8);
}

// get (block.go:74-86) on a table whose unset slots read 0.
__device__ __forceinline__ int tget(const uint16_t *t, uint32_t h, int si) {
    int i = (int)t[h] + (si & ~kWinMask);
    if (i >= si) i -= kWinSize;
    return i;
}

// Stores the low r (1..7) bytes of v at q.
__device__ __forceinline__ void st_tail(uint8_t *q, uint64_t v, int r) {
    if (r & 4) {
        const uint32_t x = (uint32_t)v;
        __builtin_memcpy(q, &x, 4);
        q += 4;
        v >>= 32;
    }
    if (r & 2) {
        st16(q, (uint32_t)v);
        q += 2;
        v >>= 16;
    }
    if (r & 1) *q = (uint8_t)v;
}
__device__ __forceinline__ void st64(uint8_t *q, uint64_t v) { __builtin_memcpy(q, &v, 8); }

// Whole parse of one block by one lane.  Same probes, in the same order, as
// block.go:124-256; the loads of a step are grouped so that a step costs few
// dependent memory round trips: {3 table reads} -> {3 candidates} ->
// {extension + literal bytes} -> stores.
__device__ int fast4_block(const uint8_t *__restrict__ s, int n, uint8_t *__restrict__ d, int dlen,
                           uint16_t *__restrict__ t) {
    const bool not_compressible = dlen < (int)compress_bound32((uint32_t)n);
    int si = 0, di = 0, anchor = 0;
    const int sn = n - kMfLimit;
    if (sn > 0) {
        uint64_t w = ld64(s);
        while (si < sn) {
            const uint32_t h1 = hash6(w), h2 = hash6(w >> 8), h3 = hash6(w >> 16);
            uint32_t v1 = t[h1], v2 = t[h2], v3 = t[h3];
            // T3: the third get sees the two puts of this step (block.go:134-145).
            if (h3 == h2)
                v3 = (uint16_t)(si + 1);
            else if (h3 == h1)
                v3 = (uint16_t)si;
            t[h1] = (uint16_t)si;
            asm volatile("" ::: "memory");
            t[h2] = (uint16_t)(si + 1);
            int r1 = (int)v1 + (si & ~kWinMask);
            if (r1 >= si) r1 -= kWinSize;
            int r2 = (int)v2 + ((si + 1) & ~kWinMask);
            if (r2 >= si + 1) r2 -= kWinSize;
            int r3 = (int)v3 + ((si + 2) & ~kWinMask);
            if (r3 >= si + 2) r3 -= kWinSize;
            // offsets are in [1, 65536] by construction; 65536 is the invalid one
            const bool o1 = si - r1 < kWinSize, o2 = si + 1 - r2 < kWinSize, o3 = si + 2 - r3 < kWinSize;
            uint32_t c1 = 0, c2 = 0, c3 = 0;
            if (o1) c1 = ld32(s + r1);
            if (o2) c2 = ld32(s + r2);
            if (o3) c3 = ld32(s + r3);
            const bool m1 = o1 && c1 == (uint32_t)w;
            const bool m2 = o2 && c2 == (uint32_t)(w >> 8);
            const bool m3 = o3 && c3 == (uint32_t)(w >> 16);
            if (!m1 && !m2) t[h3] = (uint16_t)(si + 2);  // block.go:155
            if (!(m1 || m2 || m3)) {
                si += 4 + ((si + 2 - anchor) >> kAdaptSkipLog);  // block.go:159
                if (si < sn) w = ld64(s + si);
                continue;
            }
            const int p = m1 ? si : m2 ? si + 1 : si + 2;  // found position
            const int r = m1 ? r1 : m2 ? r2 : r3;
            const int offset = p - r;

            // one round trip: backward bytes, first forward stride, first literal bytes
            const int l0 = p - anchor;
            const int blim = l0 < r ? l0 : r;  // block.go:172: lLen > 0 && tOff >= 0
            const bool bfast = blim > 0 && r >= 8;  // then p >= 8 too
            uint64_t bs_ = 0, br_ = 0, fs_ = 0, fr_ = 0, L0 = 0, L1 = 0, L2 = 0, L3 = 0;
            if (bfast) {
                bs_ = ld64(s + p - 8);
                br_ = ld64(s + r - 8);
            }
            const bool ffast = p + 4 + 8 <= sn;
            if (ffast) {
                fs_ = ld64(s + p + 4);
                fr_ = ld64(s + r + 4);
            }
            if (l0 > 0) L0 = ld64(s + anchor);
            if (l0 > 8) L1 = ld64(s + anchor + 8);
            if (l0 > 16) L2 = ld64(s + anchor + 16);
            if (l0 > 24) L3 = ld64(s + anchor + 24);

            int back = 0;
            if (blim > 0) {
                if (bfast) {
                    const uint64_t xb = bs_ ^ br_;
                    back = xb ? __builtin_clzll(xb) >> 3 : 8;
                    if (back > blim) back = blim;
                }
                if (!bfast || (back == 8 && blim > 8))
                    while (back < blim && s[p - 1 - back] == s[r - 1 - back]) back++;
            }
            si = p + 4;
            if (ffast) {
                uint64_t x = fs_ ^ fr_;
                for (;;) {
                    if (x) {
                        si += __builtin_ctzll(x) >> 3;
                        break;
                    }
                    si += 8;
                    // long matches: four strides per round trip while they fit
                    while (si + 32 <= sn) {
                        const uint64_t x0 = ld64(s + si) ^ ld64(s + si - offset);
                        const uint64_t x1 = ld64(s + si + 8) ^ ld64(s + si + 8 - offset);
                        const uint64_t x2 = ld64(s + si + 16) ^ ld64(s + si + 16 - offset);
                        const uint64_t x3 = ld64(s + si + 24) ^ ld64(s + si + 24 - offset);
                        if (x0 | x1 | x2 | x3) break;
                        si += 32;
                    }
                    if (si + 8 > sn) break;
                    x = ld64(s + si) ^ ld64(s + si - offset);
                }
            }
            const int l_len = l0 - back;
            int m_len = si - (p - back) - kMinMatch;

            // emit: block.go:195-248
            if (di >= dlen) return -1;
            const uint32_t tok = m_len < 15 ? (uint32_t)m_len : 15u;
            if (l_len < 15) {
                d[di] = (uint8_t)(tok | ((uint32_t)l_len << 4));
            } else {
                d[di] = (uint8_t)(tok | 0xF0);
                di++;
                int l = l_len - 15;
                for (; l >= 255 && di < dlen; l -= 255) d[di++] = 0xFF;
                if (di >= dlen) return -1;
                d[di] = (uint8_t)l;
            }
            di++;
            if (di + l_len > dlen) return -1;
            {
                uint8_t *q = d + di;
                if (l_len >= 8) st64(q, L0); else if (l_len > 0) st_tail(q, L0, l_len);
                if (l_len >= 16) st64(q + 8, L1); else if (l_len > 8) st_tail(q + 8, L1, l_len - 8);
                if (l_len >= 24) st64(q + 16, L2); else if (l_len > 16) st_tail(q + 16, L2, l_len - 16);
                if (l_len >= 32) st64(q + 24, L3); else if (l_len > 24) st_tail(q + 24, L3, l_len - 24);
                if (l_len > 32) {
                    const uint8_t *ps = s + anchor;
                    int k = 32;
                    for (; k + 8 <= l_len; k += 8) st64(q + k, ld64(ps + k));
                    if (k < l_len) st_tail(q + k, ld64(ps + k), l_len - k);  // ps+k+8 <= p+8 < n
                }
            }
            di += l_len + 2;
            anchor = si;
            if (di > dlen) return -1;
            st16(d + di - 2, (uint32_t)offset);
            if (m_len >= 15) {
                for (m_len -= 15; m_len >= 255 && di < dlen; m_len -= 255) d[di++] = 0xFF;
                if (di >= dlen) return -1;
                d[di++] = (uint8_t)m_len;
            }
            if (si >= sn) break;
            const uint64_t u = ld64(s + si - 2);
            w = ld64(s + si);
            t[hash6(u)] = (uint16_t)(si - 2);  // block.go:254-255
        }
    }
    if (not_compressible && anchor == 0) return 0;
    if (di >= dlen) return -1;
    {
        int l_len = n - anchor;
        if (l_len < 15) {
            d[di] = (uint8_t)(l_len << 4);
        } else {
            d[di] = 0xF0;
            di++;
            for (l_len -= 15; l_len >= 255 && di < dlen; l_len -= 255) d[di++] = 0xFF;
            if (di >= dlen) return -1;
            d[di] = (uint8_t)l_len;
        }
        di++;
    }
    if (not_compressible && di >= anchor) return 0;
    if (di + n - anchor > dlen) return -1;
    for (int k = 0; k < n - anchor; k++) d[di + k] = s[anchor + k];
    return di + n - anchor;
}

__device__ __forceinline__ int dlen_of(int dst_mode, int n, uint32_t dst_len_x) {
    return dst_mode == LZ4HIP_DST_FRAME  ? n
           : dst_mode == LZ4HIP_DST_BOUND ? (int)compress_bound32((uint32_t)n)
                                          : (int)dst_len_x;
}

}  // namespace

__global__ __launch_bounds__(64) void k_fast4_lanes(const uint8_t *__restrict__ src, size_t n_total,
                                                    uint32_t block_size, size_t nb, int dst_mode,
                                                    uint32_t dst_len_x, uint8_t *__restrict__ dst,
                                                    size_t dst_stride, int32_t *__restrict__ result,
                                                    uint16_t *__restrict__ ws) {
    const size_t b = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (b >= nb) return;
    const size_t start = b * (size_t)block_size;
    const size_t left = n_total > start ? n_total - start : 0;
    const int n = (int)(left < block_size ? left : block_size);
    result[b] = fast4_block(src + start, n, dst + b * dst_stride, dlen_of(dst_mode, n, dst_len_x),
                            ws + b * (size_t)(1 << 16));
}

// ws: nb tables of 65536 u16, zeroed here.
hipError_t launch_fast4(const uint8_t *src, size_t n, uint32_t bs, size_t nb, int dst_mode,
                        uint32_t dst_len_x, uint8_t *dst, size_t dst_stride, int32_t *result,
                        uint16_t *ws, hipStream_t st) {
    if (nb == 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(ws, 0, nb * (size_t)(1 << 17), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_fast4_lanes, dim3((unsigned)((nb + 63) / 64)), dim3(64), 0, st, src, n, bs,
                       nb, dst_mode, dst_len_x, dst, dst_stride, result, ws);
    return hipGetLastError();
}

}  // namespace lz4hip
