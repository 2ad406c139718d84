// fast.hip -- Fast (greedy) LZ4 block compressor for gfx950.
//
// Restates (*Compressor).CompressBlock, internal/lz4block/block.go:105-295, with
// the same hash (block.go:35-38), the same 65536 x u16 table and the same
// get/put order, so the parse -- and therefore every output byte -- is the
// reference's.  One wavefront per independent block:
//   * the table lives in LDS (128 KiB; an all-zero table is the reference's
//     "inUse bitmap cleared" state, block.go:74-94);
//   * the greedy walk is wave-uniform; match extension (backward and forward),
//     literal copies and length runs are done 64 lanes wide.
#include "common.h"

namespace lz4hip {

// (*Compressor).get, block.go:74-86, on a zero-initialised table.
__device__ __forceinline__ int fast_get(const uint16_t *table, uint32_t h, int si) {
    int i = (int)table[h] + (si & ~kWinMask);
    if (i >= si) i -= kWinSize;
    return i;
}

// Candidate test of block.go:141,151,157.
__device__ __forceinline__ bool fast_ok(const uint8_t *s, int pos, int ref, uint32_t word) {
    const int offset = pos - ref;
    return offset > 0 && offset < kWinSize && word == ld32(s + ref);
}

// One block, one wavefront.  `table` is 65536 x u16, zeroed by the caller's wave here;
// it may live in LDS or in global memory (same-wave accesses are ordered either way).
template <typename TablePtr>
__device__ __forceinline__ int fast_block(const uint8_t *__restrict__ s, const int n,
                                          uint8_t *__restrict__ d, const int dlen,
                                          TablePtr table, const int lane) {
    const int bound = (int)compress_bound32((uint32_t)n);
    const bool not_compressible = dlen < bound;  // block.go:110
    {  // reset(): block.go:94,107
        uint4 *t4 = (uint4 *)(table);
        for (int k = lane; k < (1 << 16) * 2 / 16; k += kWave) t4[k] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();

    int si = 0, di = 0, anchor = 0;
    const int sn = n - kMfLimit;
    int ret = 0;
    bool failed = false;  // ErrInvalidSourceShortBuffer raised

    while (si < sn) {
        const uint64_t match = ld64(s + si);
        const uint32_t h0 = hash6(match);
        const uint32_t h1 = hash6(match >> 8);
        // both gets precede both puts: block.go:134-137
        const int ref0 = fast_get(table, h0, si);
        const int ref1 = fast_get(table, h1, si + 1);
        table[h0] = (uint16_t)si;
        table[h1] = (uint16_t)(si + 1);

        int found, ref;
        if (fast_ok(s, si, ref0, (uint32_t)match)) {
            found = si;
            ref = ref0;
        } else {
            const uint32_t h2 = hash6(match >> 16);
            const int ref2 = fast_get(table, h2, si + 2);  // sees the two puts: :145
            if (fast_ok(s, si + 1, ref1, (uint32_t)(match >> 8))) {
                found = si + 1;
                ref = ref1;
            } else {
                table[h2] = (uint16_t)(si + 2);  // :155
                if (fast_ok(s, si + 2, ref2, (uint32_t)(match >> 16))) {
                    found = si + 2;
                    ref = ref2;
                } else {
                    si += 4 + ((si + 2 - anchor) >> kAdaptSkipLog);  // :159
                    continue;
                }
            }
        }
        const int offset = found - ref;

        // ---- backward extension, block.go:170-177, 64 bytes per round ----
        int back = 0;
        {
            const int l0 = found - anchor;
            const int maxb = l0 < ref ? l0 : ref;  // lLen > 0 && tOff >= 0
            while (back < maxb) {
                const int k = back + lane;
                const bool eq = k < maxb && s[found - 1 - k] == s[ref - 1 - k];
                const uint64_t stop = ballot(!eq);
                if (stop == 0) {
                    back += kWave;
                    continue;
                }
                back += __builtin_ctzll(stop);
                break;
            }
        }
        const int m_start = found - back;
        const int l_len = m_start - anchor;

        // ---- forward extension, block.go:184-193: 8-byte strides while si+8 <= sn ----
        si = found + kMinMatch;
        for (;;) {
            const int p = si + 8 * lane;
            const bool valid = p + 8 <= sn;
            uint64_t x = 0;
            if (valid) x = ld64(s + p) ^ ld64(s + p - offset);
            const uint64_t stop = ballot(!valid || x != 0);
            if (stop == 0) {
                si += 8 * kWave;
                continue;
            }
            const int k = __builtin_ctzll(stop);
            const uint32_t xlo = (uint32_t)__shfl((int)(uint32_t)x, k);
            const uint32_t xhi = (uint32_t)__shfl((int)(uint32_t)(x >> 32), k);
            const uint64_t xk = ((uint64_t)xhi << 32) | xlo;
            si += 8 * k;
            if (xk != 0) si += __builtin_ctzll(xk) >> 3;  // lane k was valid and differed
            break;
        }
        int m_len = si - (m_start + kMinMatch);

        // ---- emit, block.go:195-248 ----
        if (di >= dlen) {
            failed = true;
            break;
        }
        if (lane == 0)
            d[di] = (uint8_t)((m_len < 0xF ? m_len : 0xF) | ((l_len < 0xF ? l_len : 0xF) << 4));
        di++;
        if (l_len >= 0xF) {
            di = emit_len_tail(d, di, dlen, l_len - 0xF, lane);
            if (di < 0) {
                failed = true;
                break;
            }
        }
        if (di + l_len > dlen) {
            failed = true;
            break;
        }
        wave_copy(d + di, s + anchor, l_len, lane);
        di += l_len + 2;
        anchor = si;
        if (di > dlen) {
            failed = true;
            break;
        }
        if (lane == 0) {
            d[di - 2] = (uint8_t)offset;
            d[di - 1] = (uint8_t)(offset >> 8);
        }
        if (m_len >= 0xF) {
            di = emit_len_tail(d, di, dlen, m_len - 0xF, lane);
            if (di < 0) {
                failed = true;
                break;
            }
        }
        if (si >= sn) break;                                   // :250-252
        table[hash6(ld64(s + si - 2))] = (uint16_t)(si - 2);  // :254-255
    }

    // ---- last literals, block.go:258-294 ----
    if (failed) {
        ret = -1;
    } else if (not_compressible && anchor == 0) {
        ret = 0;
    } else if (di >= dlen) {
        ret = -1;
    } else {
        const int l_len = n - anchor;
        if (lane == 0) d[di] = (uint8_t)((l_len < 0xF ? l_len : 0xF) << 4);
        di++;
        if (l_len >= 0xF) di = emit_len_tail(d, di, dlen, l_len - 0xF, lane);
        if (di < 0) {
            ret = -1;
        } else if (not_compressible && di >= anchor) {
            ret = 0;
        } else if (di + l_len > dlen) {
            ret = -1;
        } else {
            wave_copy(d + di, s + anchor, l_len, lane);
            ret = di + l_len;
        }
    }
    return ret;
}

__device__ __forceinline__ int block_dlen(int dst_mode, int n, uint32_t dst_len_x) {
    return dst_mode == LZ4HIP_DST_FRAME  ? n
           : dst_mode == LZ4HIP_DST_BOUND ? (int)compress_bound32((uint32_t)n)
                                          : (int)dst_len_x;
}

// Variant A: table in LDS, one workgroup (= one wave) per block, one block per CU.
__global__ __launch_bounds__(64) void k_fast_compress(const uint8_t *__restrict__ src,
                                                      size_t n_total, uint32_t block_size,
                                                      int dst_mode, uint32_t dst_len_x,
                                                      uint8_t *__restrict__ dst,
                                                      size_t dst_stride,
                                                      int32_t *__restrict__ result) {
    __shared__ uint16_t table[1 << 16];
    const int lane = lane_id();
    const size_t b = blockIdx.x;
    const size_t start = b * (size_t)block_size;
    const size_t left = n_total > start ? n_total - start : 0;
    const int n = (int)(left < block_size ? left : block_size);
    const int r = fast_block(src + start, n, dst + b * dst_stride,
                             block_dlen(dst_mode, n, dst_len_x), table, lane);
    if (lane == 0) result[b] = r;
}

// Variant B: table in a global-memory slot per resident wave (stays in L2 / the
// Infinity Cache), many waves per CU hide the dependent-load latency of the walk.
__global__ __launch_bounds__(64) void k_fast_compress_gtab(const uint8_t *__restrict__ src,
                                                           size_t n_total, uint32_t block_size,
                                                           size_t nb, int dst_mode,
                                                           uint32_t dst_len_x,
                                                           uint8_t *__restrict__ dst,
                                                           size_t dst_stride,
                                                           int32_t *__restrict__ result,
                                                           uint16_t *__restrict__ ws) {
    const int lane = lane_id();
    uint16_t *table = ws + (size_t)blockIdx.x * (1 << 16);
    for (size_t b = blockIdx.x; b < nb; b += gridDim.x) {
        const size_t start = b * (size_t)block_size;
        const size_t left = n_total > start ? n_total - start : 0;
        const int n = (int)(left < block_size ? left : block_size);
        const int r = fast_block(src + start, n, dst + b * dst_stride,
                                 block_dlen(dst_mode, n, dst_len_x), table, lane);
        if (lane == 0) result[b] = r;
    }
}

hipError_t launch_fast(const uint8_t *src, size_t n, uint32_t bs, size_t nb, int dst_mode,
                       uint32_t dst_len_x, uint8_t *dst, size_t dst_stride, int32_t *result,
                       uint16_t *ws, size_t slots, hipStream_t st) {
    if (nb == 0) return hipSuccess;
    if (ws && slots) {
        const size_t grid = nb < slots ? nb : slots;
        hipLaunchKernelGGL(k_fast_compress_gtab, dim3((unsigned)grid), dim3(kWave), 0, st, src, n,
                           bs, nb, dst_mode, dst_len_x, dst, dst_stride, result, ws);
    } else {
        hipLaunchKernelGGL(k_fast_compress, dim3((unsigned)nb), dim3(kWave), 0, st, src, n, bs,
                           dst_mode, dst_len_x, dst, dst_stride, result);
    }
    return hipGetLastError();
}

}  // namespace lz4hip
