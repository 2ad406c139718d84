// hc.hip -- HC (hash-chain) LZ4 block compressor for gfx950.
//
// Restates (*CompressorHC).CompressBlock, internal/lz4block/block.go:319-481:
// same 4-byte Knuth hash (:298-301), same hashTable/chainTable contents
// (absolute positions, 0 = empty), same chain order, same depth accounting, so
// the chosen match -- longest capped prefix >= 4, ties to the nearest -- and every
// output byte are the reference's.  One wavefront per block; tables live in a
// per-workgroup slot of the caller's workspace (2 x 65536 x u32).
//   * Chain candidates are collected 64 at a time (one per lane) and their
//     prefix lengths are computed by the lanes in parallel; a wave max-reduction
//     resolves to the reference's winner.  The 1-byte gate of :358-360 is a pure
//     filter on that winner and is not needed.
//   * Emission (token, length runs, literal copy) is 64 lanes wide.
#include "common.h"

namespace lz4hip {

constexpr int kHcSlotWords = 2 * (1 << 16);

// Prefix length of block.go:361-372 for one candidate (lane-private loop).
__device__ __forceinline__ int hc_prefix_len(const uint8_t *s, int next, int si, int lim) {
    int ml = 0;
    while (ml < lim) {
        const uint64_t x = ld64(s + next + ml) ^ ld64(s + si + ml);
        if (x == 0) {
            ml += 8;
        } else {
            ml += __builtin_ctzll(x) >> 3;
            break;
        }
    }
    return ml;
}

__device__ __forceinline__ int wave_max(int v) {
    for (int o = 32; o > 0; o >>= 1) {
        const int t = __shfl_xor(v, o);
        v = t > v ? t : v;
    }
    return v;
}

__global__ __launch_bounds__(64) void k_hc_compress(const uint8_t *__restrict__ src,
                                                    size_t n_total, uint32_t block_size,
                                                    uint32_t depth, int dst_mode, uint32_t dst_len_x,
                                                    uint8_t *__restrict__ dst, size_t dst_stride,
                                                    int32_t *__restrict__ result,
                                                    uint32_t *__restrict__ ws, size_t nb) {
    const int lane = lane_id();
    uint32_t *hash_table = ws + (size_t)blockIdx.x * kHcSlotWords;
    uint32_t *chain_table = hash_table + (1 << 16);
    if (depth == 0) depth = kWinSize;  // block.go:343-345

    for (size_t b = blockIdx.x; b < nb; b += gridDim.x) {
        const uint8_t *s = src + b * (size_t)block_size;
        const size_t start = b * (size_t)block_size;
        const size_t left = n_total > start ? n_total - start : 0;
        const int n = (int)(left < block_size ? left : block_size);
        uint8_t *d = dst + b * dst_stride;
        const int bound = (int)compress_bound32((uint32_t)n);
        const int dlen = dst_mode == LZ4HIP_DST_FRAME ? n
                         : dst_mode == LZ4HIP_DST_BOUND ? bound : (int)dst_len_x;
        const bool not_compressible = dlen < bound;  // :330

        {  // zero both tables: :320-324
            uint4 *t4 = reinterpret_cast<uint4 *>(hash_table);
            for (int k = lane; k < kHcSlotWords / 4; k += kWave) t4[k] = make_uint4(0, 0, 0, 0);
        }
        __syncthreads();

        int si = 0, di = 0, anchor = 0;
        const int sn = n - kMfLimit;
        bool failed = false;

        while (si < sn) {
            uint32_t match = ld32(s + si);
            const uint32_t h = hash4(match);
            int m_len = 0, offset = 0;
            {
                const int lim = sn - si;
                int next = (int)hash_table[h];
                uint32_t tries = depth;
                // A prefix length never exceeds lim rounded up to 8 (:363-372) and only a
                // strictly longer candidate replaces the winner (:373), so the walk can
                // stop once that maximum is reached: the rest of the chain changes nothing.
                const int ml_max = (lim + 7) & ~7;
                while (tries > 0 && next > 0 && si - next < kWinSize && m_len < ml_max) {
                    // collect up to 64 chain entries, one per lane
                    int mine = 0, cnt = 0;
                    while (cnt < kWave && tries > 0 && next > 0 && si - next < kWinSize) {
                        if (lane == cnt) mine = next;
                        cnt++;
                        tries--;
                        next = (int)chain_table[next & kWinMask];
                    }
                    int ml = 0;
                    if (lane < cnt) ml = hc_prefix_len(s, mine, si, lim);
                    // winner: max ml, ties to the earliest chain entry (lowest lane)
                    const int key = ml >= kMinMatch ? (ml << 6) | (63 - lane) : 0;
                    const int best = wave_max(key);
                    if (best != 0 && (best >> 6) > m_len) {  // strictly longer: :373-379
                        m_len = best >> 6;
                        offset = si - __shfl(mine, 63 - (best & 63));
                    }
                }
            }
            if (lane == 0) {
                chain_table[si & kWinMask] = hash_table[h];
                hash_table[h] = (uint32_t)si;
            }
            __syncthreads();

            if (m_len == 0) {  // :386-389
                si += 1 + ((si - anchor) >> kAdaptSkipLog);
                continue;
            }

            {  // insert every position covered by the match: :394-405
                int p = si + 1;
                const int wsn = si + m_len - kWinSize;
                if (wsn > p) p = wsn;
                const int end = si + m_len;
                if (lane == 0) {
                    // The reference rolls the 4-byte value from the one at si
                    // (:399-400).  When the window clamp skips positions (match
                    // longer than 64 KiB) the first three values mix old and new
                    // bytes; reproduce that, it decides which slots are written.
                    for (; p < end; p++) {
                        match = (match >> 8) | ((uint32_t)s[p + 3] << 24);
                        const uint32_t hh = hash4(match);
                        chain_table[p & kWinMask] = hash_table[hh];
                        hash_table[hh] = (uint32_t)p;
                    }
                }
                __syncthreads();
            }

            const int l_len = si - anchor;
            si += m_len;
            m_len -= kMinMatch;

            // ---- emit, block.go:411-449; an index panic is "failed" ----
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
            di += l_len;
            anchor = si;
            di += 2;
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
        }

        // ---- tail, block.go:452-480.  For sn <= 0 the anchor==0 early-out is
        // skipped (goto lastLiterals, :339-341). ----
        int ret;
        if (failed) {
            ret = -1;
        } else if (sn > 0 && not_compressible && anchor == 0) {
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
        if (lane == 0) result[b] = ret;
        __syncthreads();
    }
}

hipError_t launch_hc(const uint8_t *src, size_t n, uint32_t bs, size_t nb, uint32_t depth,
                     int dst_mode,
                     uint32_t dst_len_x, uint8_t *dst, size_t dst_stride, int32_t *result, uint32_t *ws,
                     size_t slots, hipStream_t st) {
    if (nb == 0) return hipSuccess;
    const size_t grid = nb < slots ? nb : slots;
    hipLaunchKernelGGL(k_hc_compress, dim3((unsigned)grid), dim3(kWave), 0, st, src, n, bs, depth,
                       dst_mode, dst_len_x, dst, dst_stride, result, ws, nb);
    return hipGetLastError();
}

}  // namespace lz4hip
