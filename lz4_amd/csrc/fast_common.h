// fast_common.h -- helpers shared by the window-parallel Fast kernels (fast2/fast3).
#pragma once
#include "common.h"

namespace lz4hip {

#ifndef LZ4HIP_TAGN
#define LZ4HIP_TAGN 512
#endif
constexpr int kTagN = LZ4HIP_TAGN;  // LDS tag slots per wave (u32 each)

__device__ __forceinline__ uint32_t rdlane(uint32_t v, int l) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, l);
}

// Cooperative backward extension from `back` (block.go:170-177), 64 bytes per round.
__device__ __forceinline__ int ext_backward(const uint8_t *s, int found, int ref, int back,
                                            int maxb, int lane) {
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
    return back < maxb ? back : maxb;
}

// Cooperative forward extension from si (block.go:184-193): 8-byte strides while si+8 <= sn.
__device__ __forceinline__ int ext_forward(const uint8_t *s, int si, int offset, int sn, int lane) {
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
        const int k = __builtin_ctzll(stop);  // wave-uniform (SGPR)
        const uint32_t xlo = rdlane((uint32_t)x, k);
        const uint32_t xhi = rdlane((uint32_t)(x >> 32), k);
        const uint64_t xk = ((uint64_t)xhi << 32) | xlo;
        si += 8 * k;
        if (xk != 0) si += __builtin_ctzll(xk) >> 3;
        return si;
    }
}

// Table read of the speculative gather.  For a table in global memory the load
// bypasses the CU's vector L1 (sc1, served by L2): the wave's own recent 2-byte
// stores are written through to L2, but an L1-resident copy of the line is not
// guaranteed to be refreshed before a later gather hits it.
template <bool kGlobal>
__device__ __forceinline__ uint32_t tab_load(const uint16_t *t, uint32_t h) {
    if (kGlobal) return __hip_atomic_load(t + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return t[h];
}

__device__ __forceinline__ int block_dlen2(int dst_mode, int n, uint32_t dst_len_x) {
    return dst_mode == LZ4HIP_DST_FRAME  ? n
           : dst_mode == LZ4HIP_DST_BOUND ? (int)compress_bound32((uint32_t)n)
                                          : (int)dst_len_x;
}

}  // namespace lz4hip
