// decode_common.h -- helpers of the window-parallel decoder (decode.hip).
#pragma once
#include "common.h"

namespace lz4hip {
namespace {

constexpr int kStage = 2048;  // bytes of LDS output stage per wave
constexpr int kIn = 2048;     // bytes of LDS compressed-input buffer per wave
constexpr int kReach = 352;   // pass 1 and the literal copies of one window read < si + kReach

// One wave per workgroup: LDS operations of a wave execute in order, so cross-lane
// hand-offs through the stage only need the compiler to keep program order and the
// data to have landed.
__device__ __forceinline__ void lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__device__ __forceinline__ uint32_t rdl(uint32_t v, int l) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, l);
}

// 255-run length extension starting at in[p] (decode_other.go:62-71,96-105).
__device__ __forceinline__ bool d2_len_ext(const uint8_t *in, int n, int &p, int &len, int lane) {
    for (;;) {
        const int idx = p + lane;
        const bool valid = idx < n;
        const uint32_t x = valid ? in[idx] : 0u;
        const uint64_t stop = ballot(!valid || x != 255u);
        if (stop == 0) {
            len += 255 * kWave;
            p += kWave;
            if (len < 0) return false;
            continue;
        }
        const int k = __builtin_ctzll(stop);
        if (p + k >= n) return false;
        len += 255 * k + (int)rdl(x, k);
        p += k + 1;
        return len >= 0;
    }
}

// One sequence, wave-uniform, every check of the reference in its order.
// Returns 0 = continue, 1 = block finished, -1 = error.  `prefix` bytes of history end
// at hist_end (== d: they sit right in front of d; else a separate dictionary).
__device__ __forceinline__ int d2_serial_sequence(const uint8_t *__restrict__ in, int n, uint8_t *d,
                                                  int dlen, int prefix, const uint8_t *hist_end,
                                                  int &si, int &di, int lane) {
    const uint32_t tok = in[si++];
    int l_len = (int)(tok >> 4);
    if (l_len > 0) {
        if (l_len == 0xF && !d2_len_ext(in, n, si, l_len, lane)) return -1;
        if (l_len > n - si || l_len > dlen - di) return -1;
        wave_copy(d + di, in + si, l_len, lane);
        si += l_len;
        di += l_len;
    }
    int m_len = (int)(tok & 0xF);
    if (si == n && m_len == 0) return 1;  // decode_other.go:81-82
    if (n - si < 2) return -1;            // :83-87
    const int offset = (int)in[si] | ((int)in[si + 1] << 8);
    if (offset == 0) return -1;  // :88-90
    si += 2;
    m_len += kMinMatch;
    if (m_len == kMinMatch + 0xF && !d2_len_ext(in, n, si, m_len, lane)) return -1;
    if (m_len > dlen - di) return -1;
    if (offset > di + prefix) return -1;  // before the dictionary start: :109-113
    if (offset > di && hist_end != d) {
        // decode_other.go:109-121: the first offset - di bytes come from the dictionary's
        // tail, the rest from dst[0:]; byte k of the match is byte k % offset of that source
        const int q0 = di - offset;  // < 0: position of the match source relative to d
        for (int k = lane; k < m_len; k += kWave) {
            const int q = q0 + (m_len <= offset ? k : k % offset);
            d[di + k] = q < 0 ? hist_end[q] : d[q];
        }
    } else {
        const uint8_t *from = d + di - offset;
        if (m_len <= offset) {
            wave_copy(d + di, from, m_len, lane);
        } else {
            for (int k = lane; k < m_len; k += kWave) d[di + k] = from[k % offset];
        }
    }
    di += m_len;
    return si < n ? 0 : 1;  // a block may end right after a match (decode_other.go:28)
}

// Exact-length store of the low `len` (< 8) bytes of v.
__device__ __forceinline__ void store_tail(uint8_t *p, uint64_t v, int len) {
    if (len & 4) {
        const uint32_t w = (uint32_t)v;
        __builtin_memcpy(p, &w, 4);
        p += 4;
        v >>= 32;
    }
    if (len & 2) {
        const uint16_t w = (uint16_t)v;
        __builtin_memcpy(p, &w, 2);
        p += 2;
        v >>= 16;
    }
    if (len & 1) *p = (uint8_t)v;
}

// Lane-private copy of len bytes; the caller guarantees that 16 bytes may be read
// at src + 16*k for every chunk started (over-read, never over-write).
__device__ __forceinline__ void lane_copy16(uint8_t *dst, const uint8_t *src, int len) {
    for (;;) {
        const uint64_t a = ld64(src), b = ld64(src + 8);
        if (len >= 16) {
            __builtin_memcpy(dst, &a, 8);
            __builtin_memcpy(dst + 8, &b, 8);
            len -= 16;
            if (len == 0) return;
            dst += 16;
            src += 16;
            continue;
        }
        if (len >= 8) {
            __builtin_memcpy(dst, &a, 8);
            store_tail(dst + 8, b, len - 8);
        } else {
            store_tail(dst, a, len);
        }
        return;
    }
}

// Exact-length copy of len bytes from global memory into the LDS stage (lane-private;
// reads exactly [src, src+len)).
// `room`: bytes that may be READ from src on (>= len; the rest of the block's own dst slot).
// The wave waits for its slowest lane, and a lane that fetches 16 bytes per round trip needs
// three of them for a 40-byte match: up to 32 bytes are fetched in ONE round trip (four loads
// in flight, over-reading inside `room`) and stored exactly.
__device__ __forceinline__ void g2s_exact(uint8_t *st, const uint8_t *src, int len, int room) {
    if (len <= 32 && room >= 32) {
        const uint64_t a0 = ld64(src), a1 = ld64(src + 8), a2 = ld64(src + 16), a3 = ld64(src + 24);
        uint64_t t = a0;
        if (len >= 8) {
            __builtin_memcpy(st, &a0, 8);
            t = a1;
            if (len >= 16) {
                __builtin_memcpy(st + 8, &a1, 8);
                t = a2;
                if (len >= 24) {
                    __builtin_memcpy(st + 16, &a2, 8);
                    t = a3;
                    if (len >= 32) __builtin_memcpy(st + 24, &a3, 8);
                }
            }
        }
        store_tail(st + (len & ~7), t, len & 7);
        return;
    }
    while (len >= 16) {
        const uint64_t a = ld64(src), b = ld64(src + 8);
        __builtin_memcpy(st, &a, 8);
        __builtin_memcpy(st + 8, &b, 8);
        st += 16;
        src += 16;
        len -= 16;
    }
    if (len & 8) {
        const uint64_t a = ld64(src);
        __builtin_memcpy(st, &a, 8);
        st += 8;
        src += 8;
    }
    if (len & 4) {
        const uint32_t a = ld32(src);
        __builtin_memcpy(st, &a, 4);
        st += 4;
        src += 4;
    }
    if (len & 2) {
        uint16_t a;
        __builtin_memcpy(&a, src, 2);
        __builtin_memcpy(st, &a, 2);
        st += 2;
        src += 2;
    }
    if (len & 1) *st = *src;
}

// Literals: 16 bytes may be READ at src + 16*k for every chunk started (pass 1
// guarantees it); the stage is written exactly.
__device__ __forceinline__ void g2s_over(uint8_t *st, const uint8_t *src, int len) {
    for (;;) {
        const uint64_t a = ld64(src), b = ld64(src + 8);
        if (len >= 16) {
            __builtin_memcpy(st, &a, 8);
            __builtin_memcpy(st + 8, &b, 8);
            len -= 16;
            if (len == 0) return;
            st += 16;
            src += 16;
            continue;
        }
        if (len >= 8) {
            __builtin_memcpy(st, &a, 8);
            store_tail(st + 8, b, len - 8);
        } else {
            store_tail(st, a, len);
        }
        return;
    }
}

// `prefix` bytes of history precede d in memory (dictionary laid out in front of dst).
// LDS -> LDS literal copy; 16 bytes may be read at src + 16*k for every chunk started.
__device__ __forceinline__ void s2s_over(uint8_t *st, const uint8_t *src, int len) {
    for (;;) {
        uint64_t a, b;
        __builtin_memcpy(&a, src, 8);
        __builtin_memcpy(&b, src + 8, 8);
        if (len >= 16) {
            __builtin_memcpy(st, &a, 8);
            __builtin_memcpy(st + 8, &b, 8);
            len -= 16;
            if (len == 0) return;
            st += 16;
            src += 16;
            continue;
        }
        if (len >= 8) {
            __builtin_memcpy(st, &a, 8);
            store_tail(st + 8, b, len - 8);
        } else {
            store_tail(st, a, len);
        }
        return;
    }
}

}  // namespace
}  // namespace lz4hip
