// decode.hip -- LZ4 block decoder for gfx950.
//
// Restates decodeBlock, internal/lz4block/decode_other.go:10-137 (semantics) /
// decode_amd64.s:24-448, and UncompressBlock, block.go:44-52.  One wavefront
// per independent block.  The sequence walk is wave-uniform; length runs,
// literal copies and match copies are 64 lanes wide.  A match is written as
// the periodic extension of the `offset` bytes before it, which equals the
// reference's byte-serial overlap copy (decode_amd64.s:290-300) and has no
// dependency inside the match.  Every bound is checked before a byte is
// stored, so nothing is ever written at or beyond dst_len.
#include "common.h"

namespace lz4hip {

// Adds the 255-run length extension that starts at in[p]; returns false when the
// input ends inside it (decode_other.go:62-71,96-105 index panics).
__device__ __forceinline__ bool read_len_ext(const uint8_t *in, int n, int &p, int &len, int lane) {
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
        len += 255 * k + (int)__shfl((int)x, k);
        p += k + 1;
        return len >= 0;
    }
}

// Returns bytes written or -1.
// `prefix` bytes of history precede d in memory (external dictionary laid out in
// front of dst: decode_other.go:109-121); 0 in the batched path.
__device__ int decode_block_wave(const uint8_t *__restrict__ in, int n, uint8_t *d, int dlen,
                                 int prefix, int lane) {
    int si = 0, di = 0;
    while (si < n) {
        const uint32_t tok = in[si++];
        int l_len = (int)(tok >> 4);
        if (l_len > 0) {
            if (l_len == 0xF && !read_len_ext(in, n, si, l_len, lane)) return -1;
            if (l_len > n - si || l_len > dlen - di) return -1;
            wave_copy(d + di, in + si, l_len, lane);
            si += l_len;
            di += l_len;
        }
        int m_len = (int)(tok & 0xF);
        if (si == n && m_len == 0) break;  // decode_other.go:81-82
        if (n - si < 2) return -1;         // :83-87
        const int offset = (int)in[si] | ((int)in[si + 1] << 8);
        if (offset == 0) return -1;  // :88-90
        si += 2;
        m_len += kMinMatch;
        if (m_len == kMinMatch + 0xF && !read_len_ext(in, n, si, m_len, lane)) return -1;
        if (m_len > dlen - di) return -1;
        if (offset > di + prefix) return -1;  // before the dictionary start: :109-113
        const uint8_t *from = d + di - offset;
        if (m_len <= offset) {
            wave_copy(d + di, from, m_len, lane);
        } else {
            for (int k = lane; k < m_len; k += kWave) d[di + k] = from[k % offset];
        }
        di += m_len;
    }
    return di;
}

__global__ __launch_bounds__(64) void k_decode(const uint8_t *__restrict__ src,
                                               const uint64_t *__restrict__ off,
                                               const uint32_t *__restrict__ sw, size_t nb,
                                               uint8_t *dst, size_t dst_stride, uint32_t dst_len,
                                               uint32_t prefix, int32_t *__restrict__ out_len) {
    const int lane = lane_id();
    const size_t b = blockIdx.x;
    if (b >= nb) return;
    const uint32_t w = sw[b];
    const int n = (int)(w & 0x7FFFFFFFu);
    const uint8_t *in = src + off[b];
    uint8_t *d = dst + b * dst_stride;
    int r;
    if (w >> 31) {  // stored raw: n := copy(dst, b.data), lz4stream/block.go:320-322
        r = (uint32_t)n > dst_len ? (int)dst_len : n;
        wave_copy(d, in, r, lane);
    } else if (n == 0) {
        r = 0;  // block.go:45-47
    } else {
        r = decode_block_wave(in, n, d, (int)dst_len, (int)prefix, lane);
    }
    if (lane == 0) out_len[b] = r;
}

hipError_t launch_decode(const uint8_t *src, const uint64_t *off, const uint32_t *sw, size_t nb,
                         uint8_t *dst, size_t dst_stride, uint32_t dst_len, uint32_t prefix,
                         int32_t *out_len, hipStream_t st) {
    if (nb == 0) return hipSuccess;
    hipLaunchKernelGGL(k_decode, dim3((unsigned)nb), dim3(kWave), 0, st, src, off, sw, nb, dst,
                       dst_stride, dst_len, prefix, out_len);
    return hipGetLastError();
}

}  // namespace lz4hip
