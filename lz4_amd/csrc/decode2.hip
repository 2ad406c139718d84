// decode2.hip -- two-pass, window-parallel LZ4 block decoder for gfx950.
//
// Same results as decodeBlock (internal/lz4block/decode_other.go:10-137,
// decode_amd64.s:24-448) / UncompressBlock (block.go:44-52).  One wavefront per
// independent block; per 64-byte window of the compressed stream:
//
//   pass 1  every lane parses the sequence header that would start at its byte
//           (token, one optional length byte each side, offset) and a short
//           wave-uniform walk links the real token positions; an exclusive
//           scan of (literals + match) lengths gives every sequence its output
//           offset, and every bound the reference checks is checked here.
//   pass 2  the token lanes copy their own literals, then the matches whose
//           source lies wholly before this window's output (16 bytes per load,
//           exact-length stores), all in flight together; matches that read
//           bytes produced inside the window follow in stream order, copied 64
//           lanes wide as the periodic extension of the `offset` bytes before
//           them (byte-serial overlap semantics).
//
// Anything unusual -- 255-run lengths, the last sequences of a block, an invalid
// offset, an output overrun -- drops to the serial sequence decoder below, which
// restates the reference check by check, so error behaviour is the reference's.
#include "common.h"

namespace lz4hip {

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
// Returns 0 = continue, 1 = block finished, -1 = error.
__device__ __forceinline__ int d2_serial_sequence(const uint8_t *__restrict__ in, int n, uint8_t *d,
                                                  int dlen, int prefix, int &si, int &di,
                                                  int lane) {
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
    const uint8_t *from = d + di - offset;
    if (m_len <= offset) {
        wave_copy(d + di, from, m_len, lane);
    } else {
        for (int k = lane; k < m_len; k += kWave) d[di + k] = from[k % offset];
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

// `prefix` bytes of history precede d in memory (dictionary laid out in front of dst).
__device__ int decode2_block(const uint8_t *__restrict__ in, const int n, uint8_t *d, const int dlen,
                             const int prefix, const int lane) {
    int si = 0, di = 0;
    for (;;) {
        si = (int)rfl((uint32_t)si);
        di = (int)rfl((uint32_t)di);
        if (si >= n) return di;

        // ---------------- pass 1: parse 64 candidate headers ----------------
        const int pos = si + lane;
        // 4 header bytes at pos and 16 readable literal bytes after the token
        bool ok = pos + 18 <= n;
        const uint32_t w0 = ok ? ld32(in + pos) : 0u;
        const uint32_t tok = w0 & 0xFFu;
        int lit = (int)(tok >> 4), e1 = 0;
        if (lit == 0xF) {
            const uint32_t b1 = (w0 >> 8) & 0xFFu;
            lit += (int)b1;
            e1 = 1;
            ok = ok && b1 != 255u;
        }
        const int lpos = pos + 1 + e1;  // first literal
        const int opos = lpos + lit;    // offset field
        ok = ok && opos + 4 <= n && (lit <= 16 || lpos + ((lit + 15) & ~15) <= n);
        const uint32_t w1 = ok ? ld32(in + opos) : 0u;
        const int off = (int)(w1 & 0xFFFFu);
        int ml = (int)(tok & 0xFu) + kMinMatch, e2 = 0;
        if ((tok & 0xFu) == 0xFu) {
            const uint32_t b2 = (w1 >> 16) & 0xFFu;
            ml += (int)b2;
            e2 = 1;
            ok = ok && b2 != 255u;
        }
        const int seqlen = opos + 2 + e2 - pos;  // compressed bytes of this sequence
        // another token must follow: the last sequences go through the serial path
        ok = ok && off != 0 && pos + seqlen < n;
        const int outlen = lit + ml;

        // ---------------- link the real tokens (wave-uniform) ----------------
        const uint64_t G = ballot(ok);
        uint64_t T = 0;
        int t = 0;
        while (t < 64 && ((G >> t) & 1)) {
            T |= 1ull << t;
            t += (int)rdl((uint32_t)seqlen, t);
        }
        const bool mine = (T >> lane) & 1;

        // exclusive scan of output lengths over the token lanes
        int incl = mine ? outlen : 0;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        const int myout = incl - (mine ? outlen : 0);  // offset of my literals from di
        const int mstart = myout + lit;                 // offset of my match from di
        // the reference's bounds: literals and match fit dst, offset inside history
        const bool bad = mine && (di + mstart + ml > dlen || off > di + mstart + prefix);
        const uint64_t B = ballot(bad);
        if (B) T &= (1ull << __builtin_ctzll(B)) - 1;  // stop before the first violation
        if (T == 0) {
            const int st = d2_serial_sequence(in, n, d, dlen, prefix, si, di, lane);
            if (st < 0) return -1;
            if (st > 0) return di;
            continue;
        }
        const bool act = (T >> lane) & 1;
        const int last = 63 - __builtin_clzll(T);
        const int adv_c = last + (int)rdl((uint32_t)seqlen, last);       // compressed bytes consumed
        const int adv_o = (int)rdl((uint32_t)(myout + outlen), last);   // output bytes produced

        // ---------------- pass 2: copies ----------------
        // literals (16 bytes readable after every token: ok => pos + 18 <= n)
        if (act && lit > 0) lane_copy16(d + di + myout, in + lpos, lit);
        // matches whose source ends before this window's output, and whose 16-byte
        // chunk reads stay inside dst
        const bool indep = act && off >= mstart + ml && di + mstart + ((ml + 15) & ~15) <= dlen;
        if (indep) lane_copy16(d + di + mstart, d + di + mstart - off, ml);
        // the rest in stream order, 64 lanes wide
        uint64_t D = ballot(act && !indep);
        while (D) {
            const int l = __builtin_ctzll(D);
            D &= D - 1;
            const int o_ms = (int)rdl((uint32_t)mstart, l);
            const int o_ml = (int)rdl((uint32_t)ml, l);
            const int o_off = (int)rdl((uint32_t)off, l);
            uint8_t *to = d + di + o_ms;
            const uint8_t *from = to - o_off;
            if (o_ml <= o_off) {
                wave_copy(to, from, o_ml, lane);
            } else {
                for (int k = lane; k < o_ml; k += kWave) to[k] = from[k % o_off];
            }
        }
        si += adv_c;
        di += adv_o;
    }
}

__global__ __launch_bounds__(64) void k_decode2(const uint8_t *__restrict__ src,
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
        r = decode2_block(in, n, d, (int)dst_len, (int)prefix, lane);
    }
    if (lane == 0) out_len[b] = r;
}

// Linked-block frame (FLG bit 5 clear): every block may reference the 64 KiB of
// output before it (reader.go:176-189 keeps that history as the dictionary), so the
// blocks of one frame decode in stream order by ONE wave into one contiguous buffer,
// where "the dictionary" is simply the output already written: prefix = bytes so far.
// The reader's history holds at least min(bytes so far, 64 KiB) and offsets are
// < 64 KiB, so `offset > di + len(dict)` fires exactly when `offset > di + prefix`.
// out_off[b] = start of block b's output, out_len[b] = its length or -1; the walk
// stops at the first bad block (later blocks report -1).
__global__ __launch_bounds__(64) void k_decode2_linked(const uint8_t *__restrict__ src,
                                                       const uint64_t *__restrict__ off,
                                                       const uint32_t *__restrict__ sw, size_t nb,
                                                       uint8_t *dst, uint32_t block_cap,
                                                       size_t dst_cap,
                                                       int32_t *__restrict__ out_len,
                                                       uint64_t *__restrict__ out_off) {
    const int lane = lane_id();
    size_t total = 0;
    bool dead = false;
    for (size_t b = 0; b < nb; b++) {
        int r = -1;
        if (!dead) {
            const uint32_t w = sw[b];
            const int n = (int)(w & 0x7FFFFFFFu);
            const uint8_t *in = src + off[b];
            uint8_t *d = dst + total;
            // never past the stream buffer: the block's room is what is left of it
            const size_t left = dst_cap - total;
            const uint32_t cap = left < block_cap ? (uint32_t)left : block_cap;
            if (w >> 31) {
                r = (uint32_t)n > cap ? (int)cap : n;
                if ((uint32_t)n > cap && cap < block_cap) r = -1;  // cut by the buffer, not by the block size
                else wave_copy(d, in, r, lane);
            } else if (n == 0) {
                r = 0;
            } else {
                const int prefix = total > (size_t)kWinSize ? kWinSize : (int)total;
                r = decode2_block(in, n, d, (int)cap, prefix, lane);
            }
            r = (int)rfl((uint32_t)r);
            // this wave's stores must be visible to its own later loads
            __threadfence();
        }
        if (lane == 0) {
            out_len[b] = r;
            out_off[b] = total;
        }
        if (r < 0) dead = true;
        else total += (size_t)r;
    }
}

hipError_t launch_decode2_linked(const uint8_t *src, const uint64_t *off, const uint32_t *sw,
                                 size_t nb, uint8_t *dst, uint32_t block_cap, size_t dst_cap,
                                 int32_t *out_len, uint64_t *out_off, hipStream_t st) {
    if (nb == 0) return hipSuccess;
    hipLaunchKernelGGL(k_decode2_linked, dim3(1), dim3(kWave), 0, st, src, off, sw, nb, dst,
                       block_cap, dst_cap, out_len, out_off);
    return hipGetLastError();
}

hipError_t launch_decode2(const uint8_t *src, const uint64_t *off, const uint32_t *sw, size_t nb,
                          uint8_t *dst, size_t dst_stride, uint32_t dst_len, uint32_t prefix,
                          int32_t *out_len, hipStream_t st) {
    if (nb == 0) return hipSuccess;
    hipLaunchKernelGGL(k_decode2, dim3((unsigned)nb), dim3(kWave), 0, st, src, off, sw, nb, dst,
                       dst_stride, dst_len, prefix, out_len);
    return hipGetLastError();
}

}  // namespace lz4hip
