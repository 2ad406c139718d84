// decode3.hip -- window-parallel LZ4 block decoder, generation 3: decode2's two passes
// with the window's output assembled in LDS.
//
// Same results as decodeBlock (internal/lz4block/decode_other.go:10-137,
// decode_amd64.s:24-448) / UncompressBlock (block.go:44-52).  decode2 lets every token
// lane store its own literals and match with exact-length global stores: ~6 small
// requests per 12-byte sequence, and matches that read bytes produced inside the
// window pay a global round trip each, in stream order.  Here
//   * literals and the part of every match whose source lies before the window's
//     output are loaded lane-privately (all in flight together) into an LDS stage;
//   * the rest of the dependent matches is resolved inside the stage, in stream
//     order, at LDS latency;
//   * the stage is flushed to dst 16 bytes per lane, contiguous across the wave.
// Windows whose output exceeds the stage, and everything unusual (255-run lengths,
// the last sequences, errors), take decode2's paths unchanged.
#include "decode_common.h"

namespace lz4hip {

namespace {

// `stage`: kStage bytes of LDS owned by this wave, 16-byte aligned.
// `ibuf`: kIn bytes of LDS; holds in[ibase, iend), refilled with one coalesced load
// whenever the next window could read past it, so that the header parse and the
// literal copies of ~25 windows cost one global round trip instead of two each.
__device__ int decode3_block(const uint8_t *__restrict__ in, const int n, uint8_t *d, const int dlen,
                             const int prefix, const int lane, uint8_t *stage, uint8_t *ibuf) {
    int si = 0, di = 0, ibase = 0, iend = 0;
    for (;;) {
        si = (int)rfl((uint32_t)si);
        di = (int)rfl((uint32_t)di);
        if (si >= n) return di;
        ibase = (int)rfl((uint32_t)ibase);
        iend = (int)rfl((uint32_t)iend);
        if (si + kReach > iend && iend < n) {
            ibase = si;
            const int cnt = n - ibase < kIn ? n - ibase : kIn;
            lds_sync();
            for (int k = lane * 16; k < cnt; k += kWave * 16) {
                if (k + 16 <= cnt) {
                    const uint64_t a = ld64(in + ibase + k), b = ld64(in + ibase + k + 8);
                    *(uint64_t *)(ibuf + k) = a;
                    *(uint64_t *)(ibuf + k + 8) = b;
                } else {
                    for (int j = k; j < cnt; j++) ibuf[j] = in[ibase + j];
                }
            }
            iend = ibase + cnt;
            lds_sync();
        }
        const uint8_t *ib = ibuf - ibase;  // ib[p] = in[p] for ibase <= p < iend

        // ---------------- pass 1: parse 64 candidate headers ----------------
        const int pos = si + lane;
        // 4 header bytes at pos and 16 readable literal bytes after the token
        bool ok = pos + 18 <= n;
        uint32_t w0 = 0u;
        if (ok) __builtin_memcpy(&w0, ib + pos, 4);
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
        uint32_t w1 = 0u;
        if (ok) __builtin_memcpy(&w1, ib + opos, 4);
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
        if (adv_o <= kStage) {
            lds_sync();  // the previous window's flush has read the stage
            // literals (16 bytes readable after every token: ok => pos + 18 <= n)
            if (act && lit > 0) s2s_over(stage + myout, ib + lpos, lit);
            // leading g bytes of every match come from output older than this window
            int g = 0;
            if (act) {
                g = off - mstart;
                g = g < 0 ? 0 : (g > ml ? ml : g);
                if (g > 0) g2s_exact(stage + mstart, d + di + mstart - off, g);
            }
            lds_sync();
            // the rest reads this window's own output: stream order, inside the stage
            uint64_t D = ballot(act && g < ml);
            while (D) {
                const int l = __builtin_ctzll(D);
                D &= D - 1;
                const int o_ms = (int)rdl((uint32_t)mstart, l);
                const int o_ml = (int)rdl((uint32_t)ml, l);
                const int o_off = (int)rdl((uint32_t)off, l);
                const int o_g = (int)rdl((uint32_t)g, l);
                // first period (or the whole match): sources end before the match starts
                const int p1 = o_ml < o_off ? o_ml : o_off;
                for (int k = o_g + lane; k < p1; k += kWave) stage[o_ms + k] = stage[o_ms + k - o_off];
                lds_sync();
                // overlap: periodic extension of the first `offset` bytes of the match
                if (o_ml > o_off) {
                    for (int k = o_off + lane; k < o_ml; k += kWave)
                        stage[o_ms + k] = stage[o_ms + k % o_off];
                    lds_sync();
                }
            }
            // flush: 16 bytes per lane, contiguous across the wave
            uint8_t *to = d + di;
            for (int k = lane * 16; k < adv_o; k += kWave * 16) {
                const uint64_t a = *(const uint64_t *)(stage + k), b = *(const uint64_t *)(stage + k + 8);
                const int rem = adv_o - k;
                if (rem >= 16) {
                    __builtin_memcpy(to + k, &a, 8);
                    __builtin_memcpy(to + k + 8, &b, 8);
                } else if (rem >= 8) {
                    __builtin_memcpy(to + k, &a, 8);
                    store_tail(to + k + 8, b, rem - 8);
                } else {
                    store_tail(to + k, a, rem);
                }
            }
        } else {
        // (window output larger than the stage: decode2's direct copies)
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
        }
        si += adv_c;
        di += adv_o;
    }
}

}  // namespace

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_decode3(const uint8_t *__restrict__ src,
                                                const uint64_t *__restrict__ off,
                                                const uint32_t *__restrict__ sw, size_t nb,
                                                uint8_t *dst, size_t dst_stride, uint32_t dst_len,
                                                uint32_t prefix, int32_t *__restrict__ out_len) {
    __shared__ __attribute__((aligned(16))) uint8_t stage[kStage + 16];
    __shared__ __attribute__((aligned(16))) uint8_t ibuf[kIn + 16];
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
        wave_copy16(d, in, r, lane);
    } else if (n == 0) {
        r = 0;  // block.go:45-47
    } else {
        r = decode3_block(in, n, d, (int)dst_len, (int)prefix, lane, stage, ibuf);
    }
    if (lane == 0) out_len[b] = r;
}

hipError_t launch_decode3(const uint8_t *src, const uint64_t *off, const uint32_t *sw, size_t nb,
                          uint8_t *dst, size_t dst_stride, uint32_t dst_len, uint32_t prefix,
                          int32_t *out_len, hipStream_t st) {
    if (nb == 0) return hipSuccess;
    hipLaunchKernelGGL(k_decode3, dim3((unsigned)nb), dim3(kWave), 0, st, src, off, sw, nb, dst,
                       dst_stride, dst_len, prefix, out_len);
    return hipGetLastError();
}

}  // namespace lz4hip
