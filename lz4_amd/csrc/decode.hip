// decode.hip -- window-parallel two-pass LZ4 block decoder.
//
// Same results as decodeBlock (internal/lz4block/decode_other.go:10-137,
// decode_amd64.s:24-448) / UncompressBlock (block.go:44-52).
//
// A 64-byte window holds ~9 sequences on text, so a pass 2 over it (the copies, with
// their divergent tails) would run with 9 of 64 lanes active and every per-window cost
// (offset scan, bounds ballots, stage hand-offs, flush) would be paid per ~105 output
// bytes: instruction issue, not memory, bounds this kernel.  So
//   * pass 1 parses kSets candidate headers per lane (256 compressed bytes),
//   * the linked tokens are compacted through LDS into lanes 0..ntok-1 in stream
//     order (a sequence takes at least 3 bytes: three sets hold at most 64 tokens, and
//     the link stops before a fourth set that could overflow the 64 lanes), and
//   * the scan, the bounds and pass 2 run ONCE over the compacted lanes (~36 active on
//     text), i.e. a quarter of the per-window instructions per output byte.
// Unusual sequences (255-run lengths, the block tail, errors) take the same serial
// path as before; a window whose first byte is not a plain token skips the wide parse.
#include "decode_common.h"

namespace lz4hip {

namespace {

#ifndef LZ4HIP_DSETS
#define LZ4HIP_DSETS 5
#endif
constexpr int kSets = LZ4HIP_DSETS;       // candidate positions per lane
constexpr int kWin4 = kSets * kWave;      // compressed bytes per window
constexpr int kReach4 = kWin4 + 288;      // pass 1 and the literal copies read < si + kReach4
// A sequence takes at least 3 compressed bytes: 64 positions hold at most 22 tokens, three
// sets at most 64.  With more sets the link stops before a set that could overflow the
// 64 token lanes.
constexpr int kSetTokens = (kWave + 2) / 3;
static_assert(kSets <= 8, "the token descriptor is a 16-bit window-relative position");
static_assert(kReach4 + 16 <= kIn, "a refill must cover one window");

// -DLZ4HIP_DEC_PROF (profiling builds only): shader-clock cycles per phase summed over all
// waves: [0] refill, [1] pass 1, [2] link, [3] compaction + full parse + scan + bounds,
// [4] literal and match copies into the stage, [5] in-window matches, [6] flush, [7] windows,
// [8] serial sequences and the rest.
#ifdef LZ4HIP_DEC_PROF
__device__ unsigned long long g_dec_prof[16];
#define DEC_T(i)                                                      \
    do {                                                              \
        const unsigned long long now_ = __builtin_readcyclecounter(); \
        prof[i] += now_ - last_t;                                     \
        last_t = now_;                                                \
    } while (0)
#define DEC_N(i) prof[i]++
#else
#define DEC_T(i)
#define DEC_N(i)
#endif

struct Hdr {  // one candidate sequence header
    bool ok;
    int lit, ml, off, lpos, end;  // lpos, end: absolute compressed positions
};

// Parses the header that would start at in[pos] (decode_other.go:24-32,59-107 for the
// plain case: lengths below 15+255, 16 literal bytes readable, another token follows).
__device__ __forceinline__ Hdr parse_header(const uint8_t *ib, const int pos, const int n) {
    Hdr h;
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
    h.off = (int)(w1 & 0xFFFFu);
    int ml = (int)(tok & 0xFu) + kMinMatch, e2 = 0;
    if ((tok & 0xFu) == 0xFu) {
        const uint32_t b2 = (w1 >> 16) & 0xFFu;
        ml += (int)b2;
        e2 = 1;
        ok = ok && b2 != 255u;
    }
    h.end = opos + 2 + e2;
    // another token must follow: the last sequences go through the serial path
    h.ok = ok && h.off != 0 && h.end < n;
    h.lit = lit;
    h.ml = ml;
    h.lpos = lpos;
    return h;
}

constexpr int kNoSeq = 4096;  // "sequence length" of a position where no plain sequence starts

// Pass 1 only needs to know where the sequence that would start at in[pos] ENDS: its
// compressed length from the token and at most one length byte.  ok = "could be a plain
// sequence" (lengths below 15+255 as far as this look goes, 18 bytes readable); the full
// parse of the tokens that turn out to be real re-checks everything.
__device__ __forceinline__ int light_seqlen(const uint8_t *ib, const int pos, const int n, bool &ok) {
    uint32_t w0;
    __builtin_memcpy(&w0, ib + pos, 4);  // inside the LDS buffer even past n (stale bytes, ok = false)
    const uint32_t tok = w0 & 0xFFu;
    const uint32_t b1 = (w0 >> 8) & 0xFFu;
    const bool xl = (tok >> 4) == 0xFu;
    ok = pos + 18 <= n && !(xl && b1 == 255u);
    // 1 token + literal length byte + literals + 2 offset + match length byte; a position
    // that cannot start a plain sequence gets kNoSeq, which throws the link out of its set
    return ok ? 3 + (int)(tok >> 4) + (xl ? 1 + (int)b1 : 0) + ((tok & 0xFu) == 0xFu ? 1 : 0) : kNoSeq;
}

// `stage`: kStage bytes of LDS owned by this wave, 16-byte aligned; `ibuf`: kIn bytes
// of LDS holding in[ibase, iend) (one coalesced refill per ~6 windows); `desc`: the
// window-relative positions of up to 64 tokens.  History before d: `prefix` bytes ending at `hist_end`.  hist_end
// == d means the history sits in memory right in front of d (single-block call, linked
// frames); otherwise it is a separate dictionary (decode_other.go:109-121), and a match
// that starts before d goes through the serial sequence decoder, which reads both parts.
__device__ __forceinline__ int decode_block(const uint8_t *__restrict__ in, const int n, uint8_t *d, const int dlen,
                             const int prefix, const uint8_t *hist_end, const int lane,
                             uint8_t *stage, uint8_t *ibuf, uint16_t *desc) {
    const bool split = hist_end != d;
    int si = 0, di = 0, ibase = 0, iend = 0;
#ifdef LZ4HIP_DEC_PROF
    unsigned long long prof[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long last_t = __builtin_readcyclecounter();
    struct Dump {
        unsigned long long *p;
        int lane;
        __device__ ~Dump() {
            if (lane == 0)
                for (int i = 0; i < 12; i++) atomicAdd(&g_dec_prof[i], p[i]);
        }
    } dump{prof, lane};
#endif
    for (;;) {
        DEC_T(8);
        si = (int)rfl((uint32_t)si);
        di = (int)rfl((uint32_t)di);
        if (si >= n) return di;
        ibase = (int)rfl((uint32_t)ibase);
        iend = (int)rfl((uint32_t)iend);
        if (si + kReach4 > iend && iend < n) {
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
        DEC_T(0);
        DEC_N(7);

        // ---------------- pass 1: where would a sequence starting at each byte end ----------------
        // set j, lane l <-> position si + 64 j + l
        int slen[kSets];
        uint64_t G[kSets];
#pragma unroll
        for (int j = 0; j < kSets; ++j) {
            bool ok;
            slen[j] = light_seqlen(ib, si + kWave * j + lane, n, ok);
            G[j] = ballot(ok);
        }
        if (!(G[0] & 1)) {  // the sequence at si itself is not a plain one
            const int st = d2_serial_sequence(in, n, d, dlen, prefix, hist_end, si, di, lane);
            if (st < 0) return -1;
            if (st > 0) return di;
            continue;
        }

        DEC_T(1);
        // ---------------- link the real tokens (wave-uniform) ----------------
        uint64_t T[kSets];
        {
            int t = 0;  // window-relative position of the next token
            bool go = true;
#pragma unroll
            for (int j = 0; j < kSets; ++j) {
                uint64_t Tj = 0;
                if (kSetTokens * (j + 1) > kWave && go) {  // only sets past the third
                    int have = 0;
                    for (int q = 0; q < j; ++q) have += __builtin_popcountll(T[q]);
                    if (have > kWave - kSetTokens) go = false;
                }
                if (go) {
                    const uint64_t Gj = G[j];
                    // compressed length of the sequence that would start at my position
                    const int seqlen = slen[j];
                    int b = t - kWave * j;  // >= 0: a set is entered from the one before
                    if (b < kWave) {
                        // do { T |= 1 << b; b += seqlen[lane b]; } while (b < 64);  T &= G;
                        // Four scalar instructions per token, hand-scheduled.  b is kept
                        // biased by -64 as an unsigned number: its low six bits are still the
                        // lane (s_bitset1 and v_readlane look at no others), and the add's
                        // carry-out is "b has left the set", so no compare is needed.  A
                        // position where no plain sequence starts has length kNoSeq: the walk
                        // marks it, leaves the set with b >= kNoSeq, and the mark is taken back.
                        int tmp;
                        uint32_t bb = (uint32_t)b - (uint32_t)kWave;
                        asm volatile(
                            "s_nop 1\n\t"  // the lane's seqlen was just written by the VALU
                            "1:\n\t"
                            "s_bitset1_b64 %[T], %[b]\n\t"
                            "v_readlane_b32 %[tmp], %[sl], %[b]\n\t"
                            "s_add_u32 %[b], %[b], %[tmp]\n\t"
                            "s_cbranch_scc0 1b"
                            : [T] "+s"(Tj), [b] "+s"(bb), [tmp] "=&s"(tmp)
                            : [sl] "v"(seqlen)
                            : "scc");
                        b = (int)(bb + (uint32_t)kWave);
                        Tj &= Gj;
                        if (b >= kNoSeq) go = false;  // stopped at a position that is no plain token
                    }
                    t = b + kWave * j;
                }
                T[j] = Tj;
            }
        }

        DEC_T(2);
        // ---------------- compact: token r of the window -> lane r ----------------
        int ntok = 0;
        lds_sync();  // the previous window's descriptor reads are done
#pragma unroll
        for (int j = 0; j < kSets; ++j) {
            const uint64_t Tj = T[j];
            if ((Tj >> lane) & 1) {
                const int r = ntok + (int)__builtin_amdgcn_mbcnt_hi(
                                         (uint32_t)(Tj >> 32),
                                         __builtin_amdgcn_mbcnt_lo((uint32_t)Tj, 0u));
                desc[r] = (uint16_t)(kWave * j + lane);
            }
            ntok += __builtin_popcountll(Tj);
        }
        lds_sync();
        // ---------------- full parse, token lanes only ----------------
        Hdr h;
        h.ok = false;
        h.lit = h.ml = h.off = 0;
        h.lpos = h.end = si;
        if (lane < ntok) h = parse_header(ib, si + (int)desc[lane], n);
        const int off = h.off;
        const int lit = lane < ntok ? h.lit : 0;
        const int ml = lane < ntok ? h.ml : 0;
        const int lpos = h.lpos;
        const int endrel = h.end - si;
        const int outlen = lit + ml;  // 0 in the lanes past ntok

        // exclusive scan of output lengths over the token lanes
        const int incl = wave_incl_scan(outlen);
        const int myout = incl - outlen;  // offset of my literals from di
        const int mstart = myout + lit;   // offset of my match from di
        // the reference's bounds: literals and match fit dst, offset inside history
        // (with a separate dictionary, also a match that starts inside it: serial path)
        // (and a token the full parse does not accept as plain: 255-run match length, offset
        // 0, too close to the end of the block -- the light look of pass 1 cannot see those)
        const bool bad = lane < ntok && (!h.ok || di + mstart + ml > dlen ||
                                         off > di + mstart + (split ? 0 : prefix));
        const uint64_t B = ballot(bad);
        if (B) ntok = __builtin_ctzll(B);  // stop before the first violation
        if (ntok == 0) {
            const int st = d2_serial_sequence(in, n, d, dlen, prefix, hist_end, si, di, lane);
            if (st < 0) return -1;
            if (st > 0) return di;
            continue;
        }
        const bool act = lane < ntok;
        const int adv_c = (int)rdl((uint32_t)endrel, ntok - 1);            // compressed bytes consumed
        const int adv_o = (int)rdl((uint32_t)(myout + outlen), ntok - 1);  // output bytes produced

        DEC_T(3);
        // ---------------- pass 2: copies ----------------
        if (adv_o <= kStage) {
            lds_sync();  // the previous window's flush has read the stage
            // literals (16 bytes readable after every token: ok => pos + 18 <= n)
            if (act && lit > 0) s2s_over(stage + myout, ib + lpos, lit);
#ifdef LZ4HIP_DEC_PROF
            lds_sync();
            DEC_T(9);  // literals
#endif
            // leading g bytes of every match come from output older than this window
            int g = 0;
            if (act) {
                g = off - mstart;
                g = g < 0 ? 0 : (g > ml ? ml : g);
                // the source lies before this window's output: dlen - (its offset in the block) may be read
                if (g > 0) g2s_exact(stage + mstart, d + di + mstart - off, g, dlen - (di + mstart - off));
            }
#ifdef LZ4HIP_DEC_PROF
            __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
            DEC_T(10);  // match bytes fetched
#endif
            lds_sync();
            DEC_T(4);
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
            DEC_T(5);
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
            // window output larger than the stage: direct copies
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
        DEC_T(6);
        si += adv_c;
        di += adv_o;
    }
}

}  // namespace

#ifndef LZ4HIP_DECODE_WAVES
#define LZ4HIP_DECODE_WAVES 8
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(LZ4HIP_DECODE_WAVES, 8))) void k_decode(
    const uint8_t *__restrict__ src, size_t src_len, const uint64_t *__restrict__ off,
    const uint32_t *__restrict__ sw, size_t nb, uint8_t *dst, size_t dst_stride, uint32_t dst_len,
    uint32_t prefix, const uint8_t *__restrict__ dict_end, int32_t *__restrict__ out_len) {
    __shared__ __attribute__((aligned(16))) uint8_t stage[kStage + 16];
    __shared__ __attribute__((aligned(16))) uint8_t ibuf[kIn + 16];
    __shared__ uint16_t desc[kWave];
    const int lane = lane_id();
    const size_t b = blockIdx.x;
    if (b >= nb) return;
    const uint32_t w = sw[b];
    const int n = (int)(w & 0x7FFFFFFFu);
    const uint64_t o = off[b];
    const uint8_t *in = src + o;
    uint8_t *d = dst + b * dst_stride;
    int r;
    if (o > src_len || (uint64_t)n > src_len - o) {
        r = -1;  // the block's span leaves the source buffer: nothing of it is read
    } else if (w >> 31) {  // stored raw: n := copy(dst, b.data), lz4stream/block.go:320-322
        r = (uint32_t)n > dst_len ? (int)dst_len : n;
        wave_copy16(d, in, r, lane);
    } else if (n == 0) {
        r = 0;  // block.go:45-47
    } else {
        // dict_end: a dictionary shared by every block of the batch; NULL: the `prefix`
        // bytes in front of d are the history
        r = decode_block(in, n, d, (int)dst_len, (int)prefix, dict_end ? dict_end : d, lane, stage,
                         ibuf, desc);
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
__global__ __launch_bounds__(64) void k_decode_linked(const uint8_t *__restrict__ src,
                                                       const uint64_t *__restrict__ off,
                                                       const uint32_t *__restrict__ sw, size_t nb,
                                                       uint8_t *dst, uint32_t block_cap,
                                                       size_t dst_cap,
                                                       int32_t *__restrict__ out_len,
                                                       uint64_t *__restrict__ out_off) {
    __shared__ __attribute__((aligned(16))) uint8_t stage[kStage + 16];
    __shared__ __attribute__((aligned(16))) uint8_t ibuf[kIn + 16];
    __shared__ uint16_t desc[kWave];
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
                r = decode_block(in, n, d, (int)cap, prefix, d, lane, stage, ibuf, desc);
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

#ifdef LZ4HIP_DEC_PROF
extern "C" int lz4hip_debug_dec_prof(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dec_prof), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -1;
}
#endif

hipError_t launch_decode_linked(const uint8_t *src, const uint64_t *off, const uint32_t *sw,
                                size_t nb, uint8_t *dst, uint32_t block_cap, size_t dst_cap,
                                int32_t *out_len, uint64_t *out_off, hipStream_t st) {
    if (nb == 0) return hipSuccess;
    hipLaunchKernelGGL(k_decode_linked, dim3(1), dim3(kWave), 0, st, src, off, sw, nb, dst,
                       block_cap, dst_cap, out_len, out_off);
    return hipGetLastError();
}

hipError_t launch_decode(const uint8_t *src, size_t src_len, const uint64_t *off, const uint32_t *sw,
                         size_t nb, uint8_t *dst, size_t dst_stride, uint32_t dst_len,
                         uint32_t prefix, const uint8_t *dict_end, int32_t *out_len, hipStream_t st) {
    if (nb == 0) return hipSuccess;
    hipLaunchKernelGGL(k_decode, dim3((unsigned)nb), dim3(kWave), 0, st, src, src_len, off, sw, nb,
                       dst, dst_stride, dst_len, prefix, dict_end, out_len);
    return hipGetLastError();
}

}  // namespace lz4hip
