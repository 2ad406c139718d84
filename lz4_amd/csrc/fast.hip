// fast.hip -- window-parallel Fast LZ4 compressor, one wavefront per block, hash tables in a
// global-memory workspace (k_fast_gtab).  Launches with few blocks take fast_team.hip instead.
//
// Same parse and bytes as (*Compressor).CompressBlock
// (internal/lz4block/block.go:105-295).  Per 64-position window:
//
//   Phase A: 64 lanes hash their position, gather the table, derive the candidate by get()'s
//   rule, verify 4 bytes and pre-extend (16 B forward, 8 B backward).
//
//   Phase B, the wave-uniform walk, only decides WHERE matches are.  Its common steps --
//   three clean misses, a match whose lane answer is final -- are hand-written scalar code
//   (fast_common.h walk_common_steps: ~28 SALU instructions per step; the compiler's version
//   of the same C++ was ~70 with a dozen branches, and the walk is the serial part of the
//   kernel); steps whose probes may depend on a put of this window, or whose match needs
//   cooperative extension, are resolved here in the reference's order.  Nothing is emitted
//   and no per-lane state is written during the walk: a match's anchor is the end of the
//   match before it.
//
//   Phase C, at window close, encodes ALL sequences of the window at once: every
//   match lane derives literal/match lengths (backward extension clamped by its
//   anchor, block.go:170-177), a wave scan of the encoded sizes gives each
//   sequence its place in dst, one comparison decides the dst-capacity outcome
//   (every check of block.go:196-245 guards a byte inside the sequence's extent,
//   so they all pass iff the extents fit), and the lanes write token, lengths,
//   literals and offset themselves.
#include "fast_common.h"

namespace lz4hip {

#ifndef LZ4HIP_SRCBUF
#define LZ4HIP_SRCBUF 1024
#endif
#ifndef LZ4HIP_OUTBUF
#define LZ4HIP_OUTBUF 512
#endif
constexpr int kSrcReach = 84;   // phase A reads s[wbase-8, wbase+84)
// LDS per wave besides the 8 KiB in-use bitmap: tag slots (u32 each), bytes of buffered source,
// bytes in which a window's sequences are assembled.  Two sizes are built: the standard one
// (11.8 KiB per wave: 13 waves per CU) and a compact one of exactly 10 KiB, which admits 16
// waves per CU.  The compact one is 4 % slower wave for wave (more refills, more false tag
// collisions) and only pays when it saves a round: launches of 3329..4096 blocks -- cfg5's
// 16 GiB share in 4 MiB blocks is 4096 -- run all blocks at once instead of 1.33 rounds of
// 3072 (526 -> 466 ms; profiles/r3_fast_gtab_wave_sweep.txt).  Same bytes either way.
constexpr size_t kFastStdResident = 3328;  // 13 waves per CU x 256 CUs
struct FastStd {
    static constexpr int kTag = kTagN, kSrc = LZ4HIP_SRCBUF, kOut = LZ4HIP_OUTBUF;
};
struct FastCompact {
    static constexpr int kTag = 256, kSrc = 512, kOut = 496;
};
constexpr int kInUseWords = (1 << 16) / 32;  // block.go:57: inUse [1<<16/32]uint32
#ifndef LZ4HIP_FAST_GATE
#define LZ4HIP_FAST_GATE 0
#endif
#ifndef LZ4HIP_FAST_CANDLDS
#define LZ4HIP_FAST_CANDLDS 0
#endif
#ifndef LZ4HIP_FAST_HIST
#define LZ4HIP_FAST_HIST 8
#endif
constexpr int kSrcHist = LZ4HIP_FAST_HIST;      // source bytes behind the window kept at a refill (>= 8)
static_assert(kSrcHist >= 8 && kSrcHist + 2 * kSrcReach <= FastStd::kSrc && kSrcHist + 2 * kSrcReach <= FastCompact::kSrc,
              "a refill must cover the window");
#ifndef LZ4HIP_FAST_BITMAP
#define LZ4HIP_FAST_BITMAP 1
#endif
// 1: the reference's in-use bitmap (8 KiB of LDS per wave; an unused slot costs no memory
// request).  0: the global table is zeroed at every block start instead and every valid lane
// gathers -- 3.5 KiB of LDS per wave, so twice as many waves fit a CU.
constexpr bool kBitmap = LZ4HIP_FAST_BITMAP;

constexpr bool kGateExt = LZ4HIP_FAST_GATE;     // extension loads only behind a verified candidate
constexpr bool kCandLds = LZ4HIP_FAST_CANDLDS;  // candidates inside the LDS source buffer are read there

// -DLZ4HIP_FAST_STATS (profiling builds only, tools/prof): lanes that issue a memory access,
// by class, summed per block and added to uint64 counters behind the block counter in the
// workspace tail: [1] table gathers, [2] 4-byte candidate verifies, [3] r0, [4] r1, [5] rb
// extension loads, [6] table puts, [7] windows, [8] refill lines (64 B), [9] input bytes.
#ifdef LZ4HIP_FAST_STATS
#define FAST_STAT(i, v) stat[i] += (unsigned long long)(v)
// shader-clock cycles per phase, summed over the waves: [10] refill, [11] table gather (to the
// first use of its result), [12] candidate verify + extension, [13] walk, [14] encode, [15] puts
#define FAST_TIME(i)                                                  \
    do {                                                              \
        const unsigned long long now_ = __builtin_readcyclecounter(); \
        stat[i] += now_ - last_t;                                     \
        last_t = now_;                                                \
    } while (0)
#else
#define FAST_STAT(i, v)
#define FAST_TIME(i)
#endif

template <typename Cfg, typename TablePtr>
__device__ __forceinline__ int fast_block(const uint8_t *__restrict__ s, const int n,
                                           uint8_t *__restrict__ d, const int dlen,
                                           TablePtr table, uint32_t *tag, uint8_t *sbuf,
                                           uint8_t *obuf, uint32_t *inuse, const int lane,
                                           unsigned long long *stats = nullptr) {
#ifdef LZ4HIP_FAST_STATS
    unsigned long long stat[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    stat[9] = (unsigned long long)n;
    unsigned long long last_t = __builtin_readcyclecounter();
#endif
    constexpr int kTagN = Cfg::kTag, kSrcBuf = Cfg::kSrc, kOutBuf = Cfg::kOut;
    const int bound = (int)compress_bound32((uint32_t)n);
    const bool not_compressible = dlen < bound;  // block.go:110
    {  // reset(): block.go:94,107
        if (!kBitmap) {
            typedef uint4 __attribute__((may_alias)) uint4_a;
            uint4_a *t4 = (uint4_a *)(table);
            for (int k = lane; k < (1 << 16) * 2 / 16; k += kWave) t4[k] = make_uint4(0, 0, 0, 0);
            __threadfence();
        } else {
            // the reference's own scheme: clear the in-use bitmap (8 KiB, here in LDS), not the
            // table.  A get() of an unused slot is answered from the bitmap -- table value 0,
            // block.go:74-79 -- without touching memory, and the table is never zeroed.
            for (int k = lane; k < kInUseWords; k += kWave) inuse[k] = 0u;
        }
        for (int k = lane; k < kTagN; k += kWave) tag[k] = 0xFFFFFFFFu;
    }
    __syncthreads();

    int si = 0, di = 0, anchor = 0;
    const int sn = n - kMfLimit;
    bool failed = false, finished = false, pending = false;
    // The window's own source bytes -- s[wbase-8, wbase+84) -- are read through an
    // LDS buffer holding s[sb, se), refilled with one coalesced load every ~40
    // windows: the sequential part of the input costs no global round trip per window.
    int sb = 0, se = 0;

    while (si < sn && !failed && !finished) {
        si = (int)rfl((uint32_t)si);
        di = (int)rfl((uint32_t)di);
        anchor = (int)rfl((uint32_t)anchor);
        sb = (int)rfl((uint32_t)sb);
        se = (int)rfl((uint32_t)se);
        // ================= Phase A: lane l <-> position wbase + l =================
        const int wbase = si - 2;  // lane 0 is the position of a pending put(si-2)
        if (wbase + kSrcReach > se && se < n) {
            sb = wbase - kSrcHist < 0 ? 0 : (wbase - kSrcHist) & ~15;
            const int cnt = n - sb < kSrcBuf ? n - sb : kSrcBuf;
            __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            for (int k = lane * 16; k < cnt; k += kWave * 16) {
                if (k + 16 <= cnt) {
                    const uint64_t a = ld64(s + sb + k), b = ld64(s + sb + k + 8);
                    *(uint64_t *)(sbuf + k) = a;
                    *(uint64_t *)(sbuf + k + 8) = b;
                } else {
                    for (int j = k; j < cnt; j++) sbuf[j] = s[sb + j];
                }
            }
            se = sb + cnt;
            FAST_STAT(8, (cnt + 63) / 64);
            __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        FAST_TIME(10);
        const uint8_t *sl = sbuf - sb;  // sl[q] == s[q] for sb <= q < se
        const int p = wbase + lane;
        const bool valid = p >= 0 && p <= sn + 1;
        uint64_t x = 0;
        if (valid) __builtin_memcpy(&x, sl + p, 8);
        const uint32_t h = hash6(x);
        uint32_t tv = 0;
        {
            const bool used = valid && (!kBitmap || ((inuse[h >> 5] >> (h & 31)) & 1u));
            if (used) tv = tab_load(table, h);
        }
        const uint32_t slot = h & (kTagN - 1);
        if (valid) atomicMin(&tag[slot], (uint32_t)lane);
        __asm__ volatile("" ::: "memory");
        uint32_t tmin = 0xFFFFFFFFu;
        if (valid) tmin = tag[slot];
        __asm__ volatile("" ::: "memory");
        if (valid) tag[slot] = 0xFFFFFFFFu;
        const bool cflag = valid && tmin < (uint32_t)lane;  // an earlier lane shares my slot

        int ref = (int)tv + (p & ~kWinMask);
        if (ref >= p) ref -= kWinSize;
        int offv = p - ref;  // > 0 by construction
        const bool offok = valid && offv < kWinSize && ref >= 0;
#ifdef LZ4HIP_FAST_STATS
        if (ballot(offok) != 0x5555555555555555ull) FAST_TIME(11);  // (the ballot uses tv: the gather has landed)
#endif
        const bool f0ok = offok && p + 12 <= sn;  // stride 0 at p+4 runs iff p+4+8 <= sn
        const bool f1ok = offok && p + 20 <= sn;
        const bool bok = offok && ref >= 8;
        uint32_t cw = 0;
        uint64_t r0 = 0, q0 = 0, r1 = 0, q1 = 0, rb = 0, qb = 0;
        // Candidate bytes: from the LDS source buffer when the candidate is recent enough to
        // be in it (no memory request at all), else from global memory.
        // (LDS reads are indexed from sbuf itself: `sl` is only valid while the arithmetic stays
        // in the 32-bit LDS address space, and a pointer select would widen it to a flat one.)
        auto cand64 = [&](int q) -> uint64_t {
            uint64_t v = 0;
            const bool in = kCandLds && q >= sb && q + 8 <= se;
            if (in) __builtin_memcpy(&v, sbuf + (q - sb), 8);
            if (!in) v = ld64(s + q);
            return v;
        };
        if (offok) {
            const bool in = kCandLds && ref >= sb && ref + 4 <= se;
            if (in) __builtin_memcpy(&cw, sbuf + (ref - sb), 4);
            if (!in) cw = ld32(s + ref);
        }
        // The extension bytes lie up to 8 bytes before and 20 behind the candidate: a third
        // of them touch a second 64-byte line.  With kGateExt only the lanes whose 4 bytes
        // verified fetch them (one more round trip for those lanes).
        const bool want_ext = !kGateExt || (offok && cw == (uint32_t)x);
        if (f0ok && want_ext) {
            r0 = cand64(ref + 4);
            __builtin_memcpy(&q0, sl + p + 4, 8);
        }
        if (f1ok && want_ext) {
            r1 = cand64(ref + 12);
            __builtin_memcpy(&q1, sl + p + 12, 8);
        }
        if (bok && want_ext) {
            rb = cand64(ref - 8);
            __builtin_memcpy(&qb, sl + p - 8, 8);
        }
        FAST_STAT(7, 1);
        FAST_STAT(1, __builtin_popcountll(ballot(kBitmap ? (valid && ((inuse[h >> 5] >> (h & 31)) & 1u)) : valid)));
        FAST_STAT(2, __builtin_popcountll(ballot(offok)));
        FAST_STAT(3, __builtin_popcountll(ballot(f0ok)));
        FAST_STAT(4, __builtin_popcountll(ballot(f1ok)));
        FAST_STAT(5, __builtin_popcountll(ballot(bok)));
        const bool m4 = offok && cw == (uint32_t)x;
        int mendv = p + kMinMatch;  // end of the match that starts at p (absolute)
        bool fmore = false;
        if (m4 && f0ok) {
            uint64_t xx = r0 ^ q0;
            if (xx) {
                mendv += __builtin_ctzll(xx) >> 3;
            } else {
                mendv += 8;
                if (f1ok) {
                    xx = r1 ^ q1;
                    if (xx) {
                        mendv += __builtin_ctzll(xx) >> 3;
                    } else {
                        mendv += 8;
                        fmore = true;  // still equal after 16 bytes: continue cooperatively
                    }
                }
            }
        }
        int backv = 0;  // backward-equal bytes (upper bound; clamped by the anchor later)
        bool bmore = false;
        if (m4) {
            if (bok) {
                const uint64_t xx = rb ^ qb;
                if (xx) {
                    backv = __builtin_clzll(xx) >> 3;
                } else {
                    backv = 8;
                    bmore = true;
                }
            } else {
                bmore = true;  // too close to the block start for the 8-byte loads
            }
        }
        const uint64_t M = ballot(m4), CM = ballot(cflag);
        const uint64_t RARE = CM | ballot(fmore || bmore);  // probes that need more than a lookup
        const uint64_t ATT = M | CM;                        // lanes that need a look during the walk
        const uint16_t pos16 = (uint16_t)p;
        const uint32_t flv = (uint32_t)fmore | ((uint32_t)bmore << 1);
        const int anchor_in = anchor;  // a match's anchor is the end of the match before it, else this
        uint64_t FOUND = 0;    // match lanes of this window
        uint64_t P = pending ? 1ull : 0ull;  // lanes put() since the snapshot; lane 0 = put(si-2)
        pending = false;

#ifdef LZ4HIP_FAST_STATS
        if (ballot(m4) != 0x5555555555555557ull) FAST_TIME(12);  // (uses the candidate words)
#endif
        // ================= Phase B: where are the matches =================
        // The common steps (three clean misses; a match whose lane answer is final) run as
        // hand-written scalar code (fast_common.h); it hands the steps that need more back.
        for (;;) {
            si = (int)rfl((uint32_t)si);
            anchor = (int)rfl((uint32_t)anchor);
            const int st = walk_common_steps(si, anchor, P, FOUND, wbase, sn, ATT, RARE, mendv);
            if (st == 0) break;
            if (st == 2) {  // block.go:250-252
                finished = true;
                break;
            }
            // ---- rare: a probe may depend on a put of this window, or the match
            // needs cooperative extension.  Resolve in the reference's order. ----
            const int a = si - wbase;
            int j = -1, dep_l = -1;
#pragma unroll
            for (int t = 0; t < 3 && j < 0; ++t) {
                const int b = a + t;
                // probes 0,1 see the puts before this step; probe 2 also put(si), put(si+1)
                const uint64_t vis = t == 2 ? (P | (3ull << a)) : P;
                bool hit = (M >> b) & 1;
                int dl = -1;
                if ((CM >> b) & 1) {
                    const uint64_t E = ballot(h == rdlane(h, b)) & vis;
                    if (E) {
                        dl = 63 - __builtin_clzll(E);  // the latest put with my hash
                        hit = rdlane((uint32_t)x, b) == rdlane((uint32_t)x, dl);
                    }
                }
                if (hit) {
                    j = t;
                    dep_l = dl;
                }
            }
            if (j < 0) {
                P |= 7ull << a;
                si += 4 + ((si + 2 - anchor) >> kAdaptSkipLog);
                continue;
            }
            const int f = a + j;  // match lane
            P |= (j == 2 ? 7ull : 3ull) << a;
            const int found = si + j;
            const int l0 = found - anchor;
            int offset, back, mend;
            if (dep_l >= 0) {
                offset = f - dep_l;
                const int mref = found - offset;
                const int maxb = l0 < mref ? l0 : mref;
                back = ext_backward(s, found, mref, 0, maxb, lane);
                mend = ext_forward(s, found + kMinMatch, offset, sn, lane);
            } else {
                offset = (int)rdlane((uint32_t)offv, f);
                const int mref = found - offset;
                const int maxb = l0 < mref ? l0 : mref;
                const int bw = (int)rdlane((uint32_t)backv, f);
                back = bw < maxb ? bw : maxb;
                mend = (int)rdlane((uint32_t)mendv, f);
                const uint32_t fl = rdlane(flv, f);
                if ((fl & 2u) && maxb > bw) back = ext_backward(s, found, mref, bw, maxb, lane);
                if (fl & 1u) mend = ext_forward(s, mend, offset, sn, lane);
            }
            offv = wrlane(offv, offset, f);
            backv = wrlane(backv, back, f);
            mendv = wrlane(mendv, mend, f);
            si = mend;
            FOUND |= 1ull << f;
            anchor = si;
            if (si >= sn) {  // block.go:250-252
                finished = true;
                break;
            }
            const int q = si - 2 - wbase;  // put(hash(si-2), si-2): block.go:254-255
            if (q <= 63) P |= 1ull << q;
        }
        // a match that ended the window leaves its put(si-2) to the next one (its lane 0)
        pending = !finished && anchor == si && si - 2 - wbase > 63;

        FAST_TIME(13);
        // ================= Phase C: encode every sequence of the window =================
        if (FOUND) {
            const bool mine = (FOUND >> lane) & 1;
            // the anchor of a match: the end of the match before it in the window (block.go:249)
            const uint64_t below = FOUND & ((1ull << lane) - 1);
            const int prev_end = __shfl(mendv, below ? 63 - __builtin_clzll(below) : 0);
            const int anch = below ? prev_end : anchor_in;
            // block.go:166-182 with that anchor
            const int l0 = p - anch;
            const int mref = p - offv;
            int back = backv;
            if (back > l0) back = l0;
            if (back > mref) back = mref;
            const int m_start = p - back;
            const int l_len = m_start - anch;
            const int m_len = mendv - m_start - kMinMatch;
            const int lt = len_tail_size(l_len), mt = len_tail_size(m_len);
            const int esz = mine ? 1 + lt + l_len + 2 + mt : 0;
            const int incl = wave_incl_scan(esz);
            const int total = (int)rdlane((uint32_t)incl, 63);
            if (di + total > dlen) {
                // some byte of some sequence would land at or beyond len(dst):
                // ErrInvalidSourceShortBuffer (block.go:196-245)
                failed = true;
            } else {
                auto emit = [&](uint8_t *o) {
                    *o++ = (uint8_t)((m_len < 0xF ? m_len : 0xF) | ((l_len < 0xF ? l_len : 0xF) << 4));
                    if (l_len >= 0xF) o += lane_len_tail(o, l_len - 0xF);
                    // literals are the bytes just before the match: usually still in sbuf
                    if (anch >= sb)
                        lane_copy_exact(o, sl + anch, l_len);
                    else
                        lane_copy_exact(o, s + anch, l_len);
                    o += l_len;
                    const int ofs = offv;
                    o[0] = (uint8_t)ofs;
                    o[1] = (uint8_t)(ofs >> 8);
                    if (m_len >= 0xF) lane_len_tail(o + 2, m_len - 0xF);
                };
                if (kOutBuf > 0 && total <= kOutBuf) {
                    // The sequences of a window are a few dozen bytes each and contiguous in
                    // dst: assemble them in LDS and write them out 16 bytes per lane (two or
                    // three store requests per window instead of a dozen byte-granular ones).
                    __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the last flush has read obuf
                    if (mine) emit(obuf + (incl - esz));
                    __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    uint8_t *to = d + di;
                    for (int k = lane * 16; k < total; k += kWave * 16) {
                        const uint64_t a = *(const uint64_t *)(obuf + k), b = *(const uint64_t *)(obuf + k + 8);
                        const int rem = total - k;
                        if (rem >= 16) {
                            __builtin_memcpy(to + k, &a, 8);
                            __builtin_memcpy(to + k + 8, &b, 8);
                        } else if (rem >= 8) {
                            __builtin_memcpy(to + k, &a, 8);
                            store_tail8(to + k + 8, b, rem - 8);
                        } else {
                            store_tail8(to + k, a, rem);
                        }
                    }
                } else if (mine) {
                    emit(d + di + (incl - esz));
                }
                di += total;
            }
        }

        FAST_TIME(14);
        // ---- close the window: write this window's puts to the table ----
        if (!failed && !finished) {
            FAST_STAT(6, __builtin_popcountll(P));
            if ((P & ~CM) >> lane & 1) table[h] = pos16;  // distinct slots: one store
            uint64_t Pd = P & CM;  // may share a slot with an earlier put: position order
            while (Pd) {
                __asm__ volatile("" ::: "memory");
                const int l = __builtin_ctzll(Pd);
                Pd &= Pd - 1;
                if (lane == l) table[h] = pos16;
            }
            __asm__ volatile("" ::: "memory");
            if (kBitmap && ((P >> lane) & 1)) atomicOr(&inuse[h >> 5], 1u << (h & 31));
            __asm__ volatile("" ::: "memory");
        }
        FAST_TIME(15);
    }

    // ---------------- last literals: block.go:258-294 ----------------
    di = (int)rfl((uint32_t)di);
    anchor = (int)rfl((uint32_t)anchor);
    int ret;
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
#ifdef LZ4HIP_FAST_STATS
    if (stats && lane == 0)
        for (int i = 1; i < 16; i++) atomicAdd(&stats[i], stat[i]);
#endif
    return ret;
}

// Table in a global-memory slot per resident wave; many waves per CU.
template <typename Cfg>
__global__ __launch_bounds__(64) void k_fast_gtab(const uint8_t *__restrict__ src, size_t n_total,
                                                   uint32_t block_size, size_t nb, int dst_mode,
                                                   uint32_t dst_len_x, uint8_t *__restrict__ dst,
                                                   size_t dst_stride,
                                                   int32_t *__restrict__ result,
                                                   uint16_t *__restrict__ ws,
                                                   uint32_t *__restrict__ next_block) {
    __shared__ uint32_t tag[Cfg::kTag];
    __shared__ __attribute__((aligned(16))) uint8_t sbuf[Cfg::kSrc];
    __shared__ __attribute__((aligned(16))) uint8_t obuf[Cfg::kOut + 16];
    __shared__ uint32_t inuse[kBitmap ? kInUseWords : 1];
    const int lane = lane_id();
    uint16_t *table = ws + (size_t)blockIdx.x * (1 << 16);
    // Blocks cost very different amounts of time (incompressible ones a fraction of
    // text), so a wave takes its next block from a shared counter when it finishes one
    // (goroutines pulling from the Writer's channel do the same, writer.go:130-142);
    // without a counter the blocks are dealt round-robin.
    auto take = [&]() -> size_t {
        uint32_t t = 0;
        if (lane == 0) t = atomicAdd(next_block, 1u);
        return (size_t)rfl(t);
    };
    // with the counter even the first block is taken from it: a workgroup that only gets a
    // CU once others have finished must not sit on a block dealt to it in advance
    size_t b = next_block ? take() : (size_t)blockIdx.x;
    while (b < nb) {
        const size_t start = b * (size_t)block_size;
        const size_t left = n_total > start ? n_total - start : 0;
        const int n = (int)(left < block_size ? left : block_size);
        const int r = fast_block<Cfg>(src + start, n, dst + b * dst_stride,
                                       block_dlen2(dst_mode, n, dst_len_x), table, tag, sbuf, obuf,
                                       inuse, lane,
                                       reinterpret_cast<unsigned long long *>(next_block));
        if (lane == 0) result[b] = r;
        __syncthreads();
        if (next_block)
            b = take();
        else
            b += gridDim.x;
    }
}

hipError_t launch_fast(const uint8_t *src, size_t n, uint32_t bs, size_t nb, int dst_mode,
                        uint32_t dst_len_x, uint8_t *dst, size_t dst_stride, int32_t *result,
                        uint16_t *ws, size_t slots, uint32_t *next_block, hipStream_t st) {
    if (nb == 0) return hipSuccess;
    if (ws && slots) {
        size_t grid = nb < slots ? nb : slots;
        // More than 13 waves per CU only pay when they save a round (every block gets a wave at
        // 16 per CU); with rounds to go anyway the standard layout at its 12-per-CU plateau wins.
        if (grid > kFastStdResident && (nb > grid || grid > 4096)) grid = 3072;
        if (grid == nb) next_block = nullptr;  // every block has a wave of its own
        if (next_block) {
            const hipError_t e = hipMemsetAsync(next_block, 0, sizeof(uint32_t), st);
            if (e != hipSuccess) return e;
        }
        // more waves than 13 per CU can be resident only with the compact LDS layout
        if (grid > kFastStdResident)
            hipLaunchKernelGGL(k_fast_gtab<FastCompact>, dim3((unsigned)grid), dim3(kWave), 0, st, src, n, bs,
                               nb, dst_mode, dst_len_x, dst, dst_stride, result, ws, next_block);
        else
            hipLaunchKernelGGL(k_fast_gtab<FastStd>, dim3((unsigned)grid), dim3(kWave), 0, st, src, n, bs, nb,
                               dst_mode, dst_len_x, dst, dst_stride, result, ws, next_block);
    } else {  // few blocks: a team of waves per block, table in LDS (fast_team.hip)
        return launch_fast_team(src, n, bs, nb, dst_mode, dst_len_x, dst, dst_stride, result, st);
    }
    return hipGetLastError();
}

}  // namespace lz4hip
