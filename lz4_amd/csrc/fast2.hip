// fast2.hip -- window-parallel Fast (greedy) LZ4 block compressor for gfx950.
//
// Same parse as (*Compressor).CompressBlock (internal/lz4block/block.go:105-295),
// so the same bytes, but organised for a 64-lane wavefront:
//
//   Phase A (64 lanes, one per source position of a 64-byte window): hash6
//   (block.go:35-38), table lookup, candidate address by the rule of get()
//   (block.go:74-86), 4-byte verify, and the first 16 bytes of forward / 8 bytes
//   of backward match extension (block.go:170-193) -- all speculative, against
//   the table as it stood when the window was opened.
//
//   Phase B (wave-uniform walk): replays the reference's step sequence over the
//   window using the per-lane answers.  A lane's speculative answer is exact
//   unless a position put() since the window opened has the same hash; a small
//   LDS tag table flags the lanes that may be affected and an exact ballot
//   resolves them (the newer candidate is then a window lane itself).  Puts,
//   the get/put order of block.go:134-155, the adaptive skip (:159) and the
//   emission (:195-248) follow the reference step by step.
//
// The memory round trips of the serial walk (table load -> candidate load ->
// extension loads, per step) become three gathers per 64 positions.
#include "fast_common.h"

namespace lz4hip {

template <bool kGlobal, typename TablePtr>
__device__ __forceinline__ int fast2_block(const uint8_t *__restrict__ s, const int n,
                                           uint8_t *__restrict__ d, const int dlen,
                                           TablePtr table, uint32_t *tag, const int lane) {
    const int bound = (int)compress_bound32((uint32_t)n);
    const bool not_compressible = dlen < bound;  // block.go:110
    {  // reset(): block.go:94,107 -- an all-zero table is the cleared-bitmap state
        typedef uint4 __attribute__((may_alias)) uint4_a;
        uint4_a *t4 = (uint4_a *)(table);
        for (int k = lane; k < (1 << 16) * 2 / 16; k += kWave) t4[k] = make_uint4(0, 0, 0, 0);
        for (int k = lane; k < kTagN; k += kWave) tag[k] = 0xFFFFFFFFu;
    }
    __syncthreads();

    int si = 0, di = 0, anchor = 0;
    const int sn = n - kMfLimit;
    bool failed = false, finished = false, pending = false;

    while (si < sn && !failed && !finished) {
        // Keep the walk state in SGPRs: every value below is wave-uniform by
        // construction; readfirstlane makes that visible to the compiler.
        si = (int)rfl((uint32_t)si);
        di = (int)rfl((uint32_t)di);
        anchor = (int)rfl((uint32_t)anchor);
        // ================= Phase A: lane l <-> position wbase + l =================
        const int wbase = si - 2;  // lane 0 is the position of a pending put(si-2)
        const int p = wbase + lane;
        const bool valid = p >= 0 && p <= sn + 1;
        uint64_t x = 0;
        if (valid) x = ld64(s + p);
#ifdef LZ4HIP_EXP_SMALLTAB
        const uint32_t h = hash6(x) & 0xFFFu;  // timing experiment only: wrong bytes, L2-resident table
#else
        const uint32_t h = hash6(x);
#endif
        uint32_t tv = 0;
        if (valid) tv = tab_load<kGlobal>(table, h);
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
        const int off = p - ref;  // > 0 by construction
        const bool offok = valid && off < kWinSize && ref >= 0;
        const bool f0ok = offok && p + 12 <= sn;  // stride 0 at p+4 runs iff p+4+8 <= sn
        const bool f1ok = offok && p + 20 <= sn;
        const bool bok = offok && ref >= 8;
        uint32_t cw = 0;
        uint64_t r0 = 0, q0 = 0, r1 = 0, q1 = 0, rb = 0, qb = 0;
        if (offok) cw = ld32(s + ref);
        if (f0ok) {
            r0 = ld64(s + ref + 4);
            q0 = ld64(s + p + 4);
        }
        if (f1ok) {
            r1 = ld64(s + ref + 12);
            q1 = ld64(s + p + 12);
        }
        if (bok) {
            rb = ld64(s + ref - 8);
            qb = ld64(s + p - 8);
        }
        const bool m4 = offok && cw == (uint32_t)x;
        int fend = p + kMinMatch;
        bool fmore = false;
        if (m4 && f0ok) {
            uint64_t xx = r0 ^ q0;
            if (xx) {
                fend += __builtin_ctzll(xx) >> 3;
            } else {
                fend += 8;
                if (f1ok) {
                    xx = r1 ^ q1;
                    if (xx) {
                        fend += __builtin_ctzll(xx) >> 3;
                    } else {
                        fend += 8;
                        fmore = true;  // still equal after 16 bytes: continue cooperatively
                    }
                }
            }
        }
        int bwd = 0;
        bool bmore = false;
        if (m4) {
            if (bok) {
                const uint64_t xx = rb ^ qb;
                if (xx) {
                    bwd = __builtin_clzll(xx) >> 3;
                } else {
                    bwd = 8;
                    bmore = true;
                }
            } else {
                bmore = true;  // too close to the block start for the 8-byte loads
            }
        }
        // Per-lane answers packed for one readlane at match time:
        // offset[15:0] | bwd[19:16] | (fend - p)[24:20] | fmore[25] | bmore[26]
        const uint32_t pk = (uint32_t)off | ((uint32_t)bwd << 16) | ((uint32_t)(fend - p) << 20) |
                            ((uint32_t)fmore << 25) | ((uint32_t)bmore << 26);
        const uint64_t M = ballot(m4), CM = ballot(cflag);
        const uint64_t ATT = M | CM;  // lanes that need a look during the walk
        const uint16_t pos16 = (uint16_t)p;
        // Window lanes put() since the snapshot.  The table itself is written once,
        // when the window closes: until then every dependency on these puts is
        // resolved in registers (cflag + exact ballot below).
        uint64_t P = pending ? 1ull : 0ull;  // lane 0 = put(si-2) of the previous match (:254-255)
        pending = false;

        // ================= Phase B: the reference's steps over this window =================
        for (;;) {
            si = (int)rfl((uint32_t)si);
            di = (int)rfl((uint32_t)di);
            anchor = (int)rfl((uint32_t)anchor);
            P = ((uint64_t)rfl((uint32_t)(P >> 32)) << 32) | rfl((uint32_t)P);
            const int a = si - wbase;
            if (a > 61 || si >= sn) break;
            const uint32_t att3 = (uint32_t)(ATT >> a) & 7u;
            if (att3 == 0) {
                // No candidate verifies at si, si+1, si+2 and no put can interfere:
                // the three puts happen (:136-137,155) and the step skips (:159).
                const int d0 = si + 2 - anchor;
                if (d0 <= 64 && si + 64 < sn) {
                    // The skip stays 4 bytes for every step that starts in this window:
                    // take all the clean steps up to the next lane that needs a look.
                    const uint64_t pat = 0x7777777777777777ull << a;  // probed lanes a+4k+{0,1,2}
                    const uint64_t hot = ATT & pat;
                    const int kmax = ((61 - a) >> 2) + 1;  // steps that start at a lane <= 61
                    // clean steps before the one holding the next hot lane (>= 1: att3 == 0)
                    int k = hot ? (__builtin_ctzll(hot) - a) >> 2 : kmax;
                    if (k > kmax) k = kmax;
                    const int e = a + 4 * k;  // first lane not covered (<= 65)
                    P |= e >= 64 ? pat : (pat & ((1ull << e) - 1));
                    si += 4 * k;
                } else {
                    P |= 7ull << a;
                    si += 4 + (d0 >> kAdaptSkipLog);
                }
                continue;
            }
            int j, offset = 0;
            bool dep = false;
            const uint32_t c3 = (uint32_t)(CM >> a) & 7u;
            if (c3 == 0) {
                // Snapshot answers are exact for all three probes.
                j = __builtin_ctz(att3);
                P |= (j == 2 ? 7ull : 3ull) << a;
            } else {
                // Some probe lane shares its tag slot with an earlier lane: resolve the
                // probes in the reference's order against the puts made so far.
                bool hit[3], dp[3];
                int of[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const int b = a + t;
                    // probes 0 and 1 see the puts before this step; probe 2 also sees
                    // put(si), put(si+1)  (block.go:134-137,145)
                    const uint64_t vis = t == 2 ? (P | (3ull << a)) : P;
                    hit[t] = (M >> b) & 1;
                    dp[t] = false;
                    of[t] = 0;
                    if ((CM >> b) & 1) {
                        const uint64_t E = ballot(h == rdlane(h, b)) & vis;
                        if (E) {
                            const int l = 63 - __builtin_clzll(E);  // the latest such put
                            dp[t] = true;
                            of[t] = b - l;
                            hit[t] = rdlane((uint32_t)x, b) == rdlane((uint32_t)x, l);
                        }
                    }
                }
                if (hit[0]) {
                    j = 0;
                } else if (hit[1]) {
                    j = 1;
                } else if (hit[2]) {
                    j = 2;
                } else {
                    P |= 7ull << a;
                    si += 4 + ((si + 2 - anchor) >> kAdaptSkipLog);
                    continue;
                }
                dep = dp[j];
                offset = of[j];
                P |= (j == 2 ? 7ull : 3ull) << a;
            }

            // ---------------- match found at lane f ----------------
            const int f = a + j;
            const int found = si + j;
            const uint32_t pkf = rdlane(pk, f);
            if (!dep) offset = (int)(pkf & 0xFFFFu);
            const int mref = found - offset;
            const int l0 = found - anchor;
            const int maxb = l0 < mref ? l0 : mref;  // lLen > 0 && tOff >= 0
            int back;
            if (dep) {
                back = ext_backward(s, found, mref, 0, maxb, lane);
                si = ext_forward(s, found + kMinMatch, offset, sn, lane);
            } else {
                const int bw = (int)((pkf >> 16) & 0xFu);
                back = bw < maxb ? bw : maxb;
                if (((pkf >> 26) & 1u) && maxb > bw) back = ext_backward(s, found, mref, bw, maxb, lane);
                si = found + (int)((pkf >> 20) & 0x1Fu);
                if ((pkf >> 25) & 1u) si = ext_forward(s, si, offset, sn, lane);
            }
            const int m_start = found - back;
            const int l_len = m_start - anchor;
            int m_len = si - (m_start + kMinMatch);

            // ---------------- emit: block.go:195-248 ----------------
            const int la = anchor - wbase;  // first literal lane (may be negative)
            if (l_len < 0xF && m_len < 0xF && la >= 0 && di + l_len + 3 <= dlen) {
                // common case, one token byte and no length runs; every check of
                // :196-245 holds because di + 1 + l_len + 2 <= dlen.
                if (lane == 0) d[di] = (uint8_t)(m_len | (l_len << 4));
                if ((unsigned)(lane - la) < (unsigned)l_len) d[di + 1 + (lane - la)] = (uint8_t)x;
                di += 1 + l_len + 2;
                if (lane == 0) {
                    d[di - 2] = (uint8_t)offset;
                    d[di - 1] = (uint8_t)(offset >> 8);
                }
            } else {
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
                {  // literals [anchor, m_start): from the window registers where possible
                    int from_mem = -la;  // literals that precede the window
                    if (from_mem > 0) {
                        if (from_mem > l_len) from_mem = l_len;
                        wave_copy(d + di, s + anchor, from_mem, lane);
                    } else {
                        from_mem = 0;
                    }
                    const int lb = la + from_mem;  // first literal lane held in registers
                    const int lm = m_start - wbase;
                    if (lane >= lb && lane < lm) d[di + from_mem + (lane - lb)] = (uint8_t)x;
                }
                di += l_len + 2;
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
            anchor = si;
            if (si >= sn) {  // block.go:250-252
                finished = true;
                break;
            }
            // put(hash(si-2), si-2): block.go:254-255
            const int q = si - 2 - wbase;
            if (q <= 63)
                P |= 1ull << q;
            else
                pending = true;  // lane 0 of the next window
        }

        // ---- close the window: write this window's puts to the table ----
        if (!failed && !finished) {
            // Lanes whose tag slot no earlier lane shares hit distinct slots: one store.
            if ((P & ~CM) >> lane & 1) table[h] = pos16;
            // The others may share a slot with an earlier put: store them after, one
            // at a time in position order, so the latest put owns the slot.
            uint64_t Pd = P & CM;
            while (Pd) {
                __asm__ volatile("" ::: "memory");
                const int l = __builtin_ctzll(Pd);
                Pd &= Pd - 1;
                if (lane == l) table[h] = pos16;
            }
            __asm__ volatile("" ::: "memory");
        }
    }

    // ---------------- last literals: block.go:258-294 ----------------
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
    return ret;
}

// Table in LDS: one wave (= one workgroup) per block, one block per CU.
__global__ __launch_bounds__(64) void k_fast2_lds(const uint8_t *__restrict__ src, size_t n_total,
                                                  uint32_t block_size, int dst_mode,
                                                  uint32_t dst_len_x, uint8_t *__restrict__ dst,
                                                  size_t dst_stride,
                                                  int32_t *__restrict__ result) {
    __shared__ uint16_t table[1 << 16];
    __shared__ uint32_t tag[kTagN];
    const int lane = lane_id();
    const size_t b = blockIdx.x;
    const size_t start = b * (size_t)block_size;
    const size_t left = n_total > start ? n_total - start : 0;
    const int n = (int)(left < block_size ? left : block_size);
    const int r = fast2_block<false>(src + start, n, dst + b * dst_stride,
                                     block_dlen2(dst_mode, n, dst_len_x), table, tag, lane);
    if (lane == 0) result[b] = r;
}

// Table in a global-memory slot per resident wave; many waves per CU.
__global__ __launch_bounds__(64) void k_fast2_gtab(const uint8_t *__restrict__ src, size_t n_total,
                                                   uint32_t block_size, size_t nb, int dst_mode,
                                                   uint32_t dst_len_x, uint8_t *__restrict__ dst,
                                                   size_t dst_stride,
                                                   int32_t *__restrict__ result,
                                                   uint16_t *__restrict__ ws) {
    __shared__ uint32_t tag[kTagN];
    const int lane = lane_id();
    uint16_t *table = ws + (size_t)blockIdx.x * (1 << 16);
    for (size_t b = blockIdx.x; b < nb; b += gridDim.x) {
        const size_t start = b * (size_t)block_size;
        const size_t left = n_total > start ? n_total - start : 0;
        const int n = (int)(left < block_size ? left : block_size);
        const int r = fast2_block<true>(src + start, n, dst + b * dst_stride,
                                        block_dlen2(dst_mode, n, dst_len_x), table, tag, lane);
        if (lane == 0) result[b] = r;
        __syncthreads();
    }
}

hipError_t launch_fast2(const uint8_t *src, size_t n, uint32_t bs, size_t nb, int dst_mode,
                        uint32_t dst_len_x, uint8_t *dst, size_t dst_stride, int32_t *result,
                        uint16_t *ws, size_t slots, hipStream_t st) {
    if (nb == 0) return hipSuccess;
    if (ws && slots) {
        const size_t grid = nb < slots ? nb : slots;
        hipLaunchKernelGGL(k_fast2_gtab, dim3((unsigned)grid), dim3(kWave), 0, st, src, n, bs, nb,
                           dst_mode, dst_len_x, dst, dst_stride, result, ws);
    } else {
        hipLaunchKernelGGL(k_fast2_lds, dim3((unsigned)nb), dim3(kWave), 0, st, src, n, bs,
                           dst_mode, dst_len_x, dst, dst_stride, result);
    }
    return hipGetLastError();
}

}  // namespace lz4hip
