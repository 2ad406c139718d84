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
__device__ __forceinline__ uint32_t tab_load(const uint16_t *t, uint32_t h) {
    return __hip_atomic_load(t + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Lane `l` of `old` takes the wave-uniform value `v` (v_cmp + v_cndmask: the VOP3
// form of v_writelane_b32 cannot take two SGPR operands on gfx9).
#define wrlane(old, v, l) ((lane == (l)) ? (v) : (old))

// Lane-private exact copy (no over-read): literals of one sequence.
__device__ __forceinline__ void lane_copy_exact(uint8_t *dst, const uint8_t *src, int len) {
    while (len >= 8) {
        const uint64_t a = ld64(src);
        __builtin_memcpy(dst, &a, 8);
        dst += 8;
        src += 8;
        len -= 8;
    }
    if (len & 4) {
        const uint32_t a = ld32(src);
        __builtin_memcpy(dst, &a, 4);
        dst += 4;
        src += 4;
    }
    if (len & 2) {
        uint16_t a;
        __builtin_memcpy(&a, src, 2);
        __builtin_memcpy(dst, &a, 2);
        dst += 2;
        src += 2;
    }
    if (len & 1) *dst = *src;
}

// Lane-private "length - 15" run (block.go:211-219,239-246); returns bytes written.
__device__ __forceinline__ int lane_len_tail(uint8_t *p, int v) {
    int k = 0;
    while (v >= 255) {
        p[k++] = 0xFF;
        v -= 255;
    }
    p[k++] = (uint8_t)v;
    return k;
}

// Exact-length global store of the low `len` (< 8) bytes of v.
__device__ __forceinline__ void store_tail8(uint8_t *p, uint64_t v, int len) {
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

__device__ __forceinline__ int len_tail_size(int len) {  // bytes after the token for `len`
    return len >= 0xF ? 1 + (len - 0xF) / 255 : 0;
}

__device__ __forceinline__ uint64_t rfl64(uint64_t v) {
    return ((uint64_t)rfl((uint32_t)(v >> 32)) << 32) | rfl((uint32_t)v);
}

// ---------------------------------------------------------------------------------------
// The common steps of the walk (phase B) as hand-written scalar code.
//
// State: si, anchor, P (lanes put since the window opened), FOUND (match lanes).  Per window:
// base (position of lane 0), sn, ATT (lanes that verify or may collide), RARE (lanes whose
// probe needs more than a lookup), mendv (per-lane match end).  Steps taken here are exactly
// the two common cases of fast.hip's phase B -- three clean misses (run-skipped while the
// skip is still 4: block.go:159) and a match whose lane answer is final -- in ~28 SALU
// instructions per step with two taken branches; what the compiler makes of the same C++
// is ~70 instructions and a dozen branches, and a lone wave issues one instruction every
// four cycles.  Returns 0: the window is done (si left it, or si >= sn), 1: the step at si
// needs the rare path (nothing of it has been done), 2: a match ended at or behind sn
// (block.go:250-252).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int walk_common_steps(int &si, int &anchor, uint64_t &P, uint64_t &FOUND,
                                                 const int base, const int sn, const uint64_t ATT,
                                                 const uint64_t RARE, const int mendv) {
    int st, t0, t1, t2, t3;
    uint64_t u0, u1;
    const uint64_t pat7 = 0x7777777777777777ull;
    // (wave-uniform by construction; the scalar operands below need the compiler to know it)
    si = (int)rfl((uint32_t)si);
    anchor = (int)rfl((uint32_t)anchor);
    P = rfl64(P);
    FOUND = rfl64(FOUND);
    const int base_s = (int)rfl((uint32_t)base), sn_s = (int)rfl((uint32_t)sn);
    const uint64_t att_s = rfl64(ATT), rare_s = rfl64(RARE);
    __asm__ volatile(
        "Ltop%=:\n\t"
        "s_sub_i32 %[t0], %[si], %[base]\n\t"          // a
        "s_cmp_gt_i32 %[t0], 61\n\t"
        "s_cbranch_scc1 Ldone%=\n\t"
        "s_cmp_ge_i32 %[si], %[sn]\n\t"
        "s_cbranch_scc1 Ldone%=\n\t"
        "s_lshr_b64 %[u0], %[att], %[t0]\n\t"
        "s_and_b64 %[u1], %[u0], 7\n\t"                // att3
        "s_cbranch_scc0 Lmiss%=\n\t"
        "s_lshr_b64 %[u0], %[rare], %[t0]\n\t"
        "s_and_b64 %[u0], %[u0], 7\n\t"
        "s_cbranch_scc1 Lrare%=\n\t"
        // ---- match at lane f = a + ctz(att3), answer final ----
        "s_ff1_i32_b64 %[t2], %[u1]\n\t"               // jj
        "s_add_i32 %[t3], %[t0], %[t2]\n\t"            // f
        "s_lshl_b64 %[u0], 3, %[t0]\n\t"
        "s_lshl_b64 %[u1], 7, %[t0]\n\t"
        "s_cmp_eq_u32 %[t2], 2\n\t"
        "s_cselect_b64 %[u0], %[u1], %[u0]\n\t"
        "s_or_b64 %[P], %[P], %[u0]\n\t"               // puts of the step: 2, or 3 when the third probe hit
        "v_readlane_b32 %[si], %[mend], %[t3]\n\t"     // si = end of the match
        "s_bitset1_b64 %[F], %[t3]\n\t"
        "s_mov_b32 %[anchor], %[si]\n\t"
        "s_cmp_ge_i32 %[si], %[sn]\n\t"
        "s_cbranch_scc1 Lfin%=\n\t"
        "s_sub_i32 %[t0], %[si], %[base]\n\t"
        "s_add_i32 %[t0], %[t0], -2\n\t"               // lane of put(si-2): block.go:254-255
        "s_cmp_gt_i32 %[t0], 63\n\t"
        "s_cbranch_scc1 Ldone%=\n\t"                   // belongs to the next window (the caller sees it)
        "s_bitset1_b64 %[P], %[t0]\n\t"
        "s_branch Ltop%=\n\t"
        // ---- three failed probes ----
        "Lmiss%=:\n\t"
        "s_sub_i32 %[t1], %[si], %[anchor]\n\t"
        "s_add_i32 %[t1], %[t1], 2\n\t"                // d0
        "s_cmp_gt_i32 %[t1], 64\n\t"
        "s_cbranch_scc1 Lfar%=\n\t"
        "s_add_i32 %[t2], %[si], 64\n\t"
        "s_cmp_ge_i32 %[t2], %[sn]\n\t"
        "s_cbranch_scc1 Lfar%=\n\t"
        // the skip is 4 for every step that starts in this window: take all clean steps at once
        "s_lshl_b64 %[u0], %[pat7], %[t0]\n\t"         // probed lanes a+4k+{0,1,2}
        "s_and_b64 %[u1], %[u0], %[att]\n\t"           // hot
        "s_sub_i32 %[t2], 61, %[t0]\n\t"
        "s_lshr_b32 %[t2], %[t2], 2\n\t"
        "s_add_i32 %[t2], %[t2], 1\n\t"                // kmax
        "s_ff1_i32_b64 %[t3], %[u1]\n\t"
        "s_sub_i32 %[t3], %[t3], %[t0]\n\t"
        "s_ashr_i32 %[t3], %[t3], 2\n\t"               // clean steps in front of the first hot lane
        "s_cmp_eq_u64 %[u1], 0\n\t"
        "s_cselect_b32 %[t3], %[t2], %[t3]\n\t"
        "s_min_i32 %[t3], %[t3], %[t2]\n\t"            // k >= 1
        "s_lshl_b32 %[t3], %[t3], 2\n\t"
        "s_add_i32 %[si], %[si], %[t3]\n\t"
        "s_add_i32 %[t3], %[t3], %[t0]\n\t"            // e = a + 4k <= 65: puts are pat below lane e
        "s_min_i32 %[t3], %[t3], 64\n\t"
        "s_add_i32 %[t3], %[t3], -1\n\t"
        "s_lshl_b64 %[u1], -1, %[t3]\n\t"
        "s_lshl_b64 %[u1], %[u1], 1\n\t"               // ones from lane e up (none for e = 64)
        "s_andn2_b64 %[u0], %[u0], %[u1]\n\t"
        "s_or_b64 %[P], %[P], %[u0]\n\t"
        "s_branch Ltop%=\n\t"
        "Lfar%=:\n\t"                                  // one step, adaptive skip (block.go:159)
        "s_ashr_i32 %[t1], %[t1], 7\n\t"
        "s_add_i32 %[t1], %[t1], 4\n\t"
        "s_add_i32 %[si], %[si], %[t1]\n\t"
        "s_lshl_b64 %[u0], 7, %[t0]\n\t"
        "s_or_b64 %[P], %[P], %[u0]\n\t"
        "s_branch Ltop%=\n\t"
        "Lrare%=:\n\t"
        "s_mov_b32 %[st], 1\n\t"
        "s_branch Lout%=\n\t"
        "Lfin%=:\n\t"
        "s_mov_b32 %[st], 2\n\t"
        "s_branch Lout%=\n\t"
        "Ldone%=:\n\t"
        "s_mov_b32 %[st], 0\n\t"
        "Lout%=:\n\t"
        : [si] "+s"(si), [anchor] "+s"(anchor), [P] "+s"(P), [F] "+s"(FOUND), [st] "=&s"(st),
          [t0] "=&s"(t0), [t1] "=&s"(t1), [t2] "=&s"(t2), [t3] "=&s"(t3), [u0] "=&s"(u0), [u1] "=&s"(u1)
        : [base] "s"(base_s), [sn] "s"(sn_s), [att] "s"(att_s), [rare] "s"(rare_s), [mend] "v"(mendv),
          [pat7] "s"(pat7)
        : "scc", "memory");
    return st;
}

__device__ __forceinline__ int block_dlen2(int dst_mode, int n, uint32_t dst_len_x) {
    return dst_mode == LZ4HIP_DST_FRAME  ? n
           : dst_mode == LZ4HIP_DST_BOUND ? (int)compress_bound32((uint32_t)n)
                                          : (int)dst_len_x;
}

}  // namespace lz4hip
