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
//     filter on that winner: it is not needed for the result, and is used
//     as a shortcut -- one load instead of a long prefix compare -- once a match is held.
//   * Emission (token, length runs, literal copy) is 64 lanes wide.
#include "common.h"

namespace lz4hip {

// Per-slot workspace: hashTable u32[65536], then the chain table, kHcLinks u16 per entry.
// The reference's chainTable[p & 0xffff] holds the previous position with p's hash
// (block.go:382-383).  Here the entry holds the distances from p to the 1st .. kL-th
// previous positions of its chain, as u16 (0 = none, or 65536 or more away: the
// reference's walk stops there anyway, `si-next < winSize`, :355) -- one 64-byte line for
// kL = 32 (the default: half the entry traffic of 64 links, and the chains that were the
// reason for 64 are scanned now, hc_scan), one link per lane of the lower half-wave.  Links of
// older positions never change, so links 2.. are link 1 plus the links of the previous
// head's entry at insertion time.  A walk then costs ONE dependent fetch per kL candidates
// instead of one per candidate, in the same candidate order, and the fetch of the next entry
// is in flight while the current batch is compared.
static_assert(kHcLinks == kWave || kHcLinks == kWave / 2, "one link per lane, of all lanes or of the lower half");
constexpr int kL = kHcLinks;
#ifndef LZ4HIP_HC_GATE
#define LZ4HIP_HC_GATE 8  // held match length from which the one-load gate precedes the prefix compare
#endif
#ifndef LZ4HIP_HC_FRONT
#define LZ4HIP_HC_FRONT 1  // misses in a row from which the lock-step miss front runs (0: never)
#endif
#ifndef LZ4HIP_HC_SCAN
#define LZ4HIP_HC_SCAN 256  // a batch of 64 chain members within this many positions: the walk turns into a scan (0: never)
#endif
#ifndef LZ4HIP_HC_FRONT_PEN
#define LZ4HIP_HC_FRONT_PEN 63  // most searches the front sits out after confirming nothing
#endif
constexpr int kHcSlotWords = (int)(kHcSlotBytes / 4);

// The position map behind the chain table: byte p & 0xffff holds hc_gen(p) if the parse JUMPED
// position p (a miss's step of two or more, :386-389), so that p is in no chain.  Every other
// position that the parse has passed is in the tables, so the map says for every position of
// the window whether it is a chain member -- what the scan of a dense chain needs to know
// (hc_scan).  Written only inside long runs of misses; the marks repeat every 8 MiB, beyond
// that the scan is only used where nothing was jumped.
__device__ __forceinline__ uint8_t *hc_map(uint16_t *chain_table) {
    return reinterpret_cast<uint8_t *>(chain_table + (size_t)kWinSize * kHcLinks);
}
__device__ __forceinline__ uint32_t hc_gen(int p) { return (((uint32_t)p >> 16) & 0x7Fu) + 1u; }
__device__ __forceinline__ void hc_mark_jumped(uint8_t *map, int first, int count, int lane) {
    for (int k = lane; k < count; k += kWave) map[(first + k) & kWinMask] = (uint8_t)hc_gen(first + k);
}

// hc_shift on both halves of a word at once (packed 16-bit operations): each half x becomes
// x + d0 if x != 0 and the sum stays below 65536, else 0.
typedef unsigned short us2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t hc_shift2(uint32_t m, uint32_t d0) {
    const us2_t x = __builtin_bit_cast(us2_t, m);
    const us2_t dd = {(unsigned short)d0, (unsigned short)d0};
    const us2_t one = {1, 1};
    const us2_t sum = x + dd;                                         // wraps at 65536
    const us2_t wrapped = sum ^ __builtin_elementwise_max(sum, x);    // 0 where it did not wrap
    const us2_t keep = __builtin_elementwise_min(wrapped, one) - one;  // 0xFFFF where it did not
    const us2_t nz = (us2_t){0, 0} - __builtin_elementwise_min(x, one);  // 0xFFFF where x != 0
    return __builtin_bit_cast(uint32_t, (us2_t)(sum & keep & nz));
}

// Links of a position inserted behind `head` (distance d0 > 0, inside the window), given
// lane j-1's link of head's entry in `inh_prev` (lane 0: unused): link 0 = d0, link j = d0 +
// head's link j-1 while that stays inside the window.
__device__ __forceinline__ uint32_t hc_link(uint32_t d0, uint32_t inh_prev, int lane) {
    if (lane == 0) return d0;
    return (inh_prev != 0 && d0 + inh_prev < (uint32_t)kWinSize) ? d0 + inh_prev : 0u;
}

// chainTable[p] = hashTable[hh]; hashTable[hh] = p  (block.go:382-383,401-402) for one
// position, by the whole wave, reading head and head's entry (two dependent fetches; the
// search path below avoids them by reusing what the search has just read).
__device__ __forceinline__ void hc_insert(uint32_t *hash_table, uint16_t *chain_table, uint32_t hh,
                                          int p, int lane) {
    const int head = (int)hash_table[hh];
    uint32_t l = 0;
    if (head > 0 && p - head < kWinSize) {
        uint32_t hl = 0;
        if (lane > 0 && lane < kL) hl = chain_table[(size_t)(head & kWinMask) * kHcLinks + lane - 1];
        l = hc_link((uint32_t)(p - head), hl, lane);
    }
    if (lane < kL) chain_table[(size_t)(p & kWinMask) * kHcLinks + lane] = (uint16_t)l;
    __syncthreads();  // every lane has read the old head
    if (lane == 0) hash_table[hh] = (uint32_t)p;
}

// Prefix length of block.go:361-372 for one candidate (lane-private loop): 8-byte strides
// while ml < lim, so ml may overshoot lim by up to 7.  Most candidates differ inside the
// first stride; the few long ones decide how long the wave waits, so after the first stride
// the compare runs 32 bytes per round trip (four strides in flight) while 32 more bytes are
// inside lim -- the same first differing byte is found either way.
__device__ __forceinline__ int hc_prefix_len(const uint8_t *s, int next, int si, int lim, uint64_t w_si,
                                             bool stride1_equal = false) {
    int ml = 8;
    if (!stride1_equal) {
        const uint64_t x = ld64(s + next) ^ w_si;  // w_si: the 8 bytes at si (every lane would load the same)
        if (x != 0) return __builtin_ctzll(x) >> 3;
        ml = 8;
    }
    while (ml + 32 <= lim) {
        const uint8_t *a = s + next + ml, *b = s + si + ml;
        const uint64_t x0 = ld64(a) ^ ld64(b), x1 = ld64(a + 8) ^ ld64(b + 8);
        const uint64_t x2 = ld64(a + 16) ^ ld64(b + 16), x3 = ld64(a + 24) ^ ld64(b + 24);
        if (x0) return ml + (__builtin_ctzll(x0) >> 3);
        if (x1) return ml + 8 + (__builtin_ctzll(x1) >> 3);
        if (x2) return ml + 16 + (__builtin_ctzll(x2) >> 3);
        if (x3) return ml + 24 + (__builtin_ctzll(x3) >> 3);
        ml += 32;
    }
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

// Maximum over the wave, in every lane: DPP row shifts and row broadcasts leave it in lane 63
// (six v_max with a DPP operand; the shuffle form is six dependent LDS permutes per batch).
__device__ __forceinline__ int wave_max(int v) {
#define LZ4HIP_MAX_STEP(ctrl, rows)                                              \
    {                                                                            \
        const int t = __builtin_amdgcn_update_dpp(v, v, ctrl, rows, 0xF, false); \
        v = t > v ? t : v;                                                       \
    }
    LZ4HIP_MAX_STEP(0x111, 0xF)  // row_shr:1
    LZ4HIP_MAX_STEP(0x112, 0xF)  // row_shr:2
    LZ4HIP_MAX_STEP(0x114, 0xF)  // row_shr:4
    LZ4HIP_MAX_STEP(0x118, 0xF)  // row_shr:8
    LZ4HIP_MAX_STEP(0x142, 0xA)  // row_bcast:15 -> rows 1, 3
    LZ4HIP_MAX_STEP(0x143, 0xC)  // row_bcast:31 -> rows 2, 3
#undef LZ4HIP_MAX_STEP
    return __builtin_amdgcn_readlane(v, kWave - 1);
}

// Inserts positions p0 .. p0+n-1 in order (block.go:394-405 without the window clamp:
// every position's 4-byte value is then its own bytes), 64 per round.  Sequential
// semantics: each position links to the latest earlier position with its hash -- inside the
// round if there is one (its "twin"), else the table's head -- and the table keeps the last
// one.  Lane i first gathers head and head's entry of position i into row i of an LDS
// staging area (all 64 positions' fetches in flight together).  Then the positions are
// finished in order, the wave one link per lane: from its own staged row, or from the twin's
// FINISHED row (a run of equal bytes resolves at LDS latency, no memory round trip), and each
// finished row goes out as one 128-byte store.
//
// `pre`: the search that follows the match (at pre->pos) rides along with the first round: its
// 4 bytes, its head and the head's entry are fetched beside the round's own three fetches, so
// that search starts with everything in registers.  They are what the search itself would
// read unless a position of the run has its hash (then the run moved that head): ok = false.
struct HcPre {
    int pos;     // < 0: none
    uint64_t w;  // the 8 source bytes at pos
    uint32_t h, l;
    int head;
    bool ok;
};

// One round: up to 64 positions in increasing order, lane i <-> pos (with its hash hh), not
// necessarily consecutive.  A position without a twin depends on nothing in the round: its
// lane gathers head and head's entry into REGISTERS (all fetches in flight together), derives
// its own 64 links there and puts the finished row into LDS.  Twins are finished in order
// afterwards, the wave one link per lane, from their twin's finished row (a run of equal bytes
// resolves at LDS latency).  The finished rows leave 16 bytes per lane.
constexpr int kRowStride = kHcLinks + 8;  // u16 units: 144-byte rows spread the lanes' 16-byte accesses over the banks

__device__ __forceinline__ uint32_t hc_shift(uint32_t x, uint32_t d0) {  // one link of the head, seen from d0 further on
    // (x != 0 && x + d0 < kWinSize) ? x + d0 : 0, without condition masks (64 of these per lane and
    // round: as compares they fill the scalar registers with masks): x == 0 counts as 65536
    const uint32_t y = x - 1u;
    const uint32_t t = (y < 0xFFFFu ? y : 0xFFFFu) + d0 + 1u;  // < 2^17
    return t & ((t >> 16) - 1u);
}

__device__ __forceinline__ void hc_insert_round(uint32_t *hash_table, uint16_t *chain_table,
                                                const uint8_t *__restrict__ s, const int p,
                                                const uint32_t hh_in, const int cnt, const int lane,
                                                uint16_t *rows, HcPre &pre, const bool ride) {
    typedef uint4 __attribute__((may_alias)) uint4_a;
    const bool act = lane < cnt;
    const uint32_t hh = act ? hh_in : 0xFFFFFFFFu;
    if (ride) {
        pre.h = hash4((uint32_t)pre.w);
        pre.head = (int)hash_table[pre.h];
        pre.l = 0;
        pre.ok = true;
    }
    if (pre.pos >= 0 && ballot(act && hh == pre.h)) pre.ok = false;
    int twin = -1;        // nearest earlier lane with my hash
    bool later = false;   // a later lane has my hash: it, not I, ends up in the table
    {
        // Lanes that share their hash with another lane are few (none in most rounds of text): they
        // are found through a small table in the rows' LDS (still free here) -- every lane writes
        // its number into the slot of its hash, a lane that reads back another number flags the
        // slot, and the lanes of flagged slots (same hash, or only the same slot) are then
        // compared one by one, in lane order, instead of all 64.
        uint16_t *slot = rows + (hh & 1023u);
        if (act) *slot = (uint16_t)lane;
        __asm__ volatile("" ::: "memory");
        const uint32_t seen = act ? *slot : (uint32_t)lane;
        __asm__ volatile("" ::: "memory");
        if (act && seen != (uint32_t)lane) *slot = 0x8000u;
        __asm__ volatile("" ::: "memory");
        uint64_t G = ballot(act && (*slot & 0x8000u));
        __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        while (G) {
            const int i = __builtin_ctzll(G);
            G &= G - 1;
            const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)hh, i);
            if (act && hh == hi) {
                if (lane > i) twin = i;
                if (lane < i) later = true;
            }
        }
    }
    const int twin_pos = __shfl(p, twin < 0 ? lane : twin);  // every lane takes part in the shuffle
    // ---- positions without a twin: head's entry -> own entry, in registers ----
    if (act && twin < 0) {
        uint32_t e[kHcLinks / 2];  // two links per word
#pragma unroll
        for (int k = 0; k < kHcLinks / 2; k++) e[k] = 0;
        const int head = (int)hash_table[hh];
        if (head > 0 && p - head < kWinSize) {
            const uint32_t d0 = (uint32_t)(p - head);
            const uint4_a *g = (const uint4_a *)(chain_table + (size_t)(head & kWinMask) * kHcLinks);
#pragma unroll
            for (int q = 0; q < kHcLinks / 8; q++) {
                const uint4 v = g[q];
                e[4 * q] = v.x, e[4 * q + 1] = v.y, e[4 * q + 2] = v.z, e[4 * q + 3] = v.w;
            }
            // word k of the new entry = links 2k, 2k+1 = the head's links 2k-1, 2k moved by d0
#pragma unroll
            for (int k = kHcLinks / 2 - 1; k > 0; k--)
                e[k] = hc_shift2(__builtin_amdgcn_alignbit(e[k], e[k - 1], 16), d0);
            e[0] = d0 | hc_shift2(e[0] << 16, d0);
        }
        uint4_a *r = (uint4_a *)(rows + lane * kRowStride);
#pragma unroll
        for (int q = 0; q < kHcLinks / 8; q++) r[q] = make_uint4(e[4 * q], e[4 * q + 1], e[4 * q + 2], e[4 * q + 3]);
    }
    if (ride && pre.head > 0 && pre.pos - pre.head < kWinSize)
        pre.l = lane < kL ? chain_table[(size_t)(pre.head & kWinMask) * kHcLinks + lane] : 0;
    __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // ---- twins, in position order: from the twin's finished row (LDS operations of one wave
    // execute in order, so a row written by one step is what the next step reads) ----
    uint64_t TW = ballot(act && twin >= 0);
    while (TW) {
        const int i = __builtin_ctzll(TW);
        TW &= TW - 1;
        const int t = __builtin_amdgcn_readlane(twin, i);
        // distance to the twin (0: out of the window -- the reference's walk stops there, :355)
        uint32_t d = (uint32_t)(__builtin_amdgcn_readlane(p, i) - __builtin_amdgcn_readlane(twin_pos, i));
        if (d >= (uint32_t)kWinSize) d = 0;
        uint32_t l = 0;
        if (d != 0) {
            uint32_t inh = 0;
            if (lane > 0 && lane < kL) inh = rows[t * kRowStride + lane - 1];
            l = hc_link(d, inh, lane);
        }
        if (lane < kL) rows[i * kRowStride + lane] = (uint16_t)l;
        __asm__ volatile("" ::: "memory");
    }
    __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // ---- the finished rows go out, 16 bytes per lane ----
#pragma unroll
    for (int it = 0; it < kHcLinks / 8; it++) {
        const int c = lane + it * kWave, row = c / (kL / 8), col = c % (kL / 8);
        const int prow = __shfl(p, row);  // all lanes shuffle; only the rows that exist are stored
        if (row < cnt) {
            const uint4 v = *(const uint4_a *)(rows + row * kRowStride + col * 8);
            *(uint4_a *)(chain_table + (size_t)(prow & kWinMask) * kHcLinks + col * 8) = v;
        }
    }
    if (act && !later) hash_table[hh] = (uint32_t)p;
    __syncthreads();  // the next round gathers what this one stored, and reuses the rows
}

__device__ __forceinline__ void hc_insert_run(uint32_t *hash_table, uint16_t *chain_table,
                                              const uint8_t *__restrict__ s, int p0, int n,
                                              int lane, uint16_t *rows, HcPre &pre) {
    for (int base = 0; base < n; base += kWave) {
        const int cnt = n - base < kWave ? n - base : kWave;
        const int p = p0 + base + lane;
        const bool ride = base == 0 && pre.pos >= 0;
        if (ride) pre.w = ld64(s + pre.pos);
        const uint32_t hh = lane < cnt ? hash4(ld32(s + p)) : 0u;
        hc_insert_round(hash_table, chain_table, s, p, hh, cnt, lane, rows, pre, ride);
    }
}

// The rest of a dense chain's walk as a scan of the source (see k_hc_compress): the members of h's
// chain below `top` are the window's positions whose 4 bytes hash to h (except, where positions
// were jumped, those that the position map shows), in descending order; of them only those
// with si's own 4 bytes can be matches, so only those are looked at.  Returns the winner as
// m_len | offset << 32.
__device__ __forceinline__ uint64_t hc_scan(const uint8_t *__restrict__ s, const uint8_t *__restrict__ map, int n,
                                                      int si, int next, uint64_t w, int lim, int ml_max,
                                                      int m_len, int offset, bool jumped, int lane) {
    uint64_t gate_w = 0;
    if (m_len >= LZ4HIP_HC_GATE) gate_w = ld64(s + si + m_len - 7);
        const int lo = si - (kWinSize - 1) > 1 ? si - (kWinSize - 1) : 1;
        int top = next;
        // the first chunk's bytes; inside the loop the next chunk's are asked for
        // before this chunk is looked at
        uint64_t a0 = 0, a1 = 0;
        {
            const int pb = (top & ~511) + lane * 8;
            if (pb <= top) {
                a0 = ld64(s + pb);
                a1 = ld64(s + pb + 8);
            }
        }
        while (top >= lo && m_len < ml_max) {
            const int cb = top & ~511, pb = cb + lane * 8;
            // the gate's bytes of the lane's 8 positions (index m_len - 7 .. m_len + 7 from
            // pb) in one stretch as well, where the block has them
            const bool gated = m_len >= LZ4HIP_HC_GATE;
            const bool stretch = gated && top + m_len + 10 <= n;
            uint64_t b0 = 0, b1 = 0;
            if (stretch && pb <= top) {
                b0 = ld64(s + pb + m_len - 7);
                b1 = ld64(s + pb + m_len + 1);
            }
            uint64_t mk = 0;  // the map's bytes of the lane's 8 positions
            if (jumped && pb <= top) mk = *reinterpret_cast<const uint64_t *>(map + (pb & kWinMask));
            uint64_t na0 = 0, na1 = 0;
            {
                const int npb = cb - 512 + lane * 8;
                if (cb - 1 >= lo) {
                    na0 = ld64(s + npb);
                    na1 = ld64(s + npb + 8);
                }
            }
            // Per lane, one bit per position pb + j (the work of this loop is what a dense
            // chain costs now -- all resident waves scan at once -- so it is kept to 32-bit
            // operations: the 4 bytes at offset k of a 16-byte stretch are one v_alignbyte).
            const int jlo = lo - pb > 0 ? lo - pb : 0, jhi = top - pb < 7 ? top - pb : 7;
            uint32_t memb = jhi >= jlo ? ((2u << jhi) - 1u) & ~((1u << jlo) - 1u) : 0u;  // inside [lo, top]
            if (jumped) {  // ... and not jumped by the parse
                const uint32_t g = hc_gen(pb), mlo = (uint32_t)mk, mhi = (uint32_t)(mk >> 32);
                uint32_t out = 0;
#pragma unroll
                for (int j = 0; j < 8; j++)
                    out |= (((j < 4 ? mlo : mhi) >> (8 * (j & 3))) & 0xFFu) == g ? 1u << j : 0u;
                memb &= ~out;
            }
            const uint32_t wlo = (uint32_t)w, whi = (uint32_t)(w >> 32);
            uint32_t V[12];
            {
                const uint32_t A[4] = {(uint32_t)a0, (uint32_t)(a0 >> 32), (uint32_t)a1, (uint32_t)(a1 >> 32)};
#pragma unroll
                for (int k = 0; k < 12; k++)
                    V[k] = (k & 3) == 0 ? A[k >> 2] : __builtin_amdgcn_alignbyte(A[(k >> 2) + 1], A[k >> 2], k & 3);
            }
            // Only a member whose first 4 bytes are si's can be a match (:361-373), and such a position
            // has si's hash: the hash of the others need not be looked at -- they change nothing.
            // bit j of need: position pb + j is equal to si over 8 bytes and at the gate
            uint32_t need = 0;
            if (stretch) {
                const uint32_t B[4] = {(uint32_t)b0, (uint32_t)(b0 >> 32), (uint32_t)b1, (uint32_t)(b1 >> 32)};
                const uint32_t glo = (uint32_t)gate_w, ghi = (uint32_t)(gate_w >> 32);
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int k = j + 4;
                    const uint32_t g0 = (j & 3) == 0 ? B[j >> 2] : __builtin_amdgcn_alignbyte(B[(j >> 2) + 1], B[j >> 2], j & 3);
                    const uint32_t g1 = (k & 3) == 0 ? B[k >> 2] : __builtin_amdgcn_alignbyte(B[(k >> 2) + 1], B[k >> 2], k & 3);
                    need |= (V[j] == wlo && V[k] == whi && g0 == glo && g1 == ghi) ? 1u << j : 0u;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; j++) need |= (V[j] == wlo && V[j + 4] == whi) ? 1u << j : 0u;
            }
            need &= memb;
            int bml = 0, bj = 0;  // the lane's best so far
            if (m_len < 8) {
                // members that differ from si inside the first 8 bytes, but behind byte m_len
                const uint32_t himask = m_len >= 7 ? ~0u : m_len >= 4 ? (1u << (8 * (m_len - 3))) - 1u : 0u;
#pragma unroll
                for (int j = 7; j >= 0; j--) {
                    const uint32_t xh = V[j + 4] ^ whi;
                    if ((memb >> j & 1u) && V[j] == wlo && xh != 0 && (xh & himask) == 0) {
                        const int ml1 = 4 + (int)(__builtin_ctz(xh) >> 3);
                        if (ml1 > bml) bml = ml1, bj = j;  // descending j: a tie keeps the nearer
                    }
                }
                if (bml <= m_len) bml = 0;
            }
            // the candidates equal to si over 8 bytes (and at the gate): the full compare
            while (ballot(need != 0)) {
                if (need) {
                    const int j = 31 - __builtin_clz(need);
                    need &= ~(1u << j);
                    const int q = pb + j;
                    const bool gate = !gated || stretch || ld64(s + q + m_len - 7) == gate_w;
                    if (gate) {
                        const int ml2 = hc_prefix_len(s, q, si, lim, w, true);
                        if (ml2 > m_len && ml2 > bml) bml = ml2, bj = j;
                    }
                }
            }
            if (ballot(bml != 0)) {
                // the longest; among equals the nearest = the highest position (lengths reach 2^30 in
                // a single-block call: no room for the position in one 32-bit key)
                m_len = wave_max(bml);
                const int sel = 63 - __builtin_clzll(ballot(bml == m_len));
                offset = si - (cb + sel * 8 + __builtin_amdgcn_readlane(bj, sel));
                if (m_len >= LZ4HIP_HC_GATE) gate_w = ld64(s + si + m_len - 7);
            }
            top = cb - 1;
            a0 = na0;
            a1 = na1;
        }
    return (uint64_t)(uint32_t)m_len | ((uint64_t)(uint32_t)offset << 32);
}

// -DLZ4HIP_HC_PROF (profiling builds only, tools/prof): shader-clock cycles summed over all waves,
// [0..3] searches by batches walked (none, one, 2-8, more), [4] waiting-position bookkeeping and
// the per-search store, [5] insertion rounds, [6] emission, [7] set-up and tail; counts: [8..11]
// searches of those four kinds, [12] batches, [13] matches, [14] insertion rounds.
// (Since the round-3 kernel sits at the 128-register cap the 16 counters go to scratch and a
// marked build runs 3-4x slower: read the counts, not the shares.)
#ifdef LZ4HIP_HC_PROF2  // experiment marks (LZ4HIP_PROF_RAW=1 prints the slots): see HC_U below
#define LZ4HIP_HC_PROF
#endif
#ifdef LZ4HIP_HC_PROF
__device__ unsigned long long g_hc_prof[16];
#define HC_T(i)                                                       \
    do {                                                              \
        const unsigned long long now_ = __builtin_readcyclecounter(); \
        prof[i] += now_ - last_t;                                     \
        last_t = now_;                                                \
    } while (0)
#define HC_N(i, v) prof[i] += (unsigned long long)(v)
#ifdef LZ4HIP_HC_PROF2
#undef HC_T
#undef HC_N
#define HC_T(i)
#define HC_N(i, v)
#define HC_U(i)                                                       \
    do {                                                              \
        const unsigned long long now_ = __builtin_readcyclecounter(); \
        prof[i] += now_ - last_t;                                     \
        last_t = now_;                                                \
    } while (0)
#define HC_M(i, v) prof[i] += (unsigned long long)(v)
#define HC_V(i)                                               \
    do {                                                      \
        __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
        HC_U(i);                                              \
    } while (0)
#endif
#else
#define HC_T(i)
#define HC_N(i, v)
#endif
#ifndef HC_U
#define HC_U(i)
#define HC_M(i, v)
#define HC_V(i)
#endif

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4))) void k_hc_compress(const uint8_t *__restrict__ src,
                                                    size_t n_total, uint32_t block_size,
                                                    uint32_t depth, int dst_mode, uint32_t dst_len_x,
                                                    uint8_t *__restrict__ dst, size_t dst_stride,
                                                    int32_t *__restrict__ result,
                                                    uint32_t *__restrict__ ws, size_t nb,
                                                    uint32_t *__restrict__ next_block) {
    __shared__ __attribute__((aligned(16))) uint16_t rows[kWave * kRowStride];  // hc_insert_run's staging
    const int lane = lane_id();
    uint32_t *hash_table = ws + (size_t)blockIdx.x * kHcSlotWords;
    uint16_t *chain_table = reinterpret_cast<uint16_t *>(hash_table + (1 << 16));
    if (depth == 0) depth = kWinSize;  // block.go:343-345

    // resident waves take their next block from a counter when they finish one (as
    // k_fast_gtab does; the reference's goroutines, writer.go:130-142); without a counter
    // the blocks are dealt round-robin
    auto take = [&]() -> size_t {
        uint32_t t = 0;
        if (lane == 0) t = atomicAdd(next_block, 1u);
        return (size_t)rfl(t);
    };
    for (size_t b = next_block ? take() : (size_t)blockIdx.x; b < nb;
         b = next_block ? take() : b + gridDim.x) {
        const uint8_t *s = src + b * (size_t)block_size;
        const size_t start = b * (size_t)block_size;
        const size_t left = n_total > start ? n_total - start : 0;
        const int n = (int)(left < block_size ? left : block_size);
        uint8_t *d = dst + b * dst_stride;
        const int bound = (int)compress_bound32((uint32_t)n);
        const int dlen = dst_mode == LZ4HIP_DST_FRAME ? n
                         : dst_mode == LZ4HIP_DST_BOUND ? bound : (int)dst_len_x;
        const bool not_compressible = dlen < bound;  // :330

        {  // :320-324 zero both tables; here only the heads need it: a chain entry is only
            // ever reached from a head or a link written in this block, after the entry itself
            uint4 *t4 = reinterpret_cast<uint4 *>(hash_table);
            for (int k = lane; k < (1 << 16) / 4; k += kWave) t4[k] = make_uint4(0, 0, 0, 0);
            uint4 *m4 = reinterpret_cast<uint4 *>(hc_map(chain_table));  // no position was jumped
#pragma unroll 1
            for (int k = lane; k < kWinSize / 16; k += kWave) m4[k] = make_uint4(0, 0, 0, 0);
        }
        __syncthreads();

        int si = 0, di = 0, anchor = 0;
        const int sn = n - kMfLimit;
        bool failed = false;
#ifdef LZ4HIP_HC_PROF
        unsigned long long prof[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        unsigned long long last_t = __builtin_readcyclecounter();
        int nbatch = 0;
#endif

        // The positions under a match are not inserted at once but collected, up to 64 of them
        // (lane k: pd_pos, pd_h), and inserted together in ONE round -- its three dependent
        // fetches are what an insertion costs, however few positions it has, and text matches
        // are short.  Nothing can tell the difference: a chain only holds positions of one
        // hash, so a search flushes the pending positions first if one of them has ITS hash
        // (or if the oldest is 32 KiB back: its entry slot must not wait for the ring to come
        // round), and positions of different hashes may be inserted in any order.
        int pd_n = 0, pd_pos = 0;
        uint32_t pd_h = 0;
        HcPre pre;
        pre.pos = -1;
        pre.ok = false;
        pre.w = 0;
        pre.h = pre.l = 0;
        pre.head = 0;
        // The source window: lane k holds the 8 bytes at wb + k (those that the block has), asked
        // for when a match is known, at the position behind it, and not waited for: what the NEXT
        // match needs of the source -- the 4-byte values of the positions it covers, the bytes of
        // the search behind it -- is then in registers when that match is found (text: a match
        // every 10 bytes; each of these was a round trip the wave sat out).
        uint64_t W = 0;
        int wb = -(1 << 20);  // (nothing fetched yet)
        auto win_fetch = [&](int base) {
            wb = base;
            W = 0;
            if (base + lane + 8 <= n) W = ld64(s + base + lane);
        };
        uint64_t w_nx = 0;  // the 8 bytes at w_nx_pos, taken from the window before it moved on
        int w_nx_pos = -1;
        auto rl64 = [&](uint64_t v, int l) {
            return (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l) |
                   ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l) << 32);
        };
        // The last position that may NOT be in the tables although the parse has passed it: one that
        // a miss's step of two or more jumped (:386-389), or one inserted under a mixed value behind
        // a match longer than the window (:399-400).  64 KiB behind it, the chain of a hash is
        // exactly the window's positions with that hash (see the scan below).
        int last_skip = -kWinSize;
        int last_mixed = -kWinSize;  // ... of the second kind: there the map cannot help
        int miss_run = 0;          // searches that missed since the last match
        int fr_pen = 0, fr_skip = 0;  // the miss front backs off where it confirms nothing (text)
        // A sequence's literals (up to 64) are read at emission and stored at the end of the next
        // search, one byte per lane: a wave's loads return in order, so they have arrived with
        // that search's own loads, and the emission itself waits for nothing.
        int lit_n = 0, lit_di = 0;
        uint32_t lit = 0;
        auto put_literals = [&]() {
            if (lit_n) {
                if (lane < lit_n) d[lit_di + lane] = (uint8_t)lit;
                lit_n = 0;
            }
        };
        while (si < sn) {
#if LZ4HIP_HC_FRONT > 0
            // ---- The miss front.  77 % of the searches of kernel-image data find nothing, and each
            // is a chain of dependent fetches (head, head's entry, candidate bytes) with nothing
            // else for the wave to do.  Inside a run of misses the next positions are known
            // (:386-389), so eight searches are taken in lock-step, eight lanes each: the fetches of
            // one level fly together (a wave's loads return in order, so only loads of one level
            // overlap well).  The front only CONFIRMS misses: a search whose chain has at most 16
            // candidates inside the window, none of them equal to si over 4 bytes, is finished --
            // entry and head stored as hc_insert would -- and the leading run of such searches is
            // committed.  The first search that has a match, a longer chain, the hash of an
            // earlier search of the step (that one's insertion moves its head) or of a waiting
            // position goes to the full search below, with its lines already in the caches.
            if (miss_run >= LZ4HIP_HC_FRONT && fr_skip == 0) {
                typedef uint4 __attribute__((may_alias)) uint4_a;
                int lq = lane;  // opaque: what hangs off g and t is not kept in registers across the searches
                __asm__ volatile("" : "+v"(lq));
                const int g = lq >> 3, t = lq & 7;
                int P = si, q = si;
#pragma unroll
                for (int i = 1; i <= 8; i++) {
                    q += 1 + ((q - anchor) >> kAdaptSkipLog);
                    if (g >= i) P = q;
                }
                bool bad = P >= sn;
                // the bytes at P from the window (fetched anew when the step leaves it)
                if (ballot(!bad && (P - wb < 0 || P - wb >= kWave))) win_fetch(si);
                const int wl = bad ? 0 : P - wb;
                const uint64_t w = (uint64_t)(uint32_t)__shfl((int)(uint32_t)W, wl) |
                                   ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(W >> 32), wl) << 32);
                if (wl < 0 || wl >= kWave) bad = true;  // (a step of more than 9 bytes per search)
                const uint32_t hq = hash4((uint32_t)w);
                {
                    const uint32_t ho = (uint32_t)__shfl((int)hq, t * 8);  // the hash of group t
                    if (t < g && ho == hq) bad = true;
                }
                if (pd_n) {
                    if (P - __builtin_amdgcn_readlane(pd_pos, 0) >= (kWinSize >> 1)) bad = true;
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)hq, 8 * i);
                        if (ballot(lane < pd_n && pd_h == hi) && g == i) bad = true;
                    }
                }
                int head = 0;
                if (!bad) head = (int)hash_table[hq];
                const bool inw = !bad && head > 0 && P - head < kWinSize;
                uint4 e = make_uint4(0, 0, 0, 0);  // the lane's kL/8 links of the head's entry, from link t*kL/8 on
                uint32_t c0 = 0, c1 = 0;           // links t-1 and t+7: candidates t and t+8
                if (inw) {
                    const uint16_t *ent = chain_table + (size_t)(head & kWinMask) * kHcLinks;
                    if (kL == kWave) {
                        e = *(const uint4_a *)(ent + 8 * t);
                    } else {
                        const uint2 e2 = *(const uint2 __attribute__((may_alias)) *)(ent + 4 * t);
                        e.x = e2.x, e.y = e2.y;
                    }
                    if (t > 0) c0 = ent[t - 1];
                    c1 = ent[t + 7];
                }
                const int cand0 = t == 0 ? head : head - (int)c0;
                // (the walk looks at `depth` candidates at most: :347-356)
                const bool v0 = inw && (uint32_t)t < depth && (t == 0 || (c0 != 0 && P - cand0 < kWinSize));
                const int cand1 = head - (int)c1;
                const bool v1 = inw && (uint32_t)t + 8u < depth && c1 != 0 && P - cand1 < kWinSize;
                uint32_t x0 = 1, x1 = 1;
                if (v0) x0 = ld32(s + cand0) ^ (uint32_t)w;
                if (v1) x1 = ld32(s + cand1) ^ (uint32_t)w;
                // a 17th candidate (link 15) inside the window: not for the front
                if (inw && t == (kL == kWave ? 1 : 3) && depth > 16u) {
                    const uint32_t l15 = (kL == kWave ? e.w : e.y) >> 16;
                    if (l15 != 0 && P - (head - (int)l15) < kWinSize) bad = true;
                }
                const uint32_t prev7 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)((kL == kWave ? e.w : e.y) >> 16), 0x111, 0xF, 0xF, false);
                if (x0 == 0 || x1 == 0) bad = true;  // four equal bytes: a match
                const uint64_t B = ballot(bad);
                const int m = B ? (int)(__builtin_ctzll(B) >> 3) : 8;
                if (g < m) {
                    uint4 o = make_uint4(0, 0, 0, 0);
                    if (inw) {
                        const uint32_t d0 = (uint32_t)(P - head);
                        o.x = (t == 0 ? d0 : hc_shift(prev7, d0)) | (hc_shift(e.x & 0xFFFFu, d0) << 16);
                        o.y = hc_shift(e.x >> 16, d0) | (hc_shift(e.y & 0xFFFFu, d0) << 16);
                        if (kL == kWave) {
                            o.z = hc_shift(e.y >> 16, d0) | (hc_shift(e.z & 0xFFFFu, d0) << 16);
                            o.w = hc_shift(e.z >> 16, d0) | (hc_shift(e.w & 0xFFFFu, d0) << 16);
                        }
                    }
                    if (kL == kWave)
                        *(uint4_a *)(chain_table + (size_t)(P & kWinMask) * kHcLinks + 8 * t) = o;
                    else
                        *(uint2 __attribute__((may_alias)) *)(chain_table + (size_t)(P & kWinMask) * kHcLinks + 4 * t) = make_uint2(o.x, o.y);
                    if (t == 0) hash_table[hq] = (uint32_t)P;
                    // the positions that the step behind this search jumps
                    for (int k = t; k < ((P - anchor) >> kAdaptSkipLog); k += 8)
                        hc_map(chain_table)[(P + 1 + k) & kWinMask] = (uint8_t)hc_gen(P + 1 + k);
                }
                HC_N(15, m);
                HC_T(7);
                HC_U(6);
                if (m == 0) {
                    fr_pen = 2 * fr_pen + 1;
                    if (fr_pen > LZ4HIP_HC_FRONT_PEN) fr_pen = LZ4HIP_HC_FRONT_PEN;
                    fr_skip = fr_pen;
                } else if (m == 1) {
                    fr_skip = fr_pen;
                } else {
                    fr_pen >>= 1;
                }
                const int si_new = m == 8 ? q : __builtin_amdgcn_readlane(P, (8 * m) & 63);
                // steps only grow inside a run of misses: the last one says whether any jumped
                if (m > 0 && si_new - __builtin_amdgcn_readlane(P, (8 * (m - 1)) & 63) > 1) last_skip = si_new - 1;
                si = si_new;
                if (m == 8) continue;
                if (si >= sn) break;
            }
            if (fr_skip > 0) fr_skip--;
#endif
            // fetched beside the last match's insertion
            const bool ridden = pre.ok && pre.pos == si;
            pre.ok = false;
            // the bytes at si: kept from the last match, or in the window
            uint64_t w = pre.w;
            if (!ridden) {
                w = w_nx;
                if (w_nx_pos != si) {
                    if (si - wb < 0 || si - wb >= kWave) win_fetch(si);
                    w = rl64(W, si - wb);
                }
            }
            uint32_t match = (uint32_t)w;
            const uint32_t h = ridden ? pre.h : hash4(match);
            if (pd_n && (ballot(lane < pd_n && pd_h == h) ||
                         si - __builtin_amdgcn_readlane(pd_pos, 0) >= (kWinSize >> 1))) {
                HcPre none;
                none.pos = -1;
                none.ok = false;
                none.w = 0;
                none.h = none.l = 0;
                none.head = 0;
                HC_U(5);
                hc_insert_round(hash_table, chain_table, s, pd_pos, pd_h, pd_n, lane, rows, none, false);
                pd_n = 0;
                HC_U(7);
                HC_M(14, 1);
            }
            HC_T(4);
            HC_U(5);
#ifdef LZ4HIP_HC_PROF
            nbatch = 0;
#endif
            int m_len = 0, offset = 0;
            uint32_t my_link = 0;  // lane j: link j of si's own chain entry (from the first batch)
            {
                const int lim = sn - si;
                int next = pre.head;
                if (!ridden) {  // one lane fetches, the wave takes it from there
                    uint32_t t = 0;
                    if (lane == 0) t = hash_table[h];
                    next = (int)rfl(t);
                }
                HC_V(9);
                uint32_t tries = depth;
                // A prefix length never exceeds lim rounded up to 8 (:363-372) and only a
                // strictly longer candidate replaces the winner (:373), so the walk can
                // stop once that maximum is reached: the rest of the chain changes nothing.
                const int ml_max = (lim + 7) & ~7;
                auto in_window = [&](int q) { return q > 0 && si - q < kWinSize; };
                // `l`: lane j's link of cur's entry, fetched one batch ahead
                uint32_t l = 0;
                if (in_window(next) && lane < kL) l = ridden ? pre.l : chain_table[(size_t)(next & kWinMask) * kHcLinks + lane];
                bool first = true;
                uint64_t gate_w = 0;
                while (tries > 0 && in_window(next) && m_len < ml_max) {
#ifdef LZ4HIP_HC_PROF
                    nbatch++;
#endif
                    // one batch = cur itself (lane 0) and cur's links 0..62 (lanes 1..63);
                    // link 63 starts the next batch
                    const int cur = next;
                    const bool first_batch_mark = first;
                    if (first) HC_V(10);
                    // lane k: link k-1 (DPP wave_shr:1; lane 0 gets 0 and does not use it)
                    const uint32_t lk = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)l, 0x138, 0xF, 0xF, false);
                    const int mine = lane == 0 ? cur : cur - (int)lk;
                    const bool valid = lane == 0 || (lane < kL && lk != 0 && si - mine < kWinSize);
                    if (first) {
                        // chainTable[si] = hashTable[h] (:382): si's entry is the distances to
                        // exactly these candidates -- no second look at the table needed
                        my_link = hc_link((uint32_t)(si - cur), lk, lane);
                        first = false;
                    }
                    // links are ordered and prefix-complete: the valid lanes are 0..nv-1
                    const int nv = __builtin_popcountll(ballot(valid));
                    int cnt = nv;
                    if ((uint32_t)cnt > tries) cnt = (int)tries;
                    tries -= (uint32_t)cnt;
                    next = 0;
                    if (nv == kL && tries > 0) {
                        const uint32_t l63 = (uint32_t)__builtin_amdgcn_readlane((int)l, kL - 1);
                        if (l63 != 0) next = cur - (int)l63;  // window check: loop condition
                    }
                    // the next entry's fetch flies while this batch is compared
                    // (straight into l, whose old value is used up: through a second variable the
                    // compiler copies it at once and waits for the fetch right here)
                    l = 0;
                    if (in_window(next) && lane < kL) l = chain_table[(size_t)(next & kWinMask) * kHcLinks + lane];
                    int ml = 0;
                    if (lane < cnt) {
                        // The reference's gate (:358-360) as a shortcut once a long match is
                        // held: a candidate replaces the winner only if it is strictly longer,
                        // so it must agree with si at index m_len -- and a chain's later
                        // members mostly do not.  One load decides that instead of a compare
                        // over m_len bytes (several dependent round trips the wave waits for).
                        const bool gate = m_len < LZ4HIP_HC_GATE || ld64(s + mine + m_len - 7) == gate_w;
                        if (gate) ml = hc_prefix_len(s, mine, si, lim, w);
                    }
                    if (first_batch_mark) HC_V(11);
                    // winner: max ml, ties to the earliest chain entry (lowest lane); only a strictly
                    // longer candidate replaces the held one (:373-379), and on a long walk most
                    // batches have none: one ballot says so and the reduction is skipped
                    if (ballot(ml >= kMinMatch && ml > m_len)) {
                        // (two steps: lengths reach 2^30 in a single-block call, which leaves no
                        // room for the lane in one 32-bit key)
                        m_len = wave_max(ml);
                        offset = si - __builtin_amdgcn_readlane(mine, __builtin_ctzll(ballot(ml == m_len)));
                        // si's side of the gate: it changes with the held length only
                        if (m_len >= LZ4HIP_HC_GATE) gate_w = ld64(s + si + m_len - 7);
                    }
#if LZ4HIP_HC_SCAN > 0
                    // ---- A dense chain (runs of zero words: every earlier one a candidate, 26 k of
                    // them on average where kernel-image data has them) costs one dependent fetch per
                    // 64 members however it is walked.  Where every position of the window is known
                    // to be in the tables, the members of h's chain are simply the window's
                    // positions whose 4 bytes hash to h, in descending order -- so the rest of the
                    // walk is done as a SCAN of the source below `next`: 512 positions per step,
                    // 8 consecutive ones per lane from one 16-byte load, no dependent fetch at all.
                    // Same candidates in the same order (`tries` cannot run out: it exceeds the
                    // window), same winner: longest, ties to the nearest.
                    const bool jumped = si - last_skip < kWinSize;  // then the position map says which
                    if (cur - next <= LZ4HIP_HC_SCAN * kL / kWave && in_window(next) && tries >= (uint32_t)kWinSize &&
                        si - last_mixed >= kWinSize && (!jumped || si < (1 << 23)) && m_len < ml_max) {
                        const uint8_t *map = hc_map(chain_table);
                        const uint64_t r = hc_scan(s, map, n, si, next, w, lim, ml_max, m_len, offset, jumped, lane);
                        m_len = (int)(uint32_t)r;
                        offset = (int)(uint32_t)(r >> 32);
                        next = 0;  // the walk is over
                    }
#endif
                }
            }
#ifdef LZ4HIP_HC_PROF
            {
                const int kind = nbatch == 0 ? 0 : nbatch == 1 ? 1 : nbatch <= 8 ? 2 : 3;
                HC_T(kind);
                HC_N(8 + kind, 1);
                HC_N(12, nbatch);
            }
#endif
            // hc_insert(si): the entry was derived above (all zero when there was no head
            // inside the window); the head moves to si
            put_literals();  // behind the search's own loads
            if (lane < kL) chain_table[(size_t)(si & kWinMask) * kHcLinks + lane] = (uint16_t)my_link;
            if (lane == 0) hash_table[h] = (uint32_t)si;
            HC_U(0);
            __syncthreads();
            HC_U(8);
            HC_M(12, 1);

            if (m_len == 0) {  // :386-389
                if (si - anchor >= (1 << kAdaptSkipLog)) {
                    last_skip = si + ((si - anchor) >> kAdaptSkipLog);
                    hc_mark_jumped(hc_map(chain_table), si + 1, (si - anchor) >> kAdaptSkipLog, lane);
                }
                si += 1 + ((si - anchor) >> kAdaptSkipLog);
                miss_run++;
                continue;
            }
            miss_run = 0;

            {  // insert every position covered by the match: :394-405
                int p = si + 1;
                const int wsn = si + m_len - kWinSize;
                if (wsn > p) p = wsn;
                const int end = si + m_len;
                pre.pos = -1;
                if (wsn <= si + 1) {
                    const int cntp = end - p;
                    if (end < sn) pre.pos = end;  // the search behind this match may ride along
                    if (cntp >= kWave) {
                        // a long match: the pending ones, then the run itself, now
                        HcPre none;
                        none.pos = -1;
                        none.ok = false;
                        none.w = 0;
                        none.h = none.l = 0;
                        none.head = 0;
                        if (pd_n)
                            hc_insert_round(hash_table, chain_table, s, pd_pos, pd_h, pd_n, lane, rows, none, false);
                        pd_n = 0;
                        hc_insert_run(hash_table, chain_table, s, p, cntp, lane, rows, pre);
                    } else {
                        if (pd_n + cntp > kWave) {
                            // no room: the pending ones go in now, with the next search riding along
                            if (pre.pos >= 0) pre.w = ld64(s + pre.pos);
                            HC_U(2);
                            hc_insert_round(hash_table, chain_table, s, pd_pos, pd_h, pd_n, lane, rows, pre,
                                            pre.pos >= 0);
                            pd_n = 0;
                            HC_U(1);
                            HC_M(15, 1);
                        } else {
                            pre.pos = -1;
                        }
                        {
                            const bool mine_new = lane >= pd_n && lane < pd_n + cntp;
                            const int np = p + lane - pd_n;
                            const int wo = np - wb;  // from the window where it reaches
                            const bool cov = wo >= 0 && wo < kWave;
                            const uint32_t v = (uint32_t)__shfl((int)(uint32_t)W, cov ? wo : 0);
                            if (mine_new) {
                                pd_pos = np;
                                pd_h = hash4(cov ? v : ld32(s + np));
                            }
                        }
                        pd_n += cntp;
                        // a pending position with the next search's hash: that search will flush
                        // first, and what rode along was read before the flush
                        if (pre.pos >= 0 && ballot(lane < pd_n && pd_h == pre.h)) pre.ok = false;
                    }
                    if (pre.pos >= 0 && pre.h == h) pre.ok = false;  // si itself is then the head
                } else {
                    if (pd_n) {
                        HcPre none;
                        none.pos = -1;
                        none.ok = false;
                        none.w = 0;
                        none.h = none.l = 0;
                        none.head = 0;
                        hc_insert_round(hash_table, chain_table, s, pd_pos, pd_h, pd_n, lane, rows, none, false);
                        pd_n = 0;
                    }
                    // The reference rolls the 4-byte value from the one at si
                    // (:399-400).  When the window clamp skips positions (match
                    // longer than 64 KiB) the first three values mix old and new
                    // bytes; reproduce that, it decides which slots are written.
                    last_mixed = end - 1;
                    for (; p < end; p++) {
                        match = (match >> 8) | ((uint32_t)s[p + 3] << 24);
                        hc_insert(hash_table, chain_table, hash4(match), p, lane);
                    }
                }
                HC_U(2);
                __syncthreads();
                HC_U(3);
            }
            if (si + m_len < sn) {
                // the search behind the match: its bytes from the window, which then moves on
                // (not waited for: the match after this one is its first reader)
                const int nsi = si + m_len, wo = nsi - wb;
                if (wo >= 0 && wo < kWave) {
                    w_nx = rl64(W, wo);
                    w_nx_pos = nsi;
                    if (wo >= 24) win_fetch(nsi);
                } else {
                    win_fetch(nsi);
                }
            }

            HC_T(5);
            HC_N(13, 1);
            HC_M(13, 1);
            const int l_len = si - anchor;
            si += m_len;
            m_len -= kMinMatch;

            // ---- emit, block.go:411-449; an index panic is "failed" ----
            if (l_len < 0xF && m_len < 0xF && di + l_len + 3 <= dlen) {
                // the common sequence -- token, up to 14 literals, offset -- and room for all of
                // it: the three bytes around the literals go out in one store (lanes 0..2)
                const int at = lane == 0 ? di : di + l_len + lane;
                const uint32_t val = lane == 0 ? (uint32_t)(m_len | (l_len << 4)) : (uint32_t)offset >> (8 * (lane - 1) & 31);
                if (lane < 3) d[at] = (uint8_t)val;
                di++;
                if (lane < l_len) lit = s[anchor + lane];
                lit_n = l_len;
                lit_di = di;
                di += l_len + 2;
                anchor = si;
                HC_T(6);
                HC_U(4);
                continue;
            }
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
            if (l_len <= kWave) {
                if (lane < l_len) lit = s[anchor + lane];
                lit_n = l_len;
                lit_di = di;
            } else {
                wave_copy(d + di, s + anchor, l_len, lane);
            }
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
            HC_T(6);
            HC_U(4);
        }

        put_literals();
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
#ifdef LZ4HIP_HC_PROF
        HC_T(7);
        if (lane == 0)
            for (int i = 0; i < 16; i++) atomicAdd(&g_hc_prof[i], prof[i]);
#endif
    }
}

#ifdef LZ4HIP_HC_PROF
extern "C" int lz4hip_debug_hc_prof(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_hc_prof), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -1;
}
#endif

hipError_t launch_hc(const uint8_t *src, size_t n, uint32_t bs, size_t nb, uint32_t depth,
                     int dst_mode,
                     uint32_t dst_len_x, uint8_t *dst, size_t dst_stride, int32_t *result, uint32_t *ws,
                     size_t slots, uint32_t *next_block, hipStream_t st) {
    if (nb == 0) return hipSuccess;
    const size_t grid = nb < slots ? nb : slots;
    if (grid == nb || nb > 0xFFFFFFFFull) next_block = nullptr;  // every block has a wave of its own
    if (next_block) {
        const hipError_t e = hipMemsetAsync(next_block, 0, sizeof(uint32_t), st);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_hc_compress, dim3((unsigned)grid), dim3(kWave), 0, st, src, n, bs, depth,
                       dst_mode, dst_len_x, dst, dst_stride, result, ws, nb, next_block);
    return hipGetLastError();
}

}  // namespace lz4hip
