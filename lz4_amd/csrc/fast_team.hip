// fast_team.hip -- Fast LZ4 compressor for launches with FEW blocks: a team of four
// wavefronts per block instead of one.
//
// Same parse and bytes as (*Compressor).CompressBlock (internal/lz4block/block.go:105-295)
// and as fast.hip's one-wave kernels.  The regime is the reference's most common one:
// lz4.CompressBlock on one buffer (lz4.go:92-94), a Writer with its default options
// (Block4Mb, Concurrency 1: options.go:29-34, writer.go:123-128) or a small
// ConcurrencyOption(n) -- fewer blocks than the chip has CUs.  A one-wave kernel then runs
// phase A (hash, table gather, candidate verify: two dependent memory round trips), phase B
// (the serial walk) and phase C (encoding) one after the other on a single instruction
// stream, and a lone wave issues one instruction every ~4 cycles.  Here the phases run on
// different waves of one workgroup (one per SIMD) and overlap:
//
//   waves 2, 3  PRODUCERS.  Windows are fixed: window w covers positions [60w-2, 60w+62).
//               A producer takes every other window, hashes its 64 positions, gathers the
//               table (in LDS), verifies the candidates and pre-extends them exactly as
//               fast.hip's phase A does, and publishes the per-lane answers -- together with
//               the table values they were derived from -- in an LDS ring.  Producers run up
//               to kRingA windows ahead of the walk, so the table they read is STALE by the
//               puts of those windows.
//   wave 0      WALKER.  For the window that contains si it re-reads the table entries of
//               the 64 positions (LDS, ~100 cycles).  A lane whose entry still equals the one
//               the producer saw has an exact answer (a candidate is a function of the entry
//               and the position only); the others are re-evaluated on the spot, their new
//               candidate being at most kRingA windows old and therefore still in the LDS
//               source ring.  Then the reference's steps are replayed as in fast.hip's phase B
//               (get/put order T3, adaptive skip, in-window put -> get forwarding), the window's
//               puts go into the table, and the matches taken are handed on.
//   wave 1      ENCODER.  fast.hip's phase C per window (lengths, wave scan of the sequence
//               sizes, dst-capacity verdict T5, vector emission), then the last literals.
//
// Because windows are fixed (not anchored at si as in fast.hip), no wave ever needs another
// wave's result to know WHERE to work; only the walker's table and the order of the rings
// carry dependencies.  A step needs positions si, si+1, si+2 inside its window and the put of
// si-2 that follows a match must lie in the window that contains si: with stride 60 and 64
// lanes both hold (a = si - base is in [2, 61]).
#include "fast_common.h"

namespace lz4hip {

constexpr int kTeamStride = 60;
#ifndef LZ4HIP_TEAM_PRODUCERS
#define LZ4HIP_TEAM_PRODUCERS 4
#endif
#ifndef LZ4HIP_TEAM_LEAD
#define LZ4HIP_TEAM_LEAD 3
#endif
constexpr int kProducers = LZ4HIP_TEAM_PRODUCERS;   // producer p takes the windows w = p (mod kProducers)
constexpr int kTeamThreads = (2 + kProducers) * kWave;
// producer -> walker slots: every producer has a ring of its own (kRingP slots), so that a slot's
// sequence word is only ever written by one wave, in window order -- a producer that is still
// busy with a window the walk has jumped past can then not overwrite a newer window's word
constexpr int kRingP = 8 / kProducers;
constexpr int kRingA = kRingP * kProducers;
constexpr int kLead = LZ4HIP_TEAM_LEAD;  // windows the producers may run ahead of the walk
static_assert(kLead <= kRingA && kLead >= 1 && kRingP >= 1, "a window needs a slot");
__device__ __forceinline__ int a_slot(int w) {
    const uint32_t u = (uint32_t)w;
    return (int)((u % kProducers) * kRingP + (u / kProducers) % kRingP);
}
constexpr int kRingE = 8;          // walker -> encoder slots
constexpr int kSrcRing = 8192;     // LDS source ring; 16 more bytes mirror its start
constexpr int kFillChunk = 1024;   // one coalesced 16-byte-per-lane load
constexpr int kTeamReach = 84;     // a window reads s[base-8, base+84)
constexpr int kTeamOut = 512;      // encoder's staging buffer

struct TeamShared {
    uint32_t a_seq[kRingA];   // slot holds window a_seq-1
    uint32_t e_seq[kRingE];   // slot holds record e_seq-1
    uint32_t w_need;          // windows below this one are dead (the walker is past them)
    uint32_t e_done;          // records the encoder has taken
    uint32_t done;            // the walker has finished the block
    uint32_t abort;           // the encoder ran out of dst
    int32_t fill_lo, fill_hi; // the ring holds s[max(fill_lo, fill_hi - 7168), fill_hi)
    uint32_t e_hdr[kRingE][8];  // a record's scalars: FOUND lo, hi, window base, entry anchor, last
    uint32_t a_rec[kRingA][3][kWave];
    uint32_t e_rec[kRingE][3][kWave];
};

// (every lane reads the same word: the value is wave-uniform, and saying so keeps the waves'
// control flow on the scalar unit)
__device__ __forceinline__ uint32_t lds_ld(const uint32_t *p) {
    return rfl(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}
__device__ __forceinline__ void lds_st(uint32_t *p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// LDS operations of one wave are performed in issue order; the wait makes the data stores
// of a record complete before its sequence word is stored.
#define LDS_DRAIN() __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// Last resort against a wave that would wait for ever (a kernel that never ends can take the
// whole node down): a wait that has polled kSpinLimit times -- tens of seconds; every
// legitimate wait is bounded by the other waves' work on one block -- raises abort, every
// wave leaves, and the block's result is -1.
constexpr uint32_t kSpinLimit = 1u << 28;
#define TEAM_SPIN(cnt)                                    \
    do {                                                  \
        if (++(cnt) > kSpinLimit) lds_st(&S->abort, 2u); \
    } while (0)

__device__ __forceinline__ uint64_t ring64(const uint8_t *ring, int q) {
    uint64_t v;
    __builtin_memcpy(&v, ring + (q & (kSrcRing - 1)), 8);
    return v;
}
__device__ __forceinline__ uint32_t ring32(const uint8_t *ring, int q) {
    uint32_t v;
    __builtin_memcpy(&v, ring + (q & (kSrcRing - 1)), 4);
    return v;
}

// What phase A derives for one position from its table entry: block.go:74-86 (get), the
// 4-byte verify of :141-163, 16 bytes of forward extension under the si+8<=sn stride rule of
// :184-193 and 8 of backward extension (:171-177, clamped by the anchor later).  c32 / c64
// read candidate bytes, s64 the position's own.
struct LaneAns {
    int offv, mendv, backv;
    bool m4, fmore, bmore;
};
template <class C32, class C64, class S64>
__device__ __forceinline__ LaneAns eval_candidate(uint32_t tv, int p, bool valid, uint32_t xlo, int sn,
                                                  C32 &&c32, C64 &&c64, S64 &&s64) {
    LaneAns A;
    int ref = (int)tv + (p & ~kWinMask);
    if (ref >= p) ref -= kWinSize;
    A.offv = p - ref;
    const bool offok = valid && A.offv < kWinSize && ref >= 0;
    const bool f0ok = offok && p + 12 <= sn;
    const bool f1ok = offok && p + 20 <= sn;
    const bool bok = offok && ref >= 8;
    uint32_t cw = 0;
    uint64_t r0 = 0, q0 = 0, r1 = 0, q1 = 0, rb = 0, qb = 0;
    if (offok) cw = c32(ref);
    if (f0ok) {
        r0 = c64(ref + 4);
        q0 = s64(p + 4);
    }
    if (f1ok) {
        r1 = c64(ref + 12);
        q1 = s64(p + 12);
    }
    if (bok) {
        rb = c64(ref - 8);
        qb = s64(p - 8);
    }
    A.m4 = offok && cw == xlo;
    A.mendv = p + kMinMatch;
    A.fmore = false;
    if (A.m4 && f0ok) {
        uint64_t xx = r0 ^ q0;
        if (xx) {
            A.mendv += __builtin_ctzll(xx) >> 3;
        } else {
            A.mendv += 8;
            if (f1ok) {
                xx = r1 ^ q1;
                if (xx) {
                    A.mendv += __builtin_ctzll(xx) >> 3;
                } else {
                    A.mendv += 8;
                    A.fmore = true;
                }
            }
        }
    }
    A.backv = 0;
    A.bmore = false;
    if (A.m4) {
        if (bok) {
            const uint64_t xx = rb ^ qb;
            if (xx) {
                A.backv = __builtin_clzll(xx) >> 3;
            } else {
                A.backv = 8;
                A.bmore = true;
            }
        } else {
            A.bmore = true;
        }
    }
    if (!offok) A.offv = 0;
    return A;
}

// -DLZ4HIP_TEAM_PROF (profiling builds, tools/prof): shader-clock cycles per role and phase,
// summed over the blocks.  walker: [0] waiting for a window, [1] records + stale check +
// re-evaluation, [2] walk, [3] puts + hand-over, [4] windows, [5] windows with stale lanes,
// [6] records; producers: [8] waiting (ring full / nothing to do), [9] source fill or waiting
// for it, [10] phase A, [11] windows; encoder: [12] waiting, [13] encoding, [14] records.
#ifdef LZ4HIP_TEAM_PROF
__device__ unsigned long long g_team_prof[16];
#define TP_DECL unsigned long long tp_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tp_last = __builtin_readcyclecounter()
#define TP_T(i)                                                        \
    do {                                                               \
        const unsigned long long now_ = __builtin_readcyclecounter(); \
        tp_[i] += now_ - tp_last;                                      \
        tp_last = now_;                                                \
    } while (0)
#define TP_N(i, v) tp_[i] += (unsigned long long)(v)
#define TP_FLUSH()                                                      \
    do {                                                                \
        if (lane == 0)                                                  \
            for (int i_ = 0; i_ < 16; i_++)                             \
                if (tp_[i_]) atomicAdd(&g_team_prof[i_], tp_[i_]);      \
    } while (0)
#else
#define TP_DECL
#define TP_T(i)
#define TP_N(i, v)
#define TP_FLUSH()
#endif

// ---------------------------------------------------------------------------------------
// Producer: waves 2 and 3 (j = 0, 1), windows w = j, j+2, ...
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void team_producer(const uint8_t *__restrict__ s, const int n,
                                              const uint16_t *table, TeamShared *S, uint8_t *ring,
                                              uint32_t *tag, const int j, const int lane) {
    const int sn = n - kMfLimit;
    const int last_w = sn > 0 ? (sn - 1) / kTeamStride : -1;  // the walker needs window si/60, si < sn
    int w = j;
    int lo = 0, hi = 0;  // producer 0 owns the ring's fill state
    uint32_t spins = 0;
    TP_DECL;
    // Producer 0 keeps the source ring ahead of everyone: s[.., base(need) + kRingA windows +
    // reach + one chunk), also while it has no window of its own to produce.
    auto fill_duty = [&](const int need) {
        const int nb0 = kTeamStride * need - 2;
        if (nb0 - 16 > hi) {  // the walk jumped over everything buffered: restart there
            lo = (nb0 - 16) & ~15;
            hi = lo;
            lds_st((uint32_t *)&S->fill_hi, (uint32_t)hi);
            LDS_DRAIN();
            lds_st((uint32_t *)&S->fill_lo, (uint32_t)lo);
        }
        int want = nb0 + kTeamStride * kLead + kTeamReach + kFillChunk;
        if (want > n) want = n;
        while (hi < want) {
            const int q = hi + lane * 16;
            if (q < n) {
                uint8_t *to = ring + (q & (kSrcRing - 1));
                if (q + 16 <= n) {
                    const uint64_t a = ld64(s + q), b = ld64(s + q + 8);
                    *(uint64_t *)to = a;
                    *(uint64_t *)(to + 8) = b;
                    if ((q & (kSrcRing - 1)) == 0) {
                        *(uint64_t *)(ring + kSrcRing) = a;
                        *(uint64_t *)(ring + kSrcRing + 8) = b;
                    }
                } else {
                    for (int t = 0; q + t < n; t++) {
                        to[t] = s[q + t];
                        if ((q & (kSrcRing - 1)) == 0) ring[kSrcRing + t] = s[q + t];
                    }
                }
            }
            hi = hi + kFillChunk < n ? hi + kFillChunk : n;
            LDS_DRAIN();
            lds_st((uint32_t *)&S->fill_hi, (uint32_t)hi);
        }
    };
    for (;;) {
        uint32_t need;
        for (;;) {
            if (lds_ld(&S->done) | lds_ld(&S->abort)) {
                TP_T(8);
                TP_FLUSH();
                return;
            }
            need = lds_ld(&S->w_need);
            TP_T(8);
            if (j == 0) fill_duty((int)need);
            TP_T(9);
            if (w < (int)need)  // dead windows: catch up to the first own window >= need
                w += ((int)need - w + kProducers - 1) / kProducers * kProducers;
            if (w <= last_w && w < (int)need + kLead) break;
            __builtin_amdgcn_s_sleep(2);
            TEAM_SPIN(spins);
        }
        const int wbase = kTeamStride * w - 2;
        if (j != 0) {
            const int rd_lo = wbase - 8 < 0 ? 0 : wbase - 8;
            const int rd_hi = wbase + kTeamReach < n ? wbase + kTeamReach : n;
            bool dead = false;
            for (;;) {
                const int fl = (int)lds_ld((uint32_t *)&S->fill_lo);
                const int fh = (int)lds_ld((uint32_t *)&S->fill_hi);
                if (fl <= rd_lo && fh >= rd_hi) break;
                if (lds_ld(&S->done) | lds_ld(&S->abort)) {
                    TP_FLUSH();
                    return;
                }
                // the walk may have jumped past this window meanwhile; the ring then restarts
                // behind it and these bytes never come
                if ((int)lds_ld(&S->w_need) > w) {
                    dead = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
                TEAM_SPIN(spins);
            }
            TP_T(9);
            if (dead) continue;
        }
        __asm__ volatile("" ::: "memory");
        // ---- phase A of window w (fast.hip:185-311) ----
        const int p = wbase + lane;
        const bool valid = p >= 0 && p <= sn + 1;
        uint64_t x = 0;
        if (valid) x = ring64(ring, p);
        const uint32_t h = hash6(x);
        uint32_t tv = 0;
        if (valid) tv = __hip_atomic_load(table + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint32_t slot = h & (kTagN - 1);
        if (valid) atomicMin(&tag[slot], (uint32_t)lane);
        __asm__ volatile("" ::: "memory");
        uint32_t tmin = 0xFFFFFFFFu;
        if (valid) tmin = tag[slot];
        __asm__ volatile("" ::: "memory");
        if (valid) tag[slot] = 0xFFFFFFFFu;
        bool cflag = valid && tmin < (uint32_t)lane;  // an earlier lane shares my tag slot ...
        {   // ... and, looked at exactly, my hash: the walker leaves its fast path for these lanes
            // (one round per distinct hash among the flagged lanes: a run of equal bytes is one)
            uint64_t cm = ballot(cflag);
            cflag = false;
            while (cm) {
                const int b = __builtin_ctzll(cm);
                const uint64_t E = ballot(valid && h == rdlane(h, b));  // every lane with b's hash
                if (((E >> lane) & 1) && (E & ((1ull << lane) - 1)) != 0) cflag = true;  // all but the first of them
                cm &= ~E;
            }
        }
        const LaneAns A = eval_candidate(
            tv, p, valid, (uint32_t)x, sn, [&](int q) { return ld32(s + q); },
            [&](int q) { return ld64(s + q); }, [&](int q) { return ring64(ring, q); });
        const int sl = a_slot(w);
        S->a_rec[sl][0][lane] = tv | (h << 16);
        S->a_rec[sl][1][lane] = (uint32_t)A.mendv;
        S->a_rec[sl][2][lane] = (uint32_t)A.offv | ((uint32_t)A.backv << 16) | ((uint32_t)A.m4 << 20) |
                                ((uint32_t)A.fmore << 21) | ((uint32_t)A.bmore << 22) |
                                ((uint32_t)cflag << 23) | ((uint32_t)valid << 24);
        LDS_DRAIN();
        if (lane == 0) lds_st(&S->a_seq[sl], (uint32_t)w + 1);
        w += kProducers;
        TP_T(10);
        TP_N(11, 1);
    }
}

// ---------------------------------------------------------------------------------------
// Walker: wave 0.  fast.hip's phase B over the producers' answers.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void team_walker(const uint8_t *__restrict__ s, const int n, uint16_t *table,
                                            TeamShared *S, const uint8_t *ring, const int lane) {
    const int sn = n - kMfLimit;
    int si = 0, anchor = 0;
    bool pending = false, finished = false, aborted = false;
    uint32_t k = 0;  // records handed to the encoder
    uint32_t spins = 0;
    TP_DECL;
    uint32_t e_done_seen = 0;   // the encoder's count as last read: re-read only when the ring looks full
    int pf_w = -1;              // the window whose records were read ahead
    uint32_t pf_seq = 0, pf_r0 = 0, pf_r1 = 0, pf_r2 = 0, pf_x = 0;
    auto e_slot_wait = [&]() {  // room for record k
        while (k >= e_done_seen + kRingE) {
            e_done_seen = lds_ld(&S->e_done);
            if (k < e_done_seen + kRingE) break;
            if (lds_ld(&S->abort)) {
                aborted = true;
                break;
            }
            TEAM_SPIN(spins);
        }
    };
    while (si < sn && !finished && !aborted) {
        si = (int)rfl((uint32_t)si);
        anchor = (int)rfl((uint32_t)anchor);
        const int w = (int)((uint32_t)si / kTeamStride);
        const int wbase = kTeamStride * w - 2;
        if (lane == 0) lds_st(&S->w_need, (uint32_t)w);
        const int sl = a_slot(w);
        const int p = wbase + lane;
        // The sequence word, the window's records and the position's own bytes are read in one
        // batch (LDS reads of a wave are performed in order: records read behind a sequence
        // word that says "window w" are window w's); they are read again only if the producer
        // was not there yet.
        uint32_t r0, r1, r2, xlo;
        // (asked for while the window before was walked, if its producer was done by then)
        const bool ahead = pf_w == w && rfl(pf_seq) == (uint32_t)w + 1;
        r0 = pf_r0, r1 = pf_r1, r2 = pf_r2, xlo = pf_x;
        for (; !ahead;) {
            const uint32_t seq = __hip_atomic_load(&S->a_seq[sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __asm__ volatile("" ::: "memory");
            r0 = S->a_rec[sl][0][lane], r1 = S->a_rec[sl][1][lane], r2 = S->a_rec[sl][2][lane];
            xlo = ring32(ring, p);
            __asm__ volatile("" ::: "memory");
            if (rfl(seq) == (uint32_t)w + 1) break;
            if (lds_ld(&S->abort)) {
                aborted = true;
                break;
            }
            TEAM_SPIN(spins);
        }
        if (aborted) break;
        TP_T(0);
        TP_N(4, 1);
        const uint32_t h = r0 >> 16;
        const bool valid = (r2 >> 24) & 1;
        const bool cflag = (r2 >> 23) & 1;
        int offv = (int)(r2 & 0xFFFFu), backv = (int)((r2 >> 16) & 0xFu), mendv = (int)r1;
        bool m4 = (r2 >> 20) & 1, fmore = (r2 >> 21) & 1, bmore = (r2 >> 22) & 1;
        const uint32_t tv_now = table[h];  // (h of a lane that is not valid is some entry's index too)
        // Lanes below the walk's entry point are never probed in this window (their positions lie
        // behind si; some were put by the window before, the 4 positions the two share) and their
        // answers do not matter.
        const bool stale = valid && lane >= si - wbase && tv_now != (r0 & 0xFFFFu);
        if (ballot(stale)) {
            TP_N(5, 1);
            // The entry changed after the producer read it: by a put of one of the last kRingA
            // windows, so the new candidate is in the source ring (else: global memory).
            const int fh = (int)lds_ld((uint32_t *)&S->fill_hi);
            int fl = (int)lds_ld((uint32_t *)&S->fill_lo);
            if (fl < fh - (kSrcRing - kFillChunk)) fl = fh - (kSrcRing - kFillChunk);
            // (the new candidate's bytes are ref-8 .. ref+20)
            int ref = (int)tv_now + (p & ~kWinMask);
            if (ref >= p) ref -= kWinSize;
            const bool in_ring = !stale || (ref - 8 >= fl && ref + 20 <= fh);
            if (ballot(!in_ring) == 0) {
                if (stale) {
                    const LaneAns A = eval_candidate(
                        tv_now, p, true, xlo, sn, [&](int q) { return ring32(ring, q); },
                        [&](int q) { return ring64(ring, q); }, [&](int q) { return ring64(ring, q); });
                    offv = A.offv, backv = A.backv, mendv = A.mendv;
                    m4 = A.m4, fmore = A.fmore, bmore = A.bmore;
                }
            } else if (stale) {
                const LaneAns A = eval_candidate(
                    tv_now, p, true, xlo, sn, [&](int q) { return ld32(s + q); },
                    [&](int q) { return ld64(s + q); }, [&](int q) { return ring64(ring, q); });
                offv = A.offv, backv = A.backv, mendv = A.mendv;
                m4 = A.m4, fmore = A.fmore, bmore = A.bmore;
            }
        }
        const uint64_t M = ballot(m4), CM = ballot(cflag);
        const uint64_t RARE = CM | ballot(fmore || bmore);  // probes that need more than a lookup
        const uint64_t ATT = M | CM;
        const uint16_t pos16 = (uint16_t)p;
        const uint32_t flv = (uint32_t)fmore | ((uint32_t)bmore << 1);
        // The window's record for the encoder is written NOW (slot k is the walker's as soon as
        // the encoder has taken record k - kRingE); the rare path below overwrites the entries of
        // its match lane.  The walk itself then carries no per-lane state at all: a match's
        // anchor is the end of the match before it, which the encoder reads off the record.
        e_slot_wait();
        if (aborted) break;
        const int es = (int)(k & (kRingE - 1));
        S->e_rec[es][0][lane] = (uint32_t)mendv;
        S->e_rec[es][1][lane] = (uint32_t)offv;
        S->e_rec[es][2][lane] = (uint32_t)backv;
        const int anchor_in = anchor;
        uint64_t FOUND = 0;
        uint64_t P = pending ? 1ull << (si - 2 - wbase) : 0ull;  // put(si-2) of the match that ended the last window
        pending = false;

        // The next window's sequence word and records, asked for now: the reads complete under
        // the walk, and if the sequence word already says "window w + 1" they are that window's
        // (a slot is rewritten only several windows on), so the next turn starts without the wait.
        {
            const int sn1 = a_slot(w + 1);
            pf_w = w + 1;
            pf_seq = __hip_atomic_load(&S->a_seq[sn1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __asm__ volatile("" ::: "memory");
            pf_r0 = S->a_rec[sn1][0][lane], pf_r1 = S->a_rec[sn1][1][lane], pf_r2 = S->a_rec[sn1][2][lane];
            pf_x = ring32(ring, wbase + kTeamStride + lane);
            __asm__ volatile("" ::: "memory");
        }
        TP_T(1);
        // ================= Phase B: fast_common.h for the common steps, the rest here =========
        for (;;) {
            si = (int)rfl((uint32_t)si);
            anchor = (int)rfl((uint32_t)anchor);
            const int st = walk_common_steps(si, anchor, P, FOUND, wbase, sn, ATT, RARE, mendv);
            if (st == 0) break;
            if (st == 2) {  // block.go:250-252
                finished = true;
                break;
            }
            // ---- a probe of this step may depend on a put of this window, or its match needs
            // cooperative extension: resolve in the reference's order (fast.hip) ----
            TP_N(7, 1);
            const int a = si - wbase;
            int jj = -1, dep_l = -1;
#pragma unroll
            for (int t = 0; t < 3 && jj < 0; ++t) {
                const int b = a + t;
                // probes 0,1 see the puts before this step; probe 2 also put(si), put(si+1)
                const uint64_t vis = t == 2 ? (P | (3ull << a)) : P;
                bool hit = (M >> b) & 1;
                int dl = -1;
                if ((CM >> b) & 1) {
                    const uint64_t E = ballot(valid && h == rdlane(h, b)) & vis;
                    if (E) {
                        dl = 63 - __builtin_clzll(E);  // the latest put with my hash
                        hit = rdlane(xlo, b) == rdlane(xlo, dl);
                    }
                }
                if (hit) {
                    jj = t;
                    dep_l = dl;
                }
            }
            if (jj < 0) {
                P |= 7ull << a;
                si += 4 + ((si + 2 - anchor) >> kAdaptSkipLog);
                continue;
            }
            const int f = a + jj;
            P |= (jj == 2 ? 7ull : 3ull) << a;
            const int found = si + jj;
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
                const uint32_t flg = rdlane(flv, f);
                if ((flg & 2u) && maxb > bw) back = ext_backward(s, found, mref, bw, maxb, lane);
                if (flg & 1u) mend = ext_forward(s, mend, offset, sn, lane);
            }
            if (lane == f) {
                S->e_rec[es][0][lane] = (uint32_t)mend;
                S->e_rec[es][1][lane] = (uint32_t)offset;
                S->e_rec[es][2][lane] = (uint32_t)back;
            }
            si = mend;
            FOUND |= 1ull << f;
            anchor = si;
            if (si >= sn) {
                finished = true;
                break;
            }
            const int q = si - 2 - wbase;  // put(hash(si-2), si-2): block.go:254-255
            if (q <= 63) P |= 1ull << q;
        }
        // a match that ended the window leaves its put(si-2) to the next one when it lies beyond lane 63
        pending = !finished && anchor == si && si - 2 - wbase > 63;

        TP_T(2);
        // ---- the window's puts (fast.hip:497-511) ----
        if (!finished) {
            if ((P & ~CM) >> lane & 1) table[h] = pos16;
            uint64_t Pd = P & CM;
            while (Pd) {
                __asm__ volatile("" ::: "memory");
                const int l = __builtin_ctzll(Pd);
                Pd &= Pd - 1;
                if (lane == l) table[h] = pos16;
            }
            __asm__ volatile("" ::: "memory");
        }
        // ---- the window's matches go to the encoder ----
        if (FOUND) {
            {   // the record's scalars, one word per lane, in one store
                const uint32_t v = lane == 0 ? (uint32_t)FOUND : lane == 1 ? (uint32_t)(FOUND >> 32)
                                 : lane == 2 ? (uint32_t)wbase : lane == 3 ? (uint32_t)anchor_in : 0u;
                if (lane < 5) S->e_hdr[es][lane] = v;
            }
            // (LDS operations of a wave are performed in order: the record is complete when
            // its sequence word is)
            __asm__ volatile("" ::: "memory");
            if (lane == 0) lds_st(&S->e_seq[es], k + 1);
            k++;
            TP_N(6, 1);
        }
        TP_T(3);
    }
    if (!aborted) {
        e_slot_wait();
        if (!aborted) {
            const int es = (int)(k & (kRingE - 1));
            if (lane == 0) {
                S->e_hdr[es][3] = (uint32_t)(int)rfl((uint32_t)anchor);
                S->e_hdr[es][4] = 1u;
            }
            LDS_DRAIN();
            if (lane == 0) lds_st(&S->e_seq[es], k + 1);
        }
    }
    LDS_DRAIN();
    if (lane == 0) lds_st(&S->done, 1u);
    TP_FLUSH();
}

// ---------------------------------------------------------------------------------------
// Encoder: wave 1.  fast.hip's phase C per record and the last literals; returns the block's
// result (n, 0 or -1: block.go:258-294).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int team_encoder(const uint8_t *__restrict__ s, const int n,
                                            uint8_t *__restrict__ d, const int dlen, TeamShared *S,
                                            uint8_t *obuf, const int lane) {
    const int bound = (int)compress_bound32((uint32_t)n);
    const bool not_compressible = dlen < bound;  // block.go:110
    int di = 0, anchor = 0;
    bool failed = false;
    uint32_t spins = 0;
    TP_DECL;
    for (uint32_t k = 0;; k++) {
        TP_T(13);
        const int es = (int)(k & (kRingE - 1));
        while (lds_ld(&S->e_seq[es]) != k + 1 && !lds_ld(&S->abort)) {
            __builtin_amdgcn_s_sleep(1);
            TEAM_SPIN(spins);
        }
        if (lds_ld(&S->abort)) {
            failed = true;
            break;
        }
        TP_T(12);
        TP_N(14, 1);
        __asm__ volatile("" ::: "memory");
        const uint32_t hdr = S->e_hdr[es][lane & 7];
        const uint32_t last = rdlane(hdr, 4);
        if (last) {
            anchor = (int)rdlane(hdr, 3);
            break;
        }
        const uint64_t FOUND = ((uint64_t)rdlane(hdr, 1) << 32) | rdlane(hdr, 0);
        const int wbase = (int)rdlane(hdr, 2);
        const int anchor_in = (int)rdlane(hdr, 3);
        const int mendv = (int)S->e_rec[es][0][lane];
        const int offv = (int)S->e_rec[es][1][lane], backv = (int)S->e_rec[es][2][lane];
        LDS_DRAIN();
        if (lane == 0) lds_st(&S->e_done, k + 1);  // the slot is free again
        const int p = wbase + lane;
        const bool mine = (FOUND >> lane) & 1;
        // a match's anchor: the end of the match before it in the window, else the anchor the
        // walk entered the window with (block.go:249: anchor = si after every match)
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
        if (di + total > dlen) {  // ErrInvalidSourceShortBuffer (block.go:196-245)
            failed = true;
            LDS_DRAIN();
            if (lane == 0) lds_st(&S->abort, 1u);
            break;
        }
        auto emit = [&](uint8_t *o) {
            *o++ = (uint8_t)((m_len < 0xF ? m_len : 0xF) | ((l_len < 0xF ? l_len : 0xF) << 4));
            if (l_len >= 0xF) o += lane_len_tail(o, l_len - 0xF);
            lane_copy_exact(o, s + anch, l_len);
            o += l_len;
            o[0] = (uint8_t)offv;
            o[1] = (uint8_t)(offv >> 8);
            if (m_len >= 0xF) lane_len_tail(o + 2, m_len - 0xF);
        };
        if (total <= kTeamOut) {
            LDS_DRAIN();
            if (mine) emit(obuf + (incl - esz));
            LDS_DRAIN();
            uint8_t *to = d + di;
            for (int t = lane * 16; t < total; t += kWave * 16) {
                const uint64_t a = *(const uint64_t *)(obuf + t), b = *(const uint64_t *)(obuf + t + 8);
                const int rem = total - t;
                if (rem >= 16) {
                    __builtin_memcpy(to + t, &a, 8);
                    __builtin_memcpy(to + t + 8, &b, 8);
                } else if (rem >= 8) {
                    __builtin_memcpy(to + t, &a, 8);
                    store_tail8(to + t + 8, b, rem - 8);
                } else {
                    store_tail8(to + t, a, rem);
                }
            }
        } else if (mine) {
            emit(d + di + (incl - esz));
        }
        di += total;
    }
    TP_FLUSH();
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
            wave_copy16(d + di, s + anchor, l_len, lane);
            ret = di + l_len;
        }
    }
    return ret;
}

__global__ __launch_bounds__(kTeamThreads) void k_fast_team(const uint8_t *__restrict__ src, size_t n_total,
                                                    uint32_t block_size, int dst_mode,
                                                    uint32_t dst_len_x, uint8_t *__restrict__ dst,
                                                    size_t dst_stride, int32_t *__restrict__ result) {
    __shared__ uint16_t table[1 << 16];
    __shared__ TeamShared S;
    __shared__ __attribute__((aligned(16))) uint8_t ring[kSrcRing + 16];
    __shared__ uint32_t tag[kProducers][kTagN];
    __shared__ __attribute__((aligned(16))) uint8_t obuf[kTeamOut + 16];
    const int lane = lane_id();
    const int wv = (int)rfl(threadIdx.x >> 6);
    const size_t b = blockIdx.x;
    const size_t start = b * (size_t)block_size;
    const size_t left = n_total > start ? n_total - start : 0;
    const int n = (int)(left < block_size ? left : block_size);
    const uint8_t *s = src + start;
    {  // reset(): an all-zero table is the cleared-bitmap state (block.go:74-94)
        typedef uint4 __attribute__((may_alias)) uint4_a;
        uint4_a *t4 = (uint4_a *)(table);
        for (int t = threadIdx.x; t < (1 << 16) * 2 / 16; t += kTeamThreads) t4[t] = make_uint4(0, 0, 0, 0);
        for (int t = threadIdx.x; t < kProducers * kTagN; t += kTeamThreads) (&tag[0][0])[t] = 0xFFFFFFFFu;
        uint32_t *z = (uint32_t *)&S;
        for (int t = threadIdx.x; t < (int)(offsetof(TeamShared, a_rec) / 4); t += kTeamThreads) z[t] = 0u;
    }
    __syncthreads();
    if (wv == 0) {
        __builtin_amdgcn_s_setprio(3);
        team_walker(s, n, table, &S, ring, lane);
    } else if (wv == 1) {
        const int r = team_encoder(s, n, dst + b * dst_stride, block_dlen2(dst_mode, n, dst_len_x), &S,
                                   obuf, lane);
        if (lane == 0) result[b] = r;
    } else {
        team_producer(s, n, table, &S, ring, tag[wv - 2], wv - 2, lane);
    }
}

#ifdef LZ4HIP_TEAM_PROF
extern "C" int lz4hip_debug_team_prof(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_team_prof), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -1;
}
#endif

hipError_t launch_fast_team(const uint8_t *src, size_t n, uint32_t bs, size_t nb, int dst_mode,
                             uint32_t dst_len_x, uint8_t *dst, size_t dst_stride, int32_t *result,
                             hipStream_t st) {
    if (nb == 0) return hipSuccess;
    hipLaunchKernelGGL(k_fast_team, dim3((unsigned)nb), dim3(kTeamThreads), 0, st, src, n, bs, dst_mode,
                       dst_len_x, dst, dst_stride, result);
    return hipGetLastError();
}

}  // namespace lz4hip
