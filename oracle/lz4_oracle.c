/*
 * lz4_oracle.c -- CPU restatement of pierrec/lz4 v4's block codec, xxh32 and
 * frame step.  TEST INFRASTRUCTURE ONLY: see lz4_oracle.h for who may call it
 * and for the pinning status ("compressed-byte parity unpinned").
 *
 * Written from the reference's Go sources as a specification; each function
 * cites the file:line it restates.  Plain C11, no dependencies but pthreads
 * (batch helpers).
 */
#include "lz4_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ---- constants: internal/lz4block/block.go:11-26 ---- */
enum {
    MIN_MATCH = 4,
    WIN_SIZE = 1 << 16,
    WIN_MASK = WIN_SIZE - 1,
    HT_SIZE = 1 << 16,
    MF_LIMIT = 10 + MIN_MATCH,
    ADAPT_SKIP_LOG = 7 /* block.go:115,335 */
};

static inline uint64_t ld64(const uint8_t *p) {
    uint64_t v;
    memcpy(&v, p, 8);
    return v; /* little-endian host assumed (x86-64) */
}
static inline uint32_t ld32(const uint8_t *p) {
    uint32_t v;
    memcpy(&v, p, 4);
    return v;
}
static inline void st32(uint8_t *p, uint32_t v) { memcpy(p, &v, 4); }

/* block.go:40-42 */
size_t lz4o_compress_bound(size_t n) { return n + n / 255 + 16; }

/* block.go:35-38: hash of the low 6 bytes, 16-bit result */
uint32_t lz4o_block_hash(uint64_t x) {
    const uint64_t prime6bytes = 227718039650203ULL;
    return (uint32_t)(((x << 16) * prime6bytes) >> 48);
}

/* block.go:298-301 */
uint32_t lz4o_block_hash_hc(uint32_t x) { return (uint32_t)(x * 2654435761u) >> 16; }

/* ---- Fast compressor state: block.go:54-94 ---- */
typedef struct {
    uint16_t table[HT_SIZE];
    uint32_t in_use[HT_SIZE / 32];
} fast_state;

static inline int64_t fast_get(const fast_state *c, uint32_t h, int64_t si) {
    h &= HT_SIZE - 1;
    int64_t i = 0;
    if (c->in_use[h / 32] & (1u << (h % 32))) i = c->table[h];
    i += si & ~(int64_t)WIN_MASK;
    if (i >= si) i -= WIN_SIZE; /* previous 64 KiB segment; negative in the first */
    return i;
}
static inline void fast_put(fast_state *c, uint32_t h, int64_t si) {
    h &= HT_SIZE - 1;
    c->table[h] = (uint16_t)si;
    c->in_use[h / 32] |= 1u << (h % 32);
}

/* block.go:105-295 */
static int64_t fast_compress(fast_state *c, const uint8_t *src, int64_t n, uint8_t *dst,
                             int64_t dlen) {
    memset(c->in_use, 0, sizeof c->in_use); /* reset(): block.go:94,107 */
    const int not_compressible = dlen < (int64_t)lz4o_compress_bound((size_t)n); /* :110 */
    int64_t si = 0, di = 0, anchor = 0;
    const int64_t sn = n - MF_LIMIT;
    if (sn <= 0) goto last_literals;

    while (si < sn) {
        uint64_t match = ld64(src + si);
        uint32_t h = lz4o_block_hash(match);
        uint32_t h2 = lz4o_block_hash(match >> 8);
        /* gets before puts: block.go:134-137 */
        int64_t ref = fast_get(c, h, si);
        int64_t ref2 = fast_get(c, h2, si + 1);
        fast_put(c, h, si);
        fast_put(c, h2, si + 1);

        int64_t offset = si - ref;
        if (offset <= 0 || offset >= WIN_SIZE || (uint32_t)match != ld32(src + ref)) {
            h = lz4o_block_hash(match >> 16);
            int64_t ref3 = fast_get(c, h, si + 2); /* after the two puts: :145 */
            si += 1;
            offset = si - ref2;
            if (offset <= 0 || offset >= WIN_SIZE ||
                (uint32_t)(match >> 8) != ld32(src + ref2)) {
                si += 1;
                offset = si - ref3;
                fast_put(c, h, si); /* :155 */
                if (offset <= 0 || offset >= WIN_SIZE ||
                    (uint32_t)(match >> 16) != ld32(src + ref3)) {
                    si += 2 + ((si - anchor) >> ADAPT_SKIP_LOG); /* :159 */
                    continue;
                }
            }
        }

        /* match found: :165-193 */
        int64_t l_len = si - anchor;
        int64_t m_len = 4;
        int64_t t_off = si - offset - 1;
        while (l_len > 0 && t_off >= 0 && src[si - 1] == src[t_off]) {
            si--;
            t_off--;
            l_len--;
            m_len++;
        }
        int64_t base = si + MIN_MATCH; /* "mLen" reused as offset base: :181 */
        si += m_len;
        while (si + 8 <= sn) {
            uint64_t x = ld64(src + si) ^ ld64(src + si - offset);
            if (x == 0) {
                si += 8;
            } else {
                si += __builtin_ctzll(x) >> 3;
                break;
            }
        }
        m_len = si - base;

        /* emit: :195-248 */
        if (di >= dlen) return LZ4O_ERR_SHORT_BUFFER;
        dst[di] = m_len < 0xF ? (uint8_t)m_len : 0xF;
        if (l_len < 0xF) {
            dst[di] |= (uint8_t)(l_len << 4);
        } else {
            dst[di] |= 0xF0;
            di++;
            int64_t l = l_len - 0xF;
            for (; l >= 0xFF && di < dlen; l -= 0xFF) dst[di++] = 0xFF;
            if (di >= dlen) return LZ4O_ERR_SHORT_BUFFER;
            dst[di] = (uint8_t)l;
        }
        di++;
        if (di + l_len > dlen) return LZ4O_ERR_SHORT_BUFFER;
        memcpy(dst + di, src + anchor, (size_t)l_len);
        di += l_len + 2;
        anchor = si;
        if (di > dlen) return LZ4O_ERR_SHORT_BUFFER;
        dst[di - 2] = (uint8_t)offset;
        dst[di - 1] = (uint8_t)(offset >> 8);
        if (m_len >= 0xF) {
            for (m_len -= 0xF; m_len >= 0xFF && di < dlen; m_len -= 0xFF) dst[di++] = 0xFF;
            if (di >= dlen) return LZ4O_ERR_SHORT_BUFFER;
            dst[di++] = (uint8_t)m_len;
        }
        if (si >= sn) break;                                 /* :250-252 */
        fast_put(c, lz4o_block_hash(ld64(src + si - 2)), si - 2); /* :254-255 */
    }

last_literals:
    if (not_compressible && anchor == 0) return 0; /* :259-262 */
    if (di >= dlen) return LZ4O_ERR_SHORT_BUFFER;
    {
        int64_t l_len = n - anchor;
        if (l_len < 0xF) {
            dst[di] = (uint8_t)(l_len << 4);
        } else {
            dst[di] = 0xF0;
            di++;
            for (l_len -= 0xF; l_len >= 0xFF && di < dlen; l_len -= 0xFF) dst[di++] = 0xFF;
            if (di >= dlen) return LZ4O_ERR_SHORT_BUFFER;
            dst[di] = (uint8_t)l_len;
        }
        di++;
    }
    if (not_compressible && di >= anchor) return 0; /* :286-289 */
    if (di + n - anchor > dlen) return LZ4O_ERR_SHORT_BUFFER;
    memcpy(dst + di, src + anchor, (size_t)(n - anchor));
    di += n - anchor;
    return di;
}

int64_t lz4o_compress_fast(const uint8_t *src, size_t n, uint8_t *dst, size_t dst_len) {
    fast_state *c = (fast_state *)malloc(sizeof *c);
    if (!c) return LZ4O_ERR_SHORT_BUFFER;
    int64_t r = fast_compress(c, src, (int64_t)n, dst, (int64_t)dst_len);
    free(c);
    return r;
}

/* ---- HC compressor: block.go:303-481 ----
 * Go's unchecked dst writes panic on index >= len(dst) and on slice end >
 * cap(dst); recoverBlock (:28-32,327) turns the panic into
 * ErrInvalidSourceShortBuffer.  The C ABI has one length, so cap == len. */
typedef struct {
    int32_t hash_table[HT_SIZE];
    int32_t chain_table[HT_SIZE];
} hc_state;

#define HC_PUT(idx, val)                                    \
    do {                                                    \
        if ((idx) >= dlen) return LZ4O_ERR_SHORT_BUFFER;    \
        dst[(idx)] = (uint8_t)(val);                        \
    } while (0)

static int64_t hc_compress(hc_state *c, const uint8_t *src, int64_t n, uint8_t *dst, int64_t dlen,
                           uint32_t depth) {
    memset(c, 0, sizeof *c); /* :320-324 */
    const int not_compressible = dlen < (int64_t)lz4o_compress_bound((size_t)n); /* :330 */
    int64_t si = 0, di = 0, anchor = 0;
    const int64_t sn = n - MF_LIMIT;
    if (sn <= 0) goto last_literals; /* skips the anchor==0 early-out: :339-341 */
    if (depth == 0) depth = WIN_SIZE; /* :343-345 */

    while (si < sn) {
        uint32_t match = ld32(src + si);
        uint32_t h = lz4o_block_hash_hc(match);
        int64_t m_len = 0, offset = 0;
        int64_t next = c->hash_table[h];
        for (uint32_t try_ = depth; try_ > 0 && next > 0 && si - next < WIN_SIZE;
             next = c->chain_table[next & WIN_MASK], try_--) {
            if (src[next + m_len] != src[si + m_len]) continue; /* gate: :358-360 */
            int64_t ml = 0;
            while (ml < sn - si) { /* may overshoot sn-si by up to 7: :363-372 */
                uint64_t x = ld64(src + next + ml) ^ ld64(src + si + ml);
                if (x == 0) {
                    ml += 8;
                } else {
                    ml += __builtin_ctzll(x) >> 3;
                    break;
                }
            }
            if (ml < MIN_MATCH || ml <= m_len) continue;
            m_len = ml;
            offset = si - next;
        }
        c->chain_table[si & WIN_MASK] = c->hash_table[h];
        c->hash_table[h] = (int32_t)si;

        if (m_len == 0) { /* :386-389 */
            si += 1 + ((si - anchor) >> ADAPT_SKIP_LOG);
            continue;
        }

        /* insert every position covered by the match: :394-405 */
        int64_t win_start = si + 1;
        int64_t ws = si + m_len - WIN_SIZE;
        if (ws > win_start) win_start = ws;
        for (int64_t p = win_start, end = si + m_len; p < end; p++) {
            match >>= 8;
            match |= (uint32_t)src[p + 3] << 24;
            uint32_t hh = lz4o_block_hash_hc(match);
            c->chain_table[p & WIN_MASK] = c->hash_table[hh];
            c->hash_table[hh] = (int32_t)p;
        }

        int64_t l_len = si - anchor;
        si += m_len;
        m_len -= MIN_MATCH;

        /* emit: :411-449 */
        HC_PUT(di, m_len < 0xF ? m_len : 0xF);
        if (l_len < 0xF) {
            dst[di] |= (uint8_t)(l_len << 4);
        } else {
            dst[di] |= 0xF0;
            di++;
            int64_t l = l_len - 0xF;
            for (; l >= 0xFF; l -= 0xFF) {
                HC_PUT(di, 0xFF);
                di++;
            }
            HC_PUT(di, l);
        }
        di++;
        if (di + l_len > dlen) return LZ4O_ERR_SHORT_BUFFER; /* slice end > cap */
        memcpy(dst + di, src + anchor, (size_t)l_len);
        di += l_len;
        anchor = si;
        di += 2;
        if (di - 1 >= dlen) return LZ4O_ERR_SHORT_BUFFER;
        dst[di - 2] = (uint8_t)offset;
        dst[di - 1] = (uint8_t)(offset >> 8);
        if (m_len >= 0xF) {
            for (m_len -= 0xF; m_len >= 0xFF; m_len -= 0xFF) {
                HC_PUT(di, 0xFF);
                di++;
            }
            HC_PUT(di, m_len);
            di++;
        }
    }

    if (not_compressible && anchor == 0) return 0; /* :452-455 */

last_literals : {
    int64_t l_len = n - anchor;
    if (l_len < 0xF) {
        HC_PUT(di, l_len << 4);
    } else {
        HC_PUT(di, 0xF0);
        di++;
        l_len -= 0xF;
        for (; l_len >= 0xFF; l_len -= 0xFF) {
            HC_PUT(di, 0xFF);
            di++;
        }
        HC_PUT(di, l_len);
    }
    di++;
}
    if (not_compressible && di >= anchor) return 0; /* :475-478 */
    if (di + n - anchor > dlen) return LZ4O_ERR_SHORT_BUFFER;
    memcpy(dst + di, src + anchor, (size_t)(n - anchor));
    di += n - anchor;
    return di;
}

int64_t lz4o_compress_hc(const uint8_t *src, size_t n, uint8_t *dst, size_t dst_len,
                         uint32_t depth) {
    hc_state *c = (hc_state *)malloc(sizeof *c);
    if (!c) return LZ4O_ERR_SHORT_BUFFER;
    int64_t r = hc_compress(c, src, (int64_t)n, dst, (int64_t)dst_len, depth);
    free(c);
    return r;
}

/* ---- decoder: decode_other.go:10-137; error classes decode_amd64.s:438-448 ----
 * Restated without the 16/18-byte over-copy shortcuts (they change only bytes
 * at or beyond the returned length).  Never writes at or beyond dst_len. */
int64_t lz4o_decode_block(uint8_t *dst, size_t dst_len, const uint8_t *src, size_t n,
                          const uint8_t *dict, size_t dict_len) {
    if (n == 0) return -1; /* decode_other.go:17-19, decode_amd64.s:32-33 */
    size_t si = 0, di = 0;
    while (si < n) {
        unsigned b = src[si++];
        size_t l_len = b >> 4;
        if (l_len > 0) {
            if (l_len == 0xF) {
                for (;;) {
                    if (si >= n) return -2;
                    unsigned x = src[si++];
                    l_len += x;
                    if (x != 0xFF) break;
                }
            }
            if (l_len > n - si) return -2;
            if (l_len > dst_len - di) return -2;
            memcpy(dst + di, src + si, l_len);
            si += l_len;
            di += l_len;
        }
        size_t m_len = b & 0xF;
        if (si == n && m_len == 0) break; /* decode_other.go:81-85 */
        if (si >= n) return -2;
        if (n - si < 2) return -2;
        size_t offset = (size_t)src[si] | ((size_t)src[si + 1] << 8);
        if (offset == 0) return -1; /* :88-90 */
        si += 2;
        m_len += MIN_MATCH;
        if (m_len == MIN_MATCH + 0xF) {
            for (;;) {
                if (si >= n) return -2;
                unsigned x = src[si++];
                m_len += x;
                if (x != 0xFF) break;
            }
        }
        if (m_len > dst_len - di) return -2;
        if (di < offset) { /* first part in the dictionary: :109-121 */
            size_t need = offset - di;
            if (need > dict_len) return -3;
            size_t k = m_len < need ? m_len : need;
            memcpy(dst + di, dict + dict_len - need, k);
            di += k;
            m_len -= k;
            if (m_len == 0) continue;
        }
        /* byte-serial semantics cover overlapping matches: :123-133 */
        const uint8_t *from = dst + di - offset;
        for (size_t k = 0; k < m_len; k++) dst[di + k] = from[k];
        di += m_len;
    }
    return (int64_t)di;
}

/* block.go:44-52 */
int64_t lz4o_uncompress_block(const uint8_t *src, size_t n, uint8_t *dst, size_t dst_len,
                              const uint8_t *dict, size_t dict_len) {
    if (n == 0) return 0;
    int64_t di = lz4o_decode_block(dst, dst_len, src, n, dict, dict_len);
    return di >= 0 ? di : LZ4O_ERR_SHORT_BUFFER;
}

/* ---- xxh32: internal/xxh32/xxh32zero.go ---- */
#define P1 2654435761u
#define P2 2246822519u
#define P3 3266489917u
#define P4 668265263u
#define P5 374761393u
static inline uint32_t rol(uint32_t u, int k) { return (u << k) | (u >> (32 - k)); }
static inline uint32_t xxh_round(uint32_t v, uint32_t x) { return rol(v + x * P2, 13) * P1; }

static uint32_t xxh_tail(uint32_t h32, const uint8_t *p, size_t n) {
    size_t i = 0;
    for (; i + 4 <= n; i += 4) { /* xxh32zero.go:166-170 */
        h32 += ld32(p + i) * P3;
        h32 = rol(h32, 17) * P4;
    }
    for (; i < n; i++) { /* :171-175 */
        h32 += (uint32_t)p[i] * P5;
        h32 = rol(h32, 11) * P1;
    }
    h32 ^= h32 >> 15; /* :177-181 */
    h32 *= P2;
    h32 ^= h32 >> 13;
    h32 *= P3;
    h32 ^= h32 >> 16;
    return h32;
}

/* xxh32zero.go:142-184 */
uint32_t lz4o_xxh32(const uint8_t *p, size_t n) {
    uint32_t h32 = (uint32_t)n;
    size_t i = 0;
    if (n < 16) {
        h32 += P5;
    } else {
        uint32_t v1 = P1 + P2, v2 = P2, v3 = 0, v4 = 0u - P1;
        for (; i + 16 <= n; i += 16) {
            v1 = xxh_round(v1, ld32(p + i));
            v2 = xxh_round(v2, ld32(p + i + 4));
            v3 = xxh_round(v3, ld32(p + i + 8));
            v4 = xxh_round(v4, ld32(p + i + 12));
        }
        h32 += rol(v1, 1) + rol(v2, 7) + rol(v3, 12) + rol(v4, 18);
    }
    return xxh_tail(h32, p + i, n - i);
}

/* xxh32zero.go:37-44 */
void lz4o_xxh32_reset(lz4o_xxh32_state *s) {
    s->v[0] = P1 + P2;
    s->v[1] = P2;
    s->v[2] = 0;
    s->v[3] = 0u - P1;
    s->total_len = 0;
    s->bufused = 0;
}

/* xxh32zero.go:58-109 */
void lz4o_xxh32_write(lz4o_xxh32_state *s, const uint8_t *p, size_t n) {
    if (s->total_len == 0) lz4o_xxh32_reset(s);
    s->total_len += n;
    size_t m = (size_t)s->bufused;
    if (n < 16 - m) {
        memcpy(s->buf + m, p, n);
        s->bufused += (int)n;
        return;
    }
    if (m != 0) {
        size_t c = 16 - m;
        memcpy(s->buf + m, p, c);
        p += c;
        n -= c;
        s->v[0] = xxh_round(s->v[0], ld32(s->buf));
        s->v[1] = xxh_round(s->v[1], ld32(s->buf + 4));
        s->v[2] = xxh_round(s->v[2], ld32(s->buf + 8));
        s->v[3] = xxh_round(s->v[3], ld32(s->buf + 12));
    }
    size_t i = 0;
    for (; i + 16 <= n; i += 16) {
        s->v[0] = xxh_round(s->v[0], ld32(p + i));
        s->v[1] = xxh_round(s->v[1], ld32(p + i + 4));
        s->v[2] = xxh_round(s->v[2], ld32(p + i + 8));
        s->v[3] = xxh_round(s->v[3], ld32(p + i + 12));
    }
    s->bufused = (int)(n - i);
    memcpy(s->buf, p + i, n - i);
}

/* xxh32zero.go:112-139.  NOTE :113-114: the ">= 16 bytes seen" test is made on
 * totalLen truncated to 32 bits, so a stream of 2^32*k + r bytes with r < 16
 * takes the short-input branch and ignores the accumulators. */
uint32_t lz4o_xxh32_sum32(const lz4o_xxh32_state *s) {
    uint32_t h32 = (uint32_t)s->total_len;
    if (h32 >= 16)
        h32 += rol(s->v[0], 1) + rol(s->v[1], 7) + rol(s->v[2], 12) + rol(s->v[3], 18);
    else
        h32 += P5;
    return xxh_tail(h32, s->buf, (size_t)s->bufused);
}

/* ---- frame: internal/lz4stream ---- */
#define FRAME_MAGIC 0x184D2204u        /* frame.go:18 */
#define FRAME_SKIP_MAGIC 0x184D2A50u   /* frame.go:19 */
#define FRAME_MAGIC_LEGACY 0x184C2102u /* frame.go:20 */
#define LEGACY_BLOCK (8u << 20)         /* blocks.go:11 */

/* blocks.go:22-36 */
static int block_size_index(uint32_t bs) {
    switch (bs) {
    case 1u << 16: return 4;
    case 1u << 18: return 5;
    case 1u << 20: return 6;
    case 1u << 22: return 7;
    }
    return 0;
}
static uint32_t block_size_from_index(int idx) {
    return (idx >= 4 && idx <= 7) ? 1u << (8 + 2 * idx) : 0;
}

/* frame.go:144-167,202-204; frame_gen.go:7-29 */
size_t lz4o_frame_header(uint8_t *dst, uint32_t block_size, uint32_t flags,
                         uint64_t content_size) {
    uint16_t f = 0;
    f |= 1u << 6; /* version 1: frame.go:139-142 */
    f |= 1u << 5; /* block independence */
    if (flags & LZ4O_F_BLOCK_CHECKSUM) f |= 1u << 4;
    if (flags & LZ4O_F_CONTENT_SIZE) f |= 1u << 3;
    if (flags & LZ4O_F_CONTENT_CHECKSUM) f |= 1u << 2;
    f |= (uint16_t)(block_size_index(block_size) & 7) << 12;
    st32(dst, FRAME_MAGIC);
    size_t k = 4;
    dst[k++] = (uint8_t)f;
    dst[k++] = (uint8_t)(f >> 8);
    if (flags & LZ4O_F_CONTENT_SIZE) {
        memcpy(dst + k, &content_size, 8);
        k += 8;
    }
    dst[k] = (uint8_t)(lz4o_xxh32(dst + 4, k - 4) >> 8);
    return k + 1;
}

size_t lz4o_frame_bound(size_t n, uint32_t block_size, uint32_t flags) {
    if (flags & LZ4O_F_LEGACY) block_size = LEGACY_BLOCK;
    size_t nb = n / block_size + 2;
    /* a legacy short last block may grow to CompressBlockBound(len) */
    return 15 + n + nb * 8 + 8 + ((flags & LZ4O_F_LEGACY) ? n % block_size / 255 + 16 : 0);
}

/* lz4stream/block.go:224-276 for one block; returns bytes appended.  legacy_tmp != NULL:
 * legacy frame, dst = data[:cap(data)] = 8 MiB whatever len(src) is (:226-227), compressed
 * into legacy_tmp (8 MiB) first because `out` need not have that much room. */
static size_t frame_block(const uint8_t *src, size_t n, uint32_t level, uint32_t flags,
                          uint8_t *out, uint32_t *size_word, uint32_t *xxh, uint8_t *legacy_tmp) {
    uint8_t *cdst = legacy_tmp ? legacy_tmp : out + 4;
    const size_t dlen = legacy_tmp ? (size_t)LEGACY_BLOCK : n;
    int64_t r = level == 0 ? lz4o_compress_fast(src, n, cdst, dlen)
                           : lz4o_compress_hc(src, n, cdst, dlen, level);
    uint32_t sw;
    size_t payload;
    if (r <= 0) { /* n == 0: incompressible or error => stored raw: :238-245 */
        memcpy(out + 4, src, n);
        payload = n;
        sw = (uint32_t)n | 0x80000000u;
    } else {
        if (legacy_tmp) memcpy(out + 4, legacy_tmp, (size_t)r);
        payload = (size_t)r;
        sw = (uint32_t)r;
    }
    st32(out, sw);
    size_t k = 4 + payload;
    uint32_t cs = 0;
    if (flags & LZ4O_F_BLOCK_CHECKSUM) {
        cs = lz4o_xxh32(src, n); /* over the UNCOMPRESSED bytes: :248-250 */
        if (cs != 0) {           /* zero checksum is not written: :270-272 */
            st32(out + k, cs);
            k += 4;
        }
    }
    if (size_word) *size_word = sw;
    if (xxh) *xxh = cs;
    return k;
}

int64_t lz4o_frame_compress(const uint8_t *src, size_t n, uint32_t block_size, uint32_t flags,
                            uint32_t level, uint64_t content_size, uint8_t *dst, size_t dst_cap) {
    const int legacy = (flags & LZ4O_F_LEGACY) != 0;
    if (legacy) block_size = LEGACY_BLOCK; /* frame.go:47-50 */
    else if (!block_size_index(block_size)) return -1;
    if (dst_cap < lz4o_frame_bound(n, block_size, flags)) return -1;
    uint8_t *tmp = NULL;
    size_t k;
    if (legacy) { /* magic only: frame.go:150-165 */
        tmp = (uint8_t *)malloc(LEGACY_BLOCK);
        if (!tmp) return -1;
        st32(dst, FRAME_MAGIC_LEGACY);
        k = 4;
    } else {
        k = lz4o_frame_header(dst, block_size, flags, content_size);
    }
    lz4o_xxh32_state cs;
    memset(&cs, 0, sizeof cs);
    lz4o_xxh32_reset(&cs); /* frame.go:56 */
    size_t off = 0;
    while (n - off >= block_size) { /* full blocks: writer.go:88-119 / :220-233 */
        k += frame_block(src + off, block_size, level, flags, dst + k, NULL, NULL, tmp);
        if (flags & LZ4O_F_CONTENT_CHECKSUM) lz4o_xxh32_write(&cs, src + off, block_size);
        off += block_size;
    }
    /* Close() flushes a short block only if bytes are pending (writer.go:161-167);
     * ReadFrom writes data[:rn] even when rn == 0 (writer.go:220-233). */
    if (n - off > 0 || (flags & LZ4O_F_READFROM)) {
        k += frame_block(src + off, n - off, level, flags, dst + k, NULL, NULL, tmp);
        if (flags & LZ4O_F_CONTENT_CHECKSUM) lz4o_xxh32_write(&cs, src + off, n - off);
    }
    free(tmp);
    if (legacy) return (int64_t)k; /* no end mark, no content checksum: frame.go:63-65 */
    st32(dst + k, 0); /* end mark: frame.go:66-68 */
    k += 4;
    if (flags & LZ4O_F_CONTENT_CHECKSUM) {
        st32(dst + k, lz4o_xxh32_sum32(&cs));
        k += 4;
    }
    return (int64_t)k;
}

/* Legacy frame (magic 0x184C2102): frame.go:46-51,169-174; block.Read for legacy,
 * lz4stream/block.go:284-294: 8 MiB blocks, no checksums, no end mark; a size word
 * equal to the legacy magic starts a concatenated frame; a size word equal to the
 * bytes decoded so far ends the stream (Linux kernel appends the total size). */
static int64_t legacy_uncompress(const uint8_t *src, size_t n, size_t si, uint8_t *dst,
                                 size_t dst_cap) {
    const uint32_t cap = 8u << 20;
    size_t di = 0;
    while (n - si >= 4) {
        uint32_t x = ld32(src + si);
        si += 4;
        if (x == FRAME_MAGIC_LEGACY) continue;
        if (x == (uint32_t)di) return (int64_t)di; /* :289-293: io.EOF, whatever follows */
        uint32_t size = x & 0x7FFFFFFFu;
        if (size > cap) return LZ4O_FE_BLOCK_SIZE;
        if (n - si < size) return LZ4O_FE_TRUNCATED;
        size_t room = dst_cap - di < cap ? dst_cap - di : cap;
        int64_t got;
        if (x >> 31) {
            if (size > room) return LZ4O_FE_SHORT_BUFFER;
            memcpy(dst + di, src + si, size);
            got = size;
        } else {
            got = lz4o_uncompress_block(src + si, size, dst + di, room, NULL, 0);
            if (got < 0) return LZ4O_FE_SHORT_BUFFER;
        }
        si += size;
        di += (size_t)got;
    }
    /* readUint32 is an io.ReadFull of 4 bytes (lz4stream/block.go:342-345): none left is
     * io.EOF, the clean end; one to three left is io.ErrUnexpectedEOF, an error */
    if (n - si != 0) return LZ4O_FE_TRUNCATED;
    return (int64_t)di;
}

int64_t lz4o_frame_uncompress(const uint8_t *src, size_t n, uint8_t *dst, size_t dst_cap) {
    size_t si = 0;
    uint32_t magic;
    for (;;) { /* frame.go:86-105 */
        if (n - si < 4) return LZ4O_FE_TRUNCATED;
        magic = ld32(src + si);
        si += 4;
        if (magic == FRAME_MAGIC) break;
        if (magic >> 8 == FRAME_SKIP_MAGIC >> 8) {
            if (n - si < 4) return LZ4O_FE_TRUNCATED;
            uint32_t skip = ld32(src + si);
            si += 4;
            if (n - si < skip) return LZ4O_FE_TRUNCATED;
            si += skip;
            continue;
        }
        if (magic == FRAME_MAGIC_LEGACY) return legacy_uncompress(src, n, si, dst, dst_cap);
        return LZ4O_FE_INVALID_FRAME;
    }
    /* descriptor: frame.go:169-200 */
    if (n - si < 3) return LZ4O_FE_TRUNCATED;
    size_t d0 = si;
    uint16_t f = (uint16_t)(src[si] | (src[si + 1] << 8));
    si += 2;
    if (f >> 3 & 1) {
        if (n - si < 9) return LZ4O_FE_TRUNCATED;
        si += 8;
    }
    uint8_t hc = src[si];
    if ((uint8_t)(lz4o_xxh32(src + d0, si - d0) >> 8) != hc) return LZ4O_FE_HEADER_CHECKSUM;
    si += 1;
    uint32_t bs = block_size_from_index(f >> 12 & 7);
    if (!bs) return LZ4O_FE_BLOCK_SIZE;
    const int block_sum = f >> 4 & 1, content_sum = f >> 2 & 1, independent = f >> 5 & 1;

    lz4o_xxh32_state cs;
    memset(&cs, 0, sizeof cs);
    lz4o_xxh32_reset(&cs);
    size_t di = 0;
    for (;;) { /* lz4stream/block.go:279-340 */
        if (n - si < 4) return LZ4O_FE_TRUNCATED;
        uint32_t sw = ld32(src + si);
        si += 4;
        if (sw == 0) break;
        uint32_t size = sw & 0x7FFFFFFFu;
        if (size > bs) return LZ4O_FE_BLOCK_SIZE; /* :301-304 */
        if (n - si < size) return LZ4O_FE_TRUNCATED;
        const uint8_t *payload = src + si;
        si += size;
        uint32_t want = 0;
        if (block_sum) {
            if (n - si < 4) return LZ4O_FE_TRUNCATED;
            want = ld32(src + si);
            si += 4;
        }
        size_t room = dst_cap - di < bs ? dst_cap - di : bs;
        int64_t got;
        if (sw >> 31) {
            if (size > room) return LZ4O_FE_SHORT_BUFFER;
            memcpy(dst + di, payload, size);
            got = size;
        } else {
            /* linked blocks: previous output is the dictionary (reader.go:180-189) */
            size_t dl = independent ? 0 : (di > WIN_SIZE ? WIN_SIZE : di);
            got = lz4o_uncompress_block(payload, size, dst + di, room, dst + di - dl, dl);
            if (got < 0) return LZ4O_FE_SHORT_BUFFER;
        }
        if (block_sum && lz4o_xxh32(dst + di, (size_t)got) != want) return LZ4O_FE_BLOCK_CHECKSUM;
        if (content_sum) lz4o_xxh32_write(&cs, dst + di, (size_t)got);
        di += (size_t)got;
    }
    if (content_sum) { /* frame.go:117-131 */
        if (n - si < 4) return LZ4O_FE_TRUNCATED;
        if (ld32(src + si) != lz4o_xxh32_sum32(&cs)) return LZ4O_FE_FRAME_CHECKSUM;
    }
    return (int64_t)di;
}

/* ---- threaded batch helpers (tests, bench cpu_baseline) ---- */
typedef struct {
    int tid, threads, decode;
    const uint8_t *src;
    size_t n, stride;
    uint32_t block_size, level;
    uint8_t *dst;
    uint32_t *size_words, *block_xxh;
    const uint32_t *in_words;
    int32_t *out_len;
    size_t n_blocks;
} job_t;

static void *job_main(void *arg) {
    job_t *j = (job_t *)arg;
    if (!j->decode) {
        fast_state *fs = (fast_state *)malloc(sizeof *fs);
        hc_state *hs = j->level ? (hc_state *)malloc(sizeof *hs) : NULL;
        for (size_t b = (size_t)j->tid; b < j->n_blocks; b += (size_t)j->threads) {
            size_t off = b * j->block_size;
            size_t len = j->n - off < j->block_size ? j->n - off : j->block_size;
            uint8_t *out = j->dst + b * j->stride;
            int64_t r = j->level == 0
                            ? fast_compress(fs, j->src + off, (int64_t)len, out, (int64_t)len)
                            : hc_compress(hs, j->src + off, (int64_t)len, out, (int64_t)len,
                                          j->level);
            if (r <= 0) {
                memcpy(out, j->src + off, len);
                j->size_words[b] = (uint32_t)len | 0x80000000u;
            } else {
                j->size_words[b] = (uint32_t)r;
            }
            if (j->block_xxh) j->block_xxh[b] = lz4o_xxh32(j->src + off, len);
        }
        free(fs);
        free(hs);
    } else {
        for (size_t b = (size_t)j->tid; b < j->n_blocks; b += (size_t)j->threads) {
            uint32_t sw = j->in_words[b], size = sw & 0x7FFFFFFFu;
            const uint8_t *in = j->src + b * j->stride;
            uint8_t *out = j->dst + b * (size_t)j->block_size;
            if (sw >> 31) {
                if (size > j->block_size) {
                    j->out_len[b] = -1;
                } else {
                    memcpy(out, in, size);
                    j->out_len[b] = (int32_t)size;
                }
            } else {
                j->out_len[b] =
                    (int32_t)lz4o_uncompress_block(in, size, out, j->block_size, NULL, 0);
            }
        }
    }
    return NULL;
}

static int run_jobs(job_t *proto, int threads) {
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_t th[256];
    job_t jobs[256];
    for (int t = 0; t < threads; t++) {
        jobs[t] = *proto;
        jobs[t].tid = t;
        jobs[t].threads = threads;
        if (pthread_create(&th[t], NULL, job_main, &jobs[t]) != 0) return -1;
    }
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    return 0;
}

int lz4o_compress_blocks_mt(const uint8_t *src, size_t n, uint32_t block_size, uint32_t level,
                            uint8_t *dst, size_t dst_stride, uint32_t *size_words,
                            uint32_t *block_xxh, int threads) {
    job_t j;
    memset(&j, 0, sizeof j);
    j.src = src;
    j.n = n;
    j.block_size = block_size;
    j.level = level;
    j.dst = dst;
    j.stride = dst_stride;
    j.size_words = size_words;
    j.block_xxh = block_xxh;
    j.n_blocks = (n + block_size - 1) / block_size;
    return run_jobs(&j, threads);
}

int lz4o_uncompress_blocks_mt(const uint8_t *src, size_t src_stride, const uint32_t *size_words,
                              size_t n_blocks, uint32_t block_size, uint8_t *dst,
                              int32_t *out_len, int threads) {
    job_t j;
    memset(&j, 0, sizeof j);
    j.decode = 1;
    j.src = src;
    j.stride = src_stride;
    j.in_words = size_words;
    j.n_blocks = n_blocks;
    j.block_size = block_size;
    j.dst = dst;
    j.out_len = out_len;
    return run_jobs(&j, threads);
}
