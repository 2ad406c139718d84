/*
 * lz4_oracle.h -- CPU restatement of pierrec/lz4 v4's block codec, xxh32 and
 * frame step.  TEST INFRASTRUCTURE ONLY.
 *
 * This is the parity oracle of the repository: only tests/, the smoke() check
 * of __graft_entry__.py and bench.py's cpu_baseline leg may call it.  The
 * shipped library (lz4_amd/csrc, include/lz4hip.h) never links or loads it.
 *
 * The reference is Go and no Go toolchain exists in the build image, so there
 * is no oracle/_ref build of the reference itself.  Pinning status:
 *   - decode, xxh32, frame descriptor: pinned by the reference's own exact
 *     known-answer tests (tests/test_oracle_kats.py).
 *   - compressed bytes (Fast / HC): PARITY UNPINNED by the reference's tests
 *     (writer_test.go:114 "Cannot compare compressed outputs directly").  They
 *     are held by this restatement, a second independent Python restatement
 *     (oracle/py_restatement.py), the survey's candidate sizes and round-trip
 *     through an independent decoder.
 *
 * Every function cites the reference file:line it follows
 * (paths relative to the reference checkout).
 */
#ifndef LZ4_ORACLE_H
#define LZ4_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Return conventions shared by the compressors:
 *   n > 0  compressed size
 *   0      (0, nil): incompressible, only when dst_len < bound
 *  -1      (0, ErrInvalidSourceShortBuffer) */
#define LZ4O_ERR_SHORT_BUFFER (-1)

/* internal/lz4block/block.go:40-42 */
size_t lz4o_compress_bound(size_t n);

/* internal/lz4block/block.go:35-38 and :298-301 */
uint32_t lz4o_block_hash(uint64_t x);
uint32_t lz4o_block_hash_hc(uint32_t x);

/* internal/lz4block/block.go:105-295 ((*Compressor).CompressBlock) */
int64_t lz4o_compress_fast(const uint8_t *src, size_t n, uint8_t *dst, size_t dst_len);

/* internal/lz4block/block.go:319-481 ((*CompressorHC).CompressBlock) */
int64_t lz4o_compress_hc(const uint8_t *src, size_t n, uint8_t *dst, size_t dst_len,
                         uint32_t depth);

/* decodeBlock: internal/lz4block/decode_other.go:10-137 (semantics),
 * decode_amd64.s:24-448 (error classes).  >= 0 bytes written, < 0 error. */
int64_t lz4o_decode_block(uint8_t *dst, size_t dst_len, const uint8_t *src, size_t n,
                          const uint8_t *dict, size_t dict_len);

/* UncompressBlock: internal/lz4block/block.go:44-52.  >= 0 or -1. */
int64_t lz4o_uncompress_block(const uint8_t *src, size_t n, uint8_t *dst, size_t dst_len,
                              const uint8_t *dict, size_t dict_len);

/* internal/xxh32/xxh32zero.go:142-184 (ChecksumZero) */
uint32_t lz4o_xxh32(const uint8_t *p, size_t n);

/* internal/xxh32/xxh32zero.go:22-139 (XXHZero streaming, incl. the uint32
 * truncation of totalLen in Sum32, :113-118) */
typedef struct {
    uint32_t v[4];
    uint64_t total_len;
    uint8_t buf[16];
    int bufused;
} lz4o_xxh32_state;
void lz4o_xxh32_reset(lz4o_xxh32_state *s);
void lz4o_xxh32_write(lz4o_xxh32_state *s, const uint8_t *p, size_t n);
uint32_t lz4o_xxh32_sum32(const lz4o_xxh32_state *s);

/* Frame flags for lz4o_frame_* (bit positions are this oracle's own). */
#define LZ4O_F_BLOCK_CHECKSUM 1u   /* BlockChecksumOption(true) */
#define LZ4O_F_CONTENT_CHECKSUM 2u /* ChecksumOption(true), the default */
#define LZ4O_F_CONTENT_SIZE 4u     /* SizeOption(content_size) */
#define LZ4O_F_LEGACY 16u          /* LegacyOption(true): options.go:230-242, frame.go:46-51 */
#define LZ4O_F_READFROM 8u         /* entry point is Writer.ReadFrom / io.Copy (writer.go:199-242) */

/* Frame descriptor: internal/lz4stream/frame.go:144-167,202-204, frame_gen.go:7-29.
 * Writes 7 or 15 bytes; returns the count. */
size_t lz4o_frame_header(uint8_t *dst, uint32_t block_size, uint32_t flags, uint64_t content_size);

size_t lz4o_frame_bound(size_t n, uint32_t block_size, uint32_t flags);

/* Writer.Write(src)+Close() (or io.Copy when LZ4O_F_READFROM) as one call:
 * writer.go:74-145,173-184,199-242; lz4stream/block.go:224-276; frame.go:59-74.
 * level: 0 = Fast, else HC chain depth (options.go:154-165).
 * Returns frame size, or -1 if dst_cap is too small / block_size invalid. */
int64_t lz4o_frame_compress(const uint8_t *src, size_t n, uint32_t block_size, uint32_t flags,
                            uint32_t level, uint64_t content_size, uint8_t *dst, size_t dst_cap);

/* Reader over one frame with independent or linked blocks:
 * reader.go:81-196; lz4stream/block.go:279-340; frame.go:80-131,169-200.
 * Returns bytes decoded, or a negative LZ4O_FE_* code. */
#define LZ4O_FE_INVALID_FRAME (-2)
#define LZ4O_FE_HEADER_CHECKSUM (-3)
#define LZ4O_FE_BLOCK_SIZE (-4)
#define LZ4O_FE_BLOCK_CHECKSUM (-5)
#define LZ4O_FE_FRAME_CHECKSUM (-6)
#define LZ4O_FE_SHORT_BUFFER (-1)
#define LZ4O_FE_TRUNCATED (-7)
int64_t lz4o_frame_uncompress(const uint8_t *src, size_t n, uint8_t *dst, size_t dst_cap);

/* Batch helpers used by tests and bench.py's cpu_baseline leg: compress or
 * decode n_blocks independent blocks with `threads` pthreads, frame-block
 * semantics of lz4stream/block.go:224-252 (dst_len = len(src); n == 0 => raw).
 * size_words[i] = stored size | raw<<31; block b's payload goes to
 * dst + b * dst_stride. */
int lz4o_compress_blocks_mt(const uint8_t *src, size_t n, uint32_t block_size, uint32_t level,
                            uint8_t *dst, size_t dst_stride, uint32_t *size_words,
                            uint32_t *block_xxh, int threads);
int lz4o_uncompress_blocks_mt(const uint8_t *src, size_t src_stride, const uint32_t *size_words,
                              size_t n_blocks, uint32_t block_size, uint8_t *dst,
                              int32_t *out_len, int threads);

#ifdef __cplusplus
}
#endif
#endif
