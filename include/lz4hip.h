/*
 * lz4hip.h -- C ABI of liblz4hip.so: the MI355X (gfx950) LZ4 block codec that
 * sits behind pierrec/lz4 v4's Go API for its hot path (block codec, xxh32
 * block checksums, frame block step, frame assembly).
 *
 * Plain C: pointers and sizes only.  "d_" pointers are DEVICE pointers
 * (hipMalloc'ed memory of the current device), "h_"/unprefixed pointers in the
 * host-pointer entry points are ordinary host memory.  `stream` is a
 * hipStream_t passed as void* (NULL = the default stream).  Device entry points
 * are asynchronous on `stream`, allocate nothing and keep no pointer after
 * return; host-pointer entry points are synchronous.
 *
 * Every entry point cites the reference interface it replaces (paths relative
 * to the pierrec/lz4 checkout).  The Go-side binding is shown in INTEGRATION.md.
 *
 * Return convention of the block compressors (per block, in d_result[b]):
 *   n > 0   compressed size                         -> (n, nil)
 *   0       incompressible (only if dst_len<bound)  -> (0, nil)
 *  -1       dst too small                           -> (0, ErrInvalidSourceShortBuffer)
 * Block decoder (d_out_len[b]): n >= 0 bytes written, or -1 for every failure
 * (corrupt input, short dst, short dict) -> (0, ErrInvalidSourceShortBuffer),
 * as internal/lz4block/block.go:48-51 folds the decoder's -1/-2/-3.
 */
#ifndef LZ4HIP_H
#define LZ4HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Function status codes (distinct from the per-block result convention). */
#define LZ4HIP_OK 0
#define LZ4HIP_ERR_ARG (-1)     /* bad argument (size, alignment, NULL) */
#define LZ4HIP_ERR_HIP (-2)     /* a HIP runtime call failed: lz4hip_last_error() */
#define LZ4HIP_ERR_SHORT (-3)   /* host-pointer frame calls: dst_cap too small */
#define LZ4HIP_ERR_FRAME (-4)   /* lz4errors.ErrInvalidFrame */
#define LZ4HIP_ERR_HEADER_CHECKSUM (-5) /* ErrInvalidHeaderChecksum */
#define LZ4HIP_ERR_BLOCK_SIZE (-6)      /* ErrOptionInvalidBlockSize */
#define LZ4HIP_ERR_BLOCK_CHECKSUM (-7)  /* ErrInvalidBlockChecksum */
#define LZ4HIP_ERR_FRAME_CHECKSUM (-8)  /* ErrInvalidFrameChecksum */
#define LZ4HIP_ERR_BLOCK (-9)           /* ErrInvalidSourceShortBuffer inside a frame */
/* The single-block host entry points return the Go convention n / 0 / -1, where -1 is the
 * CODEC's verdict (ErrInvalidSourceShortBuffer).  A failure of the call itself is never
 * folded into it: LZ4HIP_ERR_HIP (-2) for a HIP runtime failure, LZ4HIP_ERR_BADARG for a
 * bad argument.  A binding must surface those as real errors: in the frame step an
 * "error" from the block compressor means "store the block raw" (lz4stream/block.go:238-245). */
#define LZ4HIP_ERR_BADARG (-10)

/* Block sizes: internal/lz4block/blocks.go:6-12 (lz4.go:16-24 asserts equality). */
#define LZ4HIP_BLOCK_64KB (1u << 16)
#define LZ4HIP_BLOCK_256KB (1u << 18)
#define LZ4HIP_BLOCK_1MB (1u << 20)
#define LZ4HIP_BLOCK_4MB (1u << 22)

/* dst_mode of the batched compressors. */
#define LZ4HIP_DST_FRAME 0 /* dst_len = len(src): lz4stream/block.go:229 */
#define LZ4HIP_DST_BOUND 1 /* dst_len = CompressBlockBound(len(src)) */

/* Frame flags (options.go: BlockChecksumOption, ChecksumOption, SizeOption) and
 * the entry-point quirk of writer.go:220-233 (io.Copy -> Writer.ReadFrom writes
 * a trailing zero-length raw block when len(src) is a block multiple). */
#define LZ4HIP_F_BLOCK_CHECKSUM 1u
#define LZ4HIP_F_CONTENT_CHECKSUM 2u
#define LZ4HIP_F_CONTENT_SIZE 4u
#define LZ4HIP_F_READFROM 8u
/* LegacyOption(true) (options.go:230-242, frame.go:46-51,63-65): magic 0x184C2102
 * only, 8 MiB blocks each compressed into an 8 MiB dst (lz4stream/block.go:226-227),
 * no end mark, no content checksum; block_size and content_size are ignored. */
#define LZ4HIP_F_LEGACY 16u
/* Reported by lz4hip_frame_parse_header only: FLG bit 5 clear, blocks depend on the
 * previous 64 KiB of output (reader.go:86-90,176-189). */
#define LZ4HIP_F_LINKED 32u

const char *lz4hip_last_error(void);          /* thread-local text of the last failure */
int lz4hip_device_count(void);                /* GPUs visible; 0 if none */
const char *lz4hip_version(void);

/* Optional process set-up for the host-pointer entry points, the counterpart of the
 * reference's sync.Pools (internal/lz4block/block.go:96,310, blocks.go:42-87): device
 * memory freed by a call stays in the device's stream-ordered pool instead of going back
 * to the driver, so the next call's buffers (source, slots, frame, the 384 MiB Fast
 * workspace) cost no driver allocation.  device < 0: every visible GPU.  lz4hip_shutdown
 * trims the pools again.  Both return LZ4HIP_OK or LZ4HIP_ERR_HIP. */
int lz4hip_init(int device);
int lz4hip_shutdown(void);

/* lz4.CompressBlockBound: lz4.go:27-29, internal/lz4block/block.go:40-42. Host only. */
size_t lz4hip_compress_bound(size_t n);

/* ------------------------------------------------------------------------
 * Batched device entry points (the hot path).
 * d_src holds n bytes cut into ceil(n/block_size) blocks (the last may be
 * short); it must be 16-byte aligned and block_size one of the four sizes
 * above.  Block b compresses into d_dst + b*dst_stride (dst_stride >= the
 * dst_len implied by dst_mode, multiple of 16).
 * ---------------------------------------------------------------------- */

/* Fast level: (*Compressor).CompressBlock, internal/lz4block/block.go:105-295,
 * as called per block by FrameDataBlock.Compress, lz4stream/block.go:232-234. */
/* d_workspace (optional): lz4hip_fast_workspace_bytes(n_blocks) bytes, 16-byte
 * aligned.  With a workspace the hash tables live in global memory (L2 /
 * Infinity-Cache resident) and many wavefronts per CU run, one per block; with NULL --
 * or with at most 512 blocks, for which lz4hip_fast_workspace_bytes returns 0 -- each
 * table lives in LDS and a team of six wavefronts works on every block, one block per CU
 * (the regime of lz4.CompressBlock and of a Writer with its default options).  The
 * output bytes are the same.  The last 256
 * bytes of a workspace of the full size hold the block counter of the dynamic
 * schedule (waves take their next block when they finish one); a workspace that
 * is a whole number of 128 KiB tables still works, with a round-robin deal.
 * lz4hip_fast_workspace_bytes asks for 3072 tables (12 resident waves per CU, the measured
 * plateau) for most launches and for one table per block for 3329..4096 blocks, which then all
 * run at once (16 waves per CU with a compact LDS layout of the same kernel) instead of one
 * round and a third of one. */
size_t lz4hip_fast_workspace_bytes(size_t n_blocks);
int lz4hip_compress_fast_blocks(const void *d_src, size_t n, uint32_t block_size, int dst_mode,
                                void *d_dst, size_t dst_stride, int32_t *d_result,
                                void *d_workspace, size_t workspace_bytes, void *stream);

/* The same compressor for callers that cannot allocate device memory -- code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.  It has NO counterpart in
 * the reference.  It takes no workspace: the hash tables and the block counter live in a
 * file-scope device array of the library (64 MiB + 256 B per GPU, held until the process
 * ends), and the call itself allocates and frees nothing.  Same arguments, kernels and
 * output bytes as lz4hip_compress_fast_blocks.  n is limited to LZ4HIP_NOALLOC_MAX_BYTES; a
 * larger n is rejected with LZ4HIP_ERR_ARG before anything is launched.  Every call on a GPU
 * shares that GPU's array: calls that overlap on one GPU (different streams or threads)
 * race on it and must be serialised by the caller.
 * Under stream capture: the FIRST call on a device resolves the array's address
 * (hipGetSymbolAddress: loads the module, takes a lock) and must not happen inside a
 * capture -- call lz4hip_init(device), or make one call outside the capture, first.  After
 * that a call only enqueues a 4-byte hipMemsetAsync and one kernel, both capturable. */
#define LZ4HIP_NOALLOC_MAX_BYTES (4ull << 30)
int lz4hip_compress_fast_blocks_noalloc(const void *d_src, size_t n, uint32_t block_size,
                                        int dst_mode, void *d_dst, size_t dst_stride,
                                        int32_t *d_result, void *stream);

/* HC level: (*CompressorHC).CompressBlock, internal/lz4block/block.go:319-481,
 * lz4stream/block.go:235-237.  depth = chain depth (0 => 65536); the frame path
 * passes the CompressionLevel constant (options.go:154-165).
 * d_workspace: lz4hip_hc_workspace_bytes(n_blocks) bytes, 256-byte aligned (tables
 * for the resident blocks and, in the last 256 bytes, the counter from which waves
 * take their next block; without room for it the blocks are dealt round-robin). */
size_t lz4hip_hc_workspace_bytes(size_t n_blocks);
int lz4hip_compress_hc_blocks(const void *d_src, size_t n, uint32_t block_size, uint32_t depth,
                              int dst_mode, void *d_dst, size_t dst_stride, int32_t *d_result,
                              void *d_workspace, size_t workspace_bytes, void *stream);

/* UncompressBlock over n_blocks independent blocks: internal/lz4block/block.go:44-52,
 * decode_amd64.s:24-448 / decode_other.go:10-137, as called by
 * FrameDataBlock.Uncompress, lz4stream/block.go:319-329.
 * Block b's input is d_src + d_src_off[b], length d_size_words[b] & 0x7fffffff;
 * bit 31 set means stored raw (copied).  Output goes to d_dst + b*dst_stride,
 * at most dst_len bytes.  A zero-length compressed input yields 0 (block.go:45-47).
 * d_src holds src_len bytes: a block whose d_src_off[b] + length passes src_len is
 * rejected on the device (d_out_len[b] = -1) without reading it.
 * d_dict / dict_len (may be NULL / 0): lz4.UncompressBlockWithDict (lz4.go:47-49,
 * decode_other.go:109-121) -- every block of the batch decodes against the same
 * dictionary, of which the last 64 KiB are reachable; a match that starts before the
 * block's own output reads the dictionary's tail and continues in the output. */
int lz4hip_uncompress_blocks(const void *d_src, size_t src_len, const uint64_t *d_src_off,
                             const uint32_t *d_size_words, size_t n_blocks, void *d_dst,
                             size_t dst_stride, uint32_t dst_len, const void *d_dict,
                             size_t dict_len, int32_t *d_out_len, void *stream);

/* xxh32.ChecksumZero per block: internal/xxh32/xxh32zero.go:142-184, as used for
 * block checksums over UNCOMPRESSED bytes, lz4stream/block.go:248-250,330-335.
 * Span i is d_base + (d_off ? d_off[i] : i*stride), length d_len ? d_len[i] :
 * min(stride, n - i*stride). */
int lz4hip_xxh32_blocks(const void *d_base, size_t n, size_t stride, const uint64_t *d_off,
                        const int32_t *d_len, size_t n_spans, uint32_t *d_out, void *stream);

/* Streaming content checksum (XXHZero.Write over blocks in order + Sum32):
 * internal/xxh32/xxh32zero.go:58-139 incl. the uint32 truncation of totalLen
 * (:113-118); lz4stream/block.go:257-259, frame.go:69-71.  Serial by nature:
 * one wave walks the n bytes.  Writes one uint32 to d_out. */
int lz4hip_xxh32_stream(const void *d_src, size_t n, uint32_t *d_out, void *stream);

/* The same checksum for a stream cut across GPUs (block-range shards): the
 * XXHZero state {v1,v2,v3,v4,totalLen lo,hi} (6 x u32 in d_state) is carried
 * from shard to shard in stream order.  first != 0 starts from Reset()
 * (xxh32zero.go:37-44); every non-final piece must be a multiple of 16 bytes
 * (block sizes are); final_piece != 0 also writes Sum32() to d_digest. */
int lz4hip_xxh32_stream_update(const void *d_src, size_t n, uint32_t *d_state, int first,
                               int final_piece, uint32_t *d_digest, void *stream);

/* The same chained checksum with the chain on the HOST: the bytes of d_src come down through
 * pinned buffers and the calling thread runs XXHZero over them (synchronous; `stream`
 * carries the copies).  h_state / h_digest are HOST memory, same layout and rules as above.
 * One host core steps the serial chain ~4x faster than a GPU lane does, so this is the
 * variant the sharded frame path uses for ChecksumOption(true). */
int lz4hip_xxh32_stream_update_host(const void *d_src, size_t n, uint32_t *h_state, int first,
                                    int final_piece, uint32_t *h_digest, void *stream);
/* The same step over bytes that already are in HOST memory (XXHZero.Write / Sum32,
 * xxh32zero.go:58-139, on the calling thread; no GPU involved).  A rank of the sharded frame
 * path that waits for the state of the ranks before it stages its shard into pinned host
 * memory meanwhile, so that its turn in the chain is the hash alone
 * (lz4stream/block.go:257-259 is serial by construction). */
int lz4hip_xxh32_host_update(const void *h_src, size_t n, uint32_t *h_state, int first,
                             int final_piece, uint32_t *h_digest);

/* Frame assembly on device: FrameDescriptor.Write (frame.go:144-167),
 * FrameDataBlock.Write per block (lz4stream/block.go:254-276: size word,
 * payload or raw src when d_result[b] <= 0, checksum word unless 0) and
 * Frame.CloseW (frame.go:59-74).
 *  - lz4hip_frame_layout: exclusive scan of the per-block entry sizes;
 *    d_entry_off[b] = offset of block b's size word from the frame start,
 *    d_entry_off[n_blocks] = offset of the end mark, d_frame_len[0] = total.
 *    first_off = bytes before the first entry (header length, or the span
 *    start of a GPU shard when frames are assembled across GPUs).
 *  - lz4hip_frame_scatter: writes the entries at those offsets.
 *  - lz4hip_frame_header / trailer are tiny host helpers (7/15 and 4/8 bytes).
 */
int lz4hip_frame_layout(const int32_t *d_result, size_t n, uint32_t block_size, size_t n_blocks,
                        const uint32_t *d_block_xxh /* NULL: no block checksums */,
                        uint64_t first_off, uint64_t *d_entry_off /* n_blocks+1 */,
                        uint32_t *d_size_words /* n_blocks, may be NULL */, void *stream);
int lz4hip_frame_scatter(const void *d_src, size_t n, uint32_t block_size, const void *d_comp,
                         size_t comp_stride, const int32_t *d_result,
                         const uint32_t *d_block_xxh, const uint64_t *d_entry_off,
                         size_t n_blocks, void *d_frame, uint64_t frame_base, void *stream);
/* Reader side on device.  lz4hip_frame_index: FrameDataBlock.Read for every block of a
 * STANDARD frame held in device memory (lz4stream/block.go:279-317), starting at the first
 * size word (first_off = header length): payload offset, size word and -- with
 * block_checksums -- the stored checksum of each block, up to max_blocks.  d_info (4 x
 * int64): [0] blocks found, [1] offset behind the end mark (where the content checksum
 * word sits) or where the walk stopped, [2] 0 or the LZ4HIP_ERR_* the reader meets there
 * (LZ4HIP_ERR_FRAME = truncated, LZ4HIP_ERR_BLOCK_SIZE = size word above block_size),
 * [3] 1 when max_blocks was reached before the end mark.  One lane walks the words (each
 * one's place depends on the sizes before it).
 * lz4hip_pack_blocks: decoded blocks sit in fixed-stride slots; block b's d_len[b] bytes
 * (skipped when <= 0) are copied to d_flat + d_flat_off[b] -- the contiguous stream the
 * content checksum and the caller want when blocks are not all full. */
int lz4hip_frame_index(const void *d_frame, size_t n, uint64_t first_off, uint32_t block_size,
                       int block_checksums, size_t max_blocks, uint64_t *d_src_off,
                       uint32_t *d_size_words, uint32_t *d_want_xxh, int64_t *d_info, void *stream);
int lz4hip_pack_blocks(const void *d_slots, size_t stride, const int32_t *d_len,
                       const uint64_t *d_flat_off, size_t n_blocks, void *d_flat, void *stream);
size_t lz4hip_frame_header(uint8_t *h_dst16, uint32_t block_size, uint32_t flags,
                           uint64_t content_size);
size_t lz4hip_frame_trailer(uint8_t *h_dst8, uint32_t flags, uint32_t content_xxh);
/* Frame.ParseHeaders + FrameDescriptor.initR (internal/lz4stream/frame.go:80-111,169-200),
 * host only: skips skippable frames, accepts the standard and the legacy magic, checks
 * the descriptor checksum and the block-size index.  On LZ4HIP_OK: *block_size (8 MiB for
 * legacy), *flags (LZ4HIP_F_BLOCK_CHECKSUM | _CONTENT_CHECKSUM | _CONTENT_SIZE | _LEGACY |
 * _LINKED), *content_size, *header_len = offset of the first block.  Errors:
 * LZ4HIP_ERR_FRAME (bad magic), _HEADER_CHECKSUM, _BLOCK_SIZE, _SHORT (input ends inside
 * the header: Go's io.ErrUnexpectedEOF).  Any out pointer may be NULL. */
int lz4hip_frame_parse_header(const uint8_t *src, size_t n, uint32_t *block_size, uint32_t *flags,
                              uint64_t *content_size, size_t *header_len);
/* Upper bound of a frame of n source bytes. */
size_t lz4hip_frame_bound(size_t n, uint32_t block_size, uint32_t flags);

/* ------------------------------------------------------------------------
 * Host-pointer entry points: what the cgo shim binds.
 * They stage through device buffers owned by the call and synchronise.
 * ---------------------------------------------------------------------- */

/* lz4.CompressBlock / Compressor.CompressBlock: lz4.go:70-94. */
int64_t lz4hip_compress_block(const uint8_t *src, size_t n, uint8_t *dst, size_t dst_len);
/* lz4.CompressBlockHC / CompressorHC.CompressBlock: lz4.go:121-131. */
int64_t lz4hip_compress_block_hc(const uint8_t *src, size_t n, uint8_t *dst, size_t dst_len,
                                 uint32_t depth);
/* lz4.UncompressBlock / UncompressBlockWithDict: lz4.go:37-49. */
int64_t lz4hip_uncompress_block(const uint8_t *src, size_t n, uint8_t *dst, size_t dst_len,
                                const uint8_t *dict, size_t dict_len);
/* xxh32.ChecksumZero: internal/xxh32/xxh32zero.go:142.  lz4hip_xxh32 returns 0 when the
 * call itself fails, which a real checksum can also be: lz4hip_xxh32_checked returns
 * LZ4HIP_OK / LZ4HIP_ERR_HIP / LZ4HIP_ERR_BADARG and stores the checksum in *out. */
uint32_t lz4hip_xxh32(const uint8_t *p, size_t n);
int lz4hip_xxh32_checked(const uint8_t *p, size_t n, uint32_t *out);

/* Writer.Write(src)+Close(), or io.Copy(zw, src) with LZ4HIP_F_READFROM:
 * writer.go:74-145,173-184,199-242.  level 0 = Fast, else HC depth.
 * Returns the frame length or a negative LZ4HIP_ERR_*. */
int64_t lz4hip_frame_compress(const uint8_t *src, size_t n, uint32_t block_size, uint32_t flags,
                              uint32_t level, uint64_t content_size, uint8_t *dst,
                              size_t dst_cap);
/* The same call, also reporting len(block.Data) of every block written, in stream order:
 * the values Writer and CompressingReader hand to an OnBlockDoneOption handler
 * (writer.go:127, compressing_reader.go:126, options.go:197-220).  Up to max_blocks values
 * are stored in block_lens (may be NULL); *n_blocks (may be NULL) receives the block count. */
int64_t lz4hip_frame_compress_lens(const uint8_t *src, size_t n, uint32_t block_size,
                                   uint32_t flags, uint32_t level, uint64_t content_size,
                                   uint8_t *dst, size_t dst_cap, uint32_t *block_lens,
                                   size_t max_blocks, size_t *n_blocks);
/* io.ReadAll(lz4.NewReader(frame)): reader.go:81-196, lz4stream/block.go:279-340,
 * frame.go:80-131,169-200.  Independent blocks decode as one batch; linked blocks
 * (FLG bit 5 clear) decode in stream order with the output so far as history
 * (reader.go:176-189); legacy frames (magic 0x184C2102) per lz4stream/block.go:284-294.
 * Returns decoded length or a negative LZ4HIP_ERR_*. */
int64_t lz4hip_frame_uncompress(const uint8_t *src, size_t n, uint8_t *dst, size_t dst_cap);
/* The same call, also reporting each block's decoded length in stream order: the values
 * Reader.WriteTo hands to an OnBlockDoneOption handler (reader.go:255, options.go:197-220).
 * Up to max_blocks lengths are stored in block_lens (may be NULL); *n_blocks (may be NULL)
 * receives the number of blocks decoded, which may exceed max_blocks. */
int64_t lz4hip_frame_uncompress_lens(const uint8_t *src, size_t n, uint8_t *dst, size_t dst_cap,
                                     uint32_t *block_lens, size_t max_blocks, size_t *n_blocks);

/* ------------------------------------------------------------------------
 * Streaming batches: what Writer.Write / Reader.Read bind when ConcurrencyOption(n)
 * is set (writer.go:123-145, internal/lz4stream/block.go:21-58,95-185).  The
 * reference fans n blocks out to goroutines and an ordered writer / collector puts
 * them back in stream order; here the n blocks are ONE call, the frame header and
 * trailer stay with the caller (Frame.InitW / CloseW, frame.go:46-74), and memory is
 * bounded by n blocks however long the stream is.
 * A handle is the per-frame state: the XXHZero content checksum (Frame.checksum,
 * frame.go:27; on the host) and device buffers + a HIP stream sized for max_blocks
 * blocks, allocated once (the reference pools its block buffers, blocks.go:42-87).
 * A handle belongs to one Writer or one Reader; calls on one handle do not overlap.
 * flags: LZ4HIP_F_BLOCK_CHECKSUM | LZ4HIP_F_CONTENT_CHECKSUM | LZ4HIP_F_LEGACY
 * (write side only); level: 0 = Fast, else the HC depth (write side only).
 * Returns NULL on failure (lz4hip_last_error()).
 * ---------------------------------------------------------------------- */
typedef struct lz4hip_stream lz4hip_stream;
lz4hip_stream *lz4hip_stream_create(uint32_t block_size, uint32_t flags, uint32_t level,
                                    size_t max_blocks);
void lz4hip_stream_destroy(lz4hip_stream *s);
/* Writer.Reset / Reader.Reset -> Frame.Reset: checksum.Reset() (frame.go:35-44). */
void lz4hip_stream_reset(lz4hip_stream *s);
/* Frame.checksum.Sum32() for CloseW / CloseR (frame.go:69-71,125). */
uint32_t lz4hip_stream_checksum(const lz4hip_stream *s);
/* FrameDataBlock.Compress + Write for up to max_blocks blocks (lz4stream/block.go:224-276):
 * src is cut every block_size bytes (the last block may be short; n == 0 is the one empty
 * block of writer.go:220-233).  dst receives, in stream order, "u32 size | data | [u32
 * checksum]" per block -- the bytes the ordered writer goroutine would have written -- and
 * src is folded into the content checksum.  dst_cap >= n + 8 * n_blocks.  block_lens (may
 * be NULL): len(block.Data) per block, for OnBlockDoneOption (writer.go:127).
 * Returns the bytes written to dst or a negative LZ4HIP_ERR_*. */
int64_t lz4hip_stream_compress_blocks(lz4hip_stream *s, const uint8_t *src, size_t n, uint8_t *dst,
                                      size_t dst_cap, uint32_t *block_lens);
/* FrameDataBlock.Read + Uncompress for a run of blocks (lz4stream/block.go:279-340; standard
 * frames with independent blocks -- the reference drops to serial for linked ones,
 * reader.go:86-90).  src[0..n) starts at a size word.  Up to max_blocks entries that are
 * COMPLETE in src are decoded into dst in stream order, block checksums verified on the
 * decoded bytes, the output folded into the content checksum.  *consumed = bytes of src
 * used; *end_mark = 1 when the end mark (size word 0) was reached and consumed -- the content
 * checksum word that may follow it is the caller's to read and compare with
 * lz4hip_stream_checksum().  block_lens / n_blocks (may be NULL): decoded length per block.
 * Returns the decoded bytes or a negative LZ4HIP_ERR_* (the first failing block in stream
 * order decides, lz4stream/block.go:187-194). */
int64_t lz4hip_stream_uncompress_blocks(lz4hip_stream *s, const uint8_t *src, size_t n, uint8_t *dst,
                                        size_t dst_cap, size_t *consumed, int *end_mark,
                                        uint32_t *block_lens, size_t *n_blocks);

#ifdef __cplusplus
}
#endif
#endif /* LZ4HIP_H */
