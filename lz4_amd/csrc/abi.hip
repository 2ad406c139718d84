// abi.hip -- the extern "C" surface of liblz4hip.so (see include/lz4hip.h).
//
// Device entry points validate shapes on the host before any launch (a faulting
// kernel can reset the whole node) and are asynchronous.  Host-pointer entry
// points own their device buffers for the duration of the call.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "common.h"

using namespace lz4hip;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}
int hip_fail(hipError_t e, const char *what) {
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return LZ4HIP_ERR_HIP;
}
#define HIPCHK(call)                                      \
    do {                                                  \
        hipError_t e_ = (call);                           \
        if (e_ != hipSuccess) return hip_fail(e_, #call); \
    } while (0)

bool valid_block_size(uint32_t bs) {
    return bs == LZ4HIP_BLOCK_64KB || bs == LZ4HIP_BLOCK_256KB || bs == LZ4HIP_BLOCK_1MB ||
           bs == LZ4HIP_BLOCK_4MB;
}
int block_size_index(uint32_t bs) {  // blocks.go:22-36
    switch (bs) {
    case LZ4HIP_BLOCK_64KB: return 4;
    case LZ4HIP_BLOCK_256KB: return 5;
    case LZ4HIP_BLOCK_1MB: return 6;
    case LZ4HIP_BLOCK_4MB: return 7;
    }
    return 0;
}
bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

constexpr size_t kFastSlots = 8192;             // resident waves of the global-table Fast kernel
constexpr size_t kFastSlotBytes = 65536 * 2;    // one u16 table
constexpr size_t kFastCounterBytes = 256;       // tail of the Fast workspace: the block counter of the dynamic schedule
constexpr size_t kFastGlobalMinBlocks = 257;    // up to one block per CU the LDS-table kernel is used
constexpr size_t kHcSlots = 4096;               // resident workgroups of the HC kernel
constexpr uint32_t kLegacyBlock = 8u << 20;      // blocks.go:11, frame.go:49
constexpr size_t kMaxSingleBlock = 1u << 30;    // int32 positions inside a block

// Host xxh32 for the 2/10-byte frame descriptor only (frame.go:202-204).
uint32_t rol(uint32_t u, int k) { return (u << k) | (u >> (32 - k)); }
uint32_t host_xxh32_short(const uint8_t *p, size_t n) {  // n < 16
    uint32_t h = (uint32_t)n + 374761393u;
    size_t i = 0;
    for (; i + 4 <= n; i += 4) {
        uint32_t w;
        memcpy(&w, p + i, 4);
        h = rol(h + w * 3266489917u, 17) * 668265263u;
    }
    for (; i < n; i++) h = rol(h + p[i] * 374761393u, 11) * 2654435761u;
    h ^= h >> 15;
    h *= 2246822519u;
    h ^= h >> 13;
    h *= 3266489917u;
    h ^= h >> 16;
    return h;
}

// Block count from which the global-table Fast kernel is used: up to one block per CU the
// LDS-table kernel finishes each block sooner (k_fast3).
// A Fast workspace is `slots` hash tables followed, when the caller sized it with
// lz4hip_fast_workspace_bytes, by the block counter of the dynamic schedule.  A workspace
// without room for the counter still works: blocks are then dealt round-robin.
uint32_t *fast_ws_split(void *ws, size_t bytes, size_t &slots) {
    slots = ws ? bytes / kFastSlotBytes : 0;
    if (slots > kFastSlots) slots = kFastSlots;
    if (!slots || bytes - slots * kFastSlotBytes < sizeof(uint32_t)) return nullptr;
    return reinterpret_cast<uint32_t *>(static_cast<uint8_t *>(ws) + slots * kFastSlotBytes);
}

// The host-pointer entry points run on the calling thread's own stream
// (hipStreamPerThread), never on the legacy null stream: calls made from several OS
// threads (cgo runs each goroutine's call on its own thread) then overlap on the GPU
// instead of queueing behind each other.  hcopy = a blocking copy on that stream.
const hipStream_t kHostStream = hipStreamPerThread;
// Piece size of the source upload in lz4hip_frame_compress: a multiple of every block
// size (the legacy 8 MiB included) and of XXH32's 16-byte stripe.
constexpr size_t kUploadPiece = 64u << 20;
hipError_t hcopy(void *dst, const void *src, size_t n, hipMemcpyKind kind) {
    hipError_t e = hipMemcpyAsync(dst, src, n, kind, kHostStream);
    if (e != hipSuccess) return e;
    return hipStreamSynchronize(kHostStream);
}

// RAII device buffer for the host-pointer entry points: stream-ordered allocation on the
// calling thread's stream.  hipMalloc/hipFree would do, but hipFree waits for EVERY stream
// of the device, which serialises concurrent callers behind each other's kernels.
// Every use of the buffer on another stream is synchronised before the owner returns.
struct DevBuf {
    void *p = nullptr;
    hipError_t alloc(size_t n) { return hipMallocAsync(&p, n ? n : 16, kHostStream); }
    ~DevBuf() {
        if (p) (void)hipFreeAsync(p, kHostStream);
    }
    template <class T> T *as() { return static_cast<T *>(p); }
};

// A second, non-blocking stream owned by one call (the content-checksum walk).  The
// destructor waits for it, so buffers declared BEFORE it outlive everything queued on it.
struct SideStream {
    hipStream_t s = nullptr;
    hipError_t create() { return hipStreamCreateWithFlags(&s, hipStreamNonBlocking); }
    ~SideStream() {
        if (s) {
            (void)hipStreamSynchronize(s);
            (void)hipStreamDestroy(s);
        }
    }
};

}  // namespace

extern "C" {

const char *lz4hip_last_error(void) { return g_err.c_str(); }
const char *lz4hip_version(void) { return "lz4hip 0.1 (gfx950)"; }

int lz4hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

size_t lz4hip_compress_bound(size_t n) { return n + n / 255 + 16; }

size_t lz4hip_fast_workspace_bytes(size_t n_blocks) {
    if (n_blocks < kFastGlobalMinBlocks) return 0;  // LDS-table kernel, no workspace needed
    const size_t want = 3072;  // 12 waves per CU on 256 CUs: the measured plateau
    return (n_blocks < want ? n_blocks : want) * kFastSlotBytes + kFastCounterBytes;
}

int lz4hip_compress_fast_blocks(const void *d_src, size_t n, uint32_t block_size, int dst_mode,
                                void *d_dst, size_t dst_stride, int32_t *d_result,
                                void *d_workspace, size_t workspace_bytes, void *stream) {
    if (!valid_block_size(block_size)) return fail(LZ4HIP_ERR_ARG, "invalid block size");
    if (dst_mode != LZ4HIP_DST_FRAME && dst_mode != LZ4HIP_DST_BOUND)
        return fail(LZ4HIP_ERR_ARG, "invalid dst_mode");
    if (n == 0) return LZ4HIP_OK;
    if (!d_src || !d_dst || !d_result) return fail(LZ4HIP_ERR_ARG, "null pointer");
    if (!aligned16(d_src) || !aligned16(d_dst) || (dst_stride & 15))
        return fail(LZ4HIP_ERR_ARG, "d_src, d_dst and dst_stride must be 16-byte aligned");
    const size_t need =
        dst_mode == LZ4HIP_DST_FRAME ? block_size : lz4hip_compress_bound(block_size);
    if (dst_stride < need) return fail(LZ4HIP_ERR_ARG, "dst_stride smaller than dst_len");
    const size_t nb = (n + block_size - 1) / block_size;
    size_t slots = 0;
    uint32_t *next_block = fast_ws_split(d_workspace, workspace_bytes, slots);
    if (d_workspace && (!aligned16(d_workspace) || slots == 0))
        return fail(LZ4HIP_ERR_ARG, "workspace unaligned or smaller than one table");
    // Few blocks: one block per CU with the table in LDS finishes each block sooner.
    // Many blocks: global tables and many waves per CU give the higher throughput.
    if (nb < kFastGlobalMinBlocks) slots = 0;
    HIPCHK(launch_fast((const uint8_t *)d_src, n, block_size, nb, dst_mode, 0, (uint8_t *)d_dst,
                       dst_stride, d_result, (uint16_t *)d_workspace, slots, next_block,
                       (hipStream_t)stream));
    return LZ4HIP_OK;
}

size_t lz4hip_hc_workspace_bytes(size_t n_blocks) {
    // the last 256 bytes hold the block counter of the dynamic schedule
    return (n_blocks < kHcSlots ? (n_blocks ? n_blocks : 1) : kHcSlots) * kHcSlotBytes + kFastCounterBytes;
}

int lz4hip_compress_hc_blocks(const void *d_src, size_t n, uint32_t block_size, uint32_t depth,
                              int dst_mode, void *d_dst, size_t dst_stride, int32_t *d_result,
                              void *d_workspace, size_t workspace_bytes, void *stream) {
    if (!valid_block_size(block_size)) return fail(LZ4HIP_ERR_ARG, "invalid block size");
    if (dst_mode != LZ4HIP_DST_FRAME && dst_mode != LZ4HIP_DST_BOUND)
        return fail(LZ4HIP_ERR_ARG, "invalid dst_mode");
    if (n == 0) return LZ4HIP_OK;
    if (!d_src || !d_dst || !d_result || !d_workspace) return fail(LZ4HIP_ERR_ARG, "null pointer");
    if (!aligned16(d_src) || !aligned16(d_dst) || (dst_stride & 15) || !aligned16(d_workspace))
        return fail(LZ4HIP_ERR_ARG, "pointers and dst_stride must be 16-byte aligned");
    const size_t need =
        dst_mode == LZ4HIP_DST_FRAME ? block_size : lz4hip_compress_bound(block_size);
    if (dst_stride < need) return fail(LZ4HIP_ERR_ARG, "dst_stride smaller than dst_len");
    const size_t nb = (n + block_size - 1) / block_size;
    size_t slots = workspace_bytes / kHcSlotBytes;
    if (slots > kHcSlots) slots = kHcSlots;
    if (slots == 0 || (slots < nb && slots < 64))
        return fail(LZ4HIP_ERR_ARG, "workspace smaller than lz4hip_hc_workspace_bytes(n_blocks)");
    // room behind the tables for the counter: dynamic schedule; else round-robin
    uint32_t *next_block =
        workspace_bytes - slots * kHcSlotBytes >= sizeof(uint32_t)
            ? reinterpret_cast<uint32_t *>(static_cast<uint8_t *>(d_workspace) + slots * kHcSlotBytes)
            : nullptr;
    HIPCHK(launch_hc((const uint8_t *)d_src, n, block_size, nb, depth, dst_mode, 0,
                     (uint8_t *)d_dst, dst_stride, d_result, (uint32_t *)d_workspace, slots,
                     next_block, (hipStream_t)stream));
    return LZ4HIP_OK;
}

int lz4hip_uncompress_blocks(const void *d_src, const uint64_t *d_src_off,
                             const uint32_t *d_size_words, size_t n_blocks, void *d_dst,
                             size_t dst_stride, uint32_t dst_len, int32_t *d_out_len,
                             void *stream) {
    if (n_blocks == 0) return LZ4HIP_OK;
    if (!d_src || !d_src_off || !d_size_words || !d_dst || !d_out_len)
        return fail(LZ4HIP_ERR_ARG, "null pointer");
    if (dst_stride < dst_len) return fail(LZ4HIP_ERR_ARG, "dst_stride smaller than dst_len");
    if (dst_len > kMaxSingleBlock) return fail(LZ4HIP_ERR_ARG, "dst_len too large");
    if (n_blocks > 0x7fffffffu) return fail(LZ4HIP_ERR_ARG, "too many blocks");
    HIPCHK(launch_decode((const uint8_t *)d_src, d_src_off, d_size_words, n_blocks,
                         (uint8_t *)d_dst, dst_stride, dst_len, 0, d_out_len,
                         (hipStream_t)stream));
    return LZ4HIP_OK;
}

int lz4hip_xxh32_blocks(const void *d_base, size_t n, size_t stride, const uint64_t *d_off,
                        const int32_t *d_len, size_t n_spans, uint32_t *d_out, void *stream) {
    if (n_spans == 0) return LZ4HIP_OK;
    if (!d_base || !d_out) return fail(LZ4HIP_ERR_ARG, "null pointer");
    if (!d_off && stride == 0) return fail(LZ4HIP_ERR_ARG, "stride is 0 and no offsets given");
    HIPCHK(launch_xxh32_blocks((const uint8_t *)d_base, n, stride, d_off, d_len, n_spans, d_out,
                               (hipStream_t)stream));
    return LZ4HIP_OK;
}

int lz4hip_xxh32_stream(const void *d_src, size_t n, uint32_t *d_out, void *stream) {
    if (!d_out || (!d_src && n)) return fail(LZ4HIP_ERR_ARG, "null pointer");
    HIPCHK(launch_xxh32_stream((const uint8_t *)d_src, n, d_out, (hipStream_t)stream));
    return LZ4HIP_OK;
}

int lz4hip_xxh32_stream_update(const void *d_src, size_t n, uint32_t *d_state, int first,
                               int final_piece, uint32_t *d_digest, void *stream) {
    if (!d_state || (!d_src && n) || (final_piece && !d_digest))
        return fail(LZ4HIP_ERR_ARG, "null pointer");
    if (!final_piece && (n & 15))
        return fail(LZ4HIP_ERR_ARG, "a non-final piece must be a multiple of 16 bytes");
    HIPCHK(launch_xxh32_update((const uint8_t *)d_src, n, d_state, first, final_piece, d_digest,
                               (hipStream_t)stream));
    return LZ4HIP_OK;
}

int lz4hip_frame_layout(const int32_t *d_result, size_t n, uint32_t block_size, size_t n_blocks,
                        const uint32_t *d_block_xxh, uint64_t first_off, uint64_t *d_entry_off,
                        uint32_t *d_size_words, void *stream) {
    if (!valid_block_size(block_size)) return fail(LZ4HIP_ERR_ARG, "invalid block size");
    if (!d_entry_off || (!d_result && n_blocks)) return fail(LZ4HIP_ERR_ARG, "null pointer");
    HIPCHK(launch_frame_layout(d_result, n, block_size, n_blocks, d_block_xxh, first_off,
                               d_entry_off, d_size_words, (hipStream_t)stream));
    return LZ4HIP_OK;
}

int lz4hip_frame_scatter(const void *d_src, size_t n, uint32_t block_size, const void *d_comp,
                         size_t comp_stride, const int32_t *d_result,
                         const uint32_t *d_block_xxh, const uint64_t *d_entry_off,
                         size_t n_blocks, void *d_frame, uint64_t frame_base, void *stream) {
    if (!valid_block_size(block_size)) return fail(LZ4HIP_ERR_ARG, "invalid block size");
    if (n_blocks == 0) return LZ4HIP_OK;
    if (!d_result || !d_entry_off || !d_frame || (n && (!d_src || !d_comp)))
        return fail(LZ4HIP_ERR_ARG, "null pointer");
    HIPCHK(launch_frame_scatter((const uint8_t *)d_src, n, block_size, (const uint8_t *)d_comp,
                                comp_stride, d_result, d_block_xxh, d_entry_off, n_blocks,
                                (uint8_t *)d_frame, frame_base, (hipStream_t)stream));
    return LZ4HIP_OK;
}

// frame.go:144-167,202-204; frame_gen.go:7-29
size_t lz4hip_frame_header(uint8_t *dst, uint32_t block_size, uint32_t flags,
                           uint64_t content_size) {
    uint16_t f = (1u << 6) | (1u << 5);  // version 1, independent blocks: frame.go:139-142
    if (flags & LZ4HIP_F_BLOCK_CHECKSUM) f |= 1u << 4;
    if (flags & LZ4HIP_F_CONTENT_SIZE) f |= 1u << 3;
    if (flags & LZ4HIP_F_CONTENT_CHECKSUM) f |= 1u << 2;
    f |= (uint16_t)(block_size_index(block_size) & 7) << 12;
    const uint32_t magic = 0x184D2204u;
    memcpy(dst, &magic, 4);
    size_t k = 4;
    dst[k++] = (uint8_t)f;
    dst[k++] = (uint8_t)(f >> 8);
    if (flags & LZ4HIP_F_CONTENT_SIZE) {
        memcpy(dst + k, &content_size, 8);
        k += 8;
    }
    dst[k] = (uint8_t)(host_xxh32_short(dst + 4, k - 4) >> 8);
    return k + 1;
}

// frame.go:59-74
size_t lz4hip_frame_trailer(uint8_t *dst, uint32_t flags, uint32_t content_xxh) {
    memset(dst, 0, 4);
    if (!(flags & LZ4HIP_F_CONTENT_CHECKSUM)) return 4;
    memcpy(dst + 4, &content_xxh, 4);
    return 8;
}

size_t lz4hip_frame_bound(size_t n, uint32_t block_size, uint32_t flags) {
    if (flags & LZ4HIP_F_LEGACY) block_size = kLegacyBlock;
    const size_t bs = block_size ? block_size : 1;
    const size_t nb = n / bs + 2;
    // a legacy short last block is compressed into an 8 MiB dst and may grow to
    // CompressBlockBound(len) instead of being stored raw
    return 15 + n + nb * 8 + 8 + ((flags & LZ4HIP_F_LEGACY) ? n % bs / 255 + 16 : 0);
}

/* ---------------- host-pointer entry points ---------------- */

static int64_t compress_one(const uint8_t *src, size_t n, uint8_t *dst, size_t dst_len,
                            bool hc, uint32_t depth) {
    if (n > kMaxSingleBlock || dst_len > 0x7fffffffu)
        return fail(LZ4HIP_ERR_ARG, "block larger than 1 GiB");
    if ((n && !src) || (dst_len && !dst)) return fail(LZ4HIP_ERR_ARG, "null pointer");
    DevBuf d_src, d_dst, d_res, d_ws;
    HIPCHK(d_src.alloc(n + 16));
    HIPCHK(d_dst.alloc(dst_len + 16));
    HIPCHK(d_res.alloc(sizeof(int32_t)));
    if (n) HIPCHK(hcopy(d_src.p, src, n, hipMemcpyHostToDevice));
    // one block that spans the whole input: cut size = max(n, 1)
    const uint32_t bs = (uint32_t)(n ? n : 1);
    if (hc) {
        HIPCHK(d_ws.alloc(kHcSlotBytes));
        HIPCHK(launch_hc(d_src.as<uint8_t>(), n, bs, 1, depth, kDstExplicit, (uint32_t)dst_len,
                         d_dst.as<uint8_t>(), 0, d_res.as<int32_t>(), d_ws.as<uint32_t>(), 1,
                         nullptr, kHostStream));
    } else {
        HIPCHK(launch_fast(d_src.as<uint8_t>(), n, bs, 1, kDstExplicit, (uint32_t)dst_len,
                           d_dst.as<uint8_t>(), 0, d_res.as<int32_t>(), nullptr, 0, nullptr,
                           kHostStream));
    }
    int32_t r = 0;
    HIPCHK(hcopy(&r, d_res.p, sizeof r, hipMemcpyDeviceToHost));
    if (r > 0) HIPCHK(hcopy(dst, d_dst.p, (size_t)r, hipMemcpyDeviceToHost));
    return r;  // n, 0 or -1: the Go conventions of lz4.go:59-69
}

int64_t lz4hip_compress_block(const uint8_t *src, size_t n, uint8_t *dst, size_t dst_len) {
    int64_t r = compress_one(src, n, dst, dst_len, false, 0);
    return r < -1 ? -1 : r;
}

int64_t lz4hip_compress_block_hc(const uint8_t *src, size_t n, uint8_t *dst, size_t dst_len,
                                 uint32_t depth) {
    int64_t r = compress_one(src, n, dst, dst_len, true, depth);
    return r < -1 ? -1 : r;
}

static int64_t uncompress_one(const uint8_t *src, size_t n, uint8_t *dst, size_t dst_len,
                              const uint8_t *dict, size_t dict_len) {
    if (n == 0) return 0;  // block.go:45-47
    if (n > 0x7fffffffu || dst_len > kMaxSingleBlock)
        return fail(LZ4HIP_ERR_ARG, "block larger than 1 GiB");
    if (!src || (dst_len && !dst) || (dict_len && !dict))
        return fail(LZ4HIP_ERR_ARG, "null pointer");
    // Only the last 64 KiB of a dictionary are reachable (offsets are 16-bit).
    if (dict_len > 65536) {
        dict += dict_len - 65536;
        dict_len = 65536;
    }
    DevBuf d_src, d_out, d_meta;
    HIPCHK(d_src.alloc(n));
    HIPCHK(d_out.alloc(dict_len + dst_len));  // [dictionary | dst]
    HIPCHK(d_meta.alloc(32));
    HIPCHK(hcopy(d_src.p, src, n, hipMemcpyHostToDevice));
    if (dict_len) HIPCHK(hcopy(d_out.p, dict, dict_len, hipMemcpyHostToDevice));
    struct {
        uint64_t off;
        uint32_t sw;
        int32_t out;
    } meta = {0, (uint32_t)n, 0};
    HIPCHK(hcopy(d_meta.p, &meta, sizeof meta, hipMemcpyHostToDevice));
    uint8_t *m = d_meta.as<uint8_t>();
    HIPCHK(launch_decode(d_src.as<uint8_t>(), (const uint64_t *)m, (const uint32_t *)(m + 8), 1,
                         d_out.as<uint8_t>() + dict_len, 0, (uint32_t)dst_len,
                         (uint32_t)dict_len, (int32_t *)(m + 12), kHostStream));
    HIPCHK(hcopy(&meta, d_meta.p, sizeof meta, hipMemcpyDeviceToHost));
    if (meta.out > 0)
        HIPCHK(hcopy(dst, d_out.as<uint8_t>() + dict_len, (size_t)meta.out,
                         hipMemcpyDeviceToHost));
    return meta.out;
}

int64_t lz4hip_uncompress_block(const uint8_t *src, size_t n, uint8_t *dst, size_t dst_len,
                                const uint8_t *dict, size_t dict_len) {
    int64_t r = uncompress_one(src, n, dst, dst_len, dict, dict_len);
    return r < -1 ? -1 : r;
}

uint32_t lz4hip_xxh32(const uint8_t *p, size_t n) {
    DevBuf d_src, d_out;
    if (d_src.alloc(n + 16) != hipSuccess || d_out.alloc(4) != hipSuccess) {
        g_err = "hipMalloc failed";
        return 0;
    }
    if (n && hcopy(d_src.p, p, n, hipMemcpyHostToDevice) != hipSuccess) {
        g_err = "hipMemcpy failed";
        return 0;
    }
    const int32_t len = (int32_t)n;
    uint32_t out = 0;
    if (n <= 0x7fffffffu) {
        DevBuf d_len;
        if (d_len.alloc(4) != hipSuccess) return 0;
        (void)hcopy(d_len.p, &len, 4, hipMemcpyHostToDevice);
        (void)launch_xxh32_blocks(d_src.as<uint8_t>(), n, n ? n : 1, nullptr, d_len.as<int32_t>(),
                                  1, d_out.as<uint32_t>(), kHostStream);
        (void)hcopy(&out, d_out.p, 4, hipMemcpyDeviceToHost);
    } else {
        (void)launch_xxh32_stream(d_src.as<uint8_t>(), n, d_out.as<uint32_t>(), kHostStream);
        (void)hcopy(&out, d_out.p, 4, hipMemcpyDeviceToHost);
    }
    return out;
}

static int64_t frame_compress(const uint8_t *src, size_t n, uint32_t block_size, uint32_t flags,
                              uint32_t level, uint64_t content_size, uint8_t *dst, size_t dst_cap,
                              std::vector<uint32_t> *lens) {
    const bool legacy = flags & LZ4HIP_F_LEGACY;
    if (legacy) block_size = kLegacyBlock;  // frame.go:47-50
    else if (!valid_block_size(block_size)) return fail(LZ4HIP_ERR_BLOCK_SIZE, "invalid block size");
    if ((n && !src) || !dst) return fail(LZ4HIP_ERR_ARG, "null pointer");
    if (dst_cap < lz4hip_frame_bound(n, block_size, flags))
        return fail(LZ4HIP_ERR_SHORT, "dst_cap < lz4hip_frame_bound()");
    const bool bsum = flags & LZ4HIP_F_BLOCK_CHECKSUM;
    const bool csum = (flags & LZ4HIP_F_CONTENT_CHECKSUM) && !legacy;  // frame.go:63-65
    const size_t nb_data = (n + block_size - 1) / block_size;
    // Close() flushes only pending bytes; ReadFrom writes data[:rn] even when rn == 0
    // (writer.go:161-167 vs :220-233).  A standard frame stores that empty block raw
    // (dst = data[:0]); a legacy frame compresses it into the 8 MiB dst, which yields
    // the one-byte block 00 (lz4stream/block.go:226-227, block.go:258-292).
    const bool empty_tail = (flags & LZ4HIP_F_READFROM) && n % block_size == 0;
    const size_t nb = nb_data + ((empty_tail && !legacy) ? 1 : 0);
    // dst_len of every block: len(src) in a standard frame, cap(data) = 8 MiB in a legacy one
    const int dmode = legacy ? kDstExplicit : LZ4HIP_DST_FRAME;
    const uint32_t dlen_x = legacy ? kLegacyBlock : 0;

    uint8_t hdr[16], trl[12];
    size_t hlen;
    if (legacy) {  // magic only: frame.go:150-165
        const uint32_t m = 0x184C2102u;
        memcpy(hdr, &m, 4);
        hlen = 4;
    } else {
        hlen = lz4hip_frame_header(hdr, block_size, flags, content_size);
    }

    DevBuf d_src, d_comp, d_res, d_xxh, d_off, d_frame, d_cs, d_state, d_fws;
    hipStream_t s_main = nullptr, s_sum = nullptr;
    HIPCHK(d_src.alloc(n + 16));
    HIPCHK(d_state.alloc(6 * sizeof(uint32_t)));
    HIPCHK(d_comp.alloc(nb_data * (size_t)block_size));
    HIPCHK(d_res.alloc((nb + 1) * sizeof(int32_t)));
    HIPCHK(d_xxh.alloc((nb + 1) * sizeof(uint32_t)));
    HIPCHK(d_off.alloc((nb + 1) * sizeof(uint64_t)));
    HIPCHK(d_frame.alloc(dst_cap));
    HIPCHK(d_cs.alloc(4));
    HIPCHK(hipStreamSynchronize(kHostStream));  // the allocations above are ordered on it
    HIPCHK(hipStreamCreateWithFlags(&s_main, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&s_sum, hipStreamNonBlocking));
    int rc = LZ4HIP_OK;
    int64_t total = 0;
    do {
#define STEP(call)                        \
    {                                     \
        hipError_t e_ = (call);           \
        if (e_ != hipSuccess) {           \
            rc = hip_fail(e_, #call);     \
            break;                        \
        }                                 \
    }
        // The content checksum is one serial walk (about 2.4 GB/s) and bounds the call, so it
        // starts as early as it can: the source goes up in pieces and the chained checksum
        // (XXHZero.Write per piece, xxh32zero.go:58-86) follows each piece on its own stream
        // while the next one is still on the bus and while the blocks compress.
        {
            hipError_t e = hipSuccess;
            size_t at = 0;
            do {
                const size_t len = std::min(kUploadPiece, n - at);
                if (len) e = hcopy(d_src.as<uint8_t>() + at, src + at, len, hipMemcpyHostToDevice);
                if (e == hipSuccess && csum)
                    e = launch_xxh32_update(d_src.as<uint8_t>() + at, len, d_state.as<uint32_t>(),
                                            at == 0, at + len >= n, d_cs.as<uint32_t>(), s_sum);
                at += len;
            } while (e == hipSuccess && at < n);
            STEP(e);
        }
        STEP(hipMemsetAsync(d_res.p, 0, (nb + 1) * sizeof(int32_t), s_main));
        if (nb_data) {
            if (level == 0) {
                const size_t wsb = lz4hip_fast_workspace_bytes(nb_data);
                if (wsb) {
                    STEP(d_fws.alloc(wsb));
                    STEP(hipStreamSynchronize(kHostStream));  // allocated on it, used on s_main
                }
                size_t slots = 0;
                uint32_t *next_block = fast_ws_split(d_fws.p, wsb, slots);
                STEP(launch_fast(d_src.as<uint8_t>(), n, block_size, nb_data, dmode, dlen_x,
                                 d_comp.as<uint8_t>(), block_size, d_res.as<int32_t>(),
                                 d_fws.as<uint16_t>(), slots, next_block, s_main));
            } else {
                DevBuf d_ws;
                const size_t slots = nb_data < kHcSlots ? nb_data : kHcSlots;
                STEP(d_ws.alloc(slots * kHcSlotBytes + kFastCounterBytes));
                STEP(hipStreamSynchronize(kHostStream));  // allocated on it, used on s_main
                STEP(launch_hc(d_src.as<uint8_t>(), n, block_size, nb_data, level,
                               dmode, dlen_x, d_comp.as<uint8_t>(), block_size,
                               d_res.as<int32_t>(), d_ws.as<uint32_t>(), slots,
                               reinterpret_cast<uint32_t *>(d_ws.as<uint8_t>() + slots * kHcSlotBytes),
                               s_main));
                STEP(hipStreamSynchronize(s_main));  // d_ws dies with this scope
            }
        }
        if (bsum)
            STEP(launch_xxh32_blocks(d_src.as<uint8_t>(), n, block_size, nullptr, nullptr, nb,
                                     d_xxh.as<uint32_t>(), s_main));
        STEP(launch_frame_layout(d_res.as<int32_t>(), n, block_size, nb,
                                 bsum ? d_xxh.as<uint32_t>() : nullptr, hlen,
                                 d_off.as<uint64_t>(), nullptr, s_main));
        STEP(launch_frame_scatter(d_src.as<uint8_t>(), n, block_size, d_comp.as<uint8_t>(),
                                  block_size, d_res.as<int32_t>(),
                                  bsum ? d_xxh.as<uint32_t>() : nullptr, d_off.as<uint64_t>(), nb,
                                  d_frame.as<uint8_t>(), 0, s_main));
        uint64_t end_off = 0;
        STEP(hipMemcpyAsync(&end_off, d_off.as<uint64_t>() + nb, 8, hipMemcpyDeviceToHost, s_main));
        STEP(hipStreamSynchronize(s_main));
        if (end_off > dst_cap) {
            rc = fail(LZ4HIP_ERR_SHORT, "frame larger than dst_cap");
            break;
        }
        memcpy(dst, hdr, hlen);  // the blocks go down while the checksum walk finishes
        if (end_off > hlen)
            STEP(hcopy(dst + hlen, d_frame.as<uint8_t>() + hlen, end_off - hlen,
                           hipMemcpyDeviceToHost));
        STEP(hipStreamSynchronize(s_sum));
        uint32_t cs = 0;
        if (csum) STEP(hcopy(&cs, d_cs.p, 4, hipMemcpyDeviceToHost));
        size_t tlen;
        if (legacy) {  // no end mark (frame.go:63-65); ReadFrom's empty block: size 1, token 00
            tlen = 0;
            if (empty_tail) {
                const uint8_t one[5] = {1, 0, 0, 0, 0};
                memcpy(trl, one, 5);
                tlen = 5;
                if (bsum) {  // xxh32("") = 0x02CC5D05 follows even here (lz4stream/block.go:248-250)
                    const uint32_t e = 0x02CC5D05u;
                    memcpy(trl + 5, &e, 4);
                    tlen = 9;
                }
            }
        } else {
            tlen = lz4hip_frame_trailer(trl, flags, cs);
        }
        if (end_off + tlen > dst_cap) {
            rc = fail(LZ4HIP_ERR_SHORT, "frame larger than dst_cap");
            break;
        }
        memcpy(dst + end_off, trl, tlen);
        total = (int64_t)(end_off + tlen);
        if (lens) {  // len(block.Data) of every block = its size word without the raw bit
            std::vector<uint64_t> off(nb + 1);
            STEP(hcopy(off.data(), d_off.p, (nb + 1) * 8, hipMemcpyDeviceToHost));
            for (size_t b = 0; b < nb; b++) {
                uint32_t w;
                memcpy(&w, dst + off[b], 4);
                lens->push_back(w & 0x7FFFFFFFu);
            }
            if (legacy && empty_tail) lens->push_back(1);
        }
#undef STEP
    } while (0);
    (void)hipStreamDestroy(s_main);
    (void)hipStreamDestroy(s_sum);
    return rc == LZ4HIP_OK ? total : rc;
}

int64_t lz4hip_frame_compress(const uint8_t *src, size_t n, uint32_t block_size, uint32_t flags,
                              uint32_t level, uint64_t content_size, uint8_t *dst,
                              size_t dst_cap) {
    return frame_compress(src, n, block_size, flags, level, content_size, dst, dst_cap, nullptr);
}

int64_t lz4hip_frame_compress_lens(const uint8_t *src, size_t n, uint32_t block_size,
                                   uint32_t flags, uint32_t level, uint64_t content_size,
                                   uint8_t *dst, size_t dst_cap, uint32_t *block_lens,
                                   size_t max_blocks, size_t *n_blocks) {
    std::vector<uint32_t> lens;
    const int64_t k =
        frame_compress(src, n, block_size, flags, level, content_size, dst, dst_cap, &lens);
    if (k < 0) return k;
    if (n_blocks) *n_blocks = lens.size();
    if (block_lens && max_blocks)
        memcpy(block_lens, lens.data(), std::min(max_blocks, lens.size()) * sizeof(uint32_t));
    return k;
}

int lz4hip_frame_parse_header(const uint8_t *src, size_t n, uint32_t *block_size, uint32_t *flags,
                              uint64_t *content_size, size_t *header_len) {
    if (n && !src) return fail(LZ4HIP_ERR_ARG, "null pointer");
    size_t si = 0;
    auto rd32 = [&](size_t at) {
        uint32_t v;
        memcpy(&v, src + at, 4);
        return v;
    };
    uint32_t fl = 0, bs = 0;
    uint64_t cs = 0;
    for (;;) {  // frame.go:80-111
        if (n - si < 4) return fail(LZ4HIP_ERR_SHORT, "unexpected EOF in the frame header");
        const uint32_t magic = rd32(si);
        si += 4;
        if (magic == 0x184D2204u) break;
        if (magic == 0x184C2102u) {  // frame.go:169-174
            fl = LZ4HIP_F_LEGACY;
            bs = kLegacyBlock;
            break;
        }
        if (magic >> 8 == 0x184D2A50u >> 8) {  // skippable frame
            if (n - si < 4) return fail(LZ4HIP_ERR_SHORT, "unexpected EOF in the frame header");
            const uint32_t skip = rd32(si);
            si += 4;
            if (n - si < skip) return fail(LZ4HIP_ERR_SHORT, "unexpected EOF in a skippable frame");
            si += skip;
            continue;
        }
        return fail(LZ4HIP_ERR_FRAME, "invalid frame magic");
    }
    if (!(fl & LZ4HIP_F_LEGACY)) {  // frame.go:175-200
        if (n - si < 3) return fail(LZ4HIP_ERR_SHORT, "unexpected EOF in the frame header");
        const size_t d0 = si;
        const uint16_t f = (uint16_t)(src[si] | (src[si + 1] << 8));
        si += 2;
        if (f >> 3 & 1) {
            if (n - si < 9) return fail(LZ4HIP_ERR_SHORT, "unexpected EOF in the frame header");
            memcpy(&cs, src + si, 8);
            si += 8;
            fl |= LZ4HIP_F_CONTENT_SIZE;
        }
        if ((uint8_t)(host_xxh32_short(src + d0, si - d0) >> 8) != src[si])
            return fail(LZ4HIP_ERR_HEADER_CHECKSUM, "invalid header checksum");
        si += 1;
        const int idx = f >> 12 & 7;
        if (idx < 4) return fail(LZ4HIP_ERR_BLOCK_SIZE, "invalid block size");
        bs = 1u << (8 + 2 * idx);
        if (f >> 4 & 1) fl |= LZ4HIP_F_BLOCK_CHECKSUM;
        if (f >> 2 & 1) fl |= LZ4HIP_F_CONTENT_CHECKSUM;
        if (!(f >> 5 & 1)) fl |= LZ4HIP_F_LINKED;
    }
    if (block_size) *block_size = bs;
    if (flags) *flags = fl;
    if (content_size) *content_size = cs;
    if (header_len) *header_len = si;
    return LZ4HIP_OK;
}

// Legacy frame (magic 0x184C2102, frame.go:20,46-51,169-174): 8 MiB independent
// blocks, no checksums, no end mark.  block.Read's legacy rules
// (lz4stream/block.go:284-294): a size word equal to the magic starts a
// concatenated frame; a size word equal to the bytes decoded so far ends the
// stream (the Linux kernel appends the total size).  That last rule needs the
// decoded sizes, so: index every candidate block on the host, decode them all in
// one batch, then apply the rule in stream order and drop what follows the end.
static int64_t legacy_uncompress(const uint8_t *src, size_t n, size_t si, uint8_t *dst,
                                 size_t dst_cap, std::vector<uint32_t> *lens) {
    const uint32_t cap = 8u << 20;
    auto rd32 = [&](size_t at) {
        uint32_t v;
        memcpy(&v, src + at, 4);
        return v;
    };
    struct Cand {
        size_t word_at;
        uint32_t word;
    };
    std::vector<Cand> cand;
    std::vector<uint64_t> off;
    std::vector<uint32_t> sw;
    int stop = 0;  // why the index ended: 0 = end of input, 1 = oversize word, 2 = truncated payload
    while (n - si >= 4) {
        const uint32_t x = rd32(si);
        if (x == 0x184C2102u) {
            si += 4;
            continue;
        }
        const uint32_t size = x & 0x7FFFFFFFu;
        if (size > cap) {
            cand.push_back({si, x});
            stop = 1;
            break;
        }
        if (n - si - 4 < size) {
            cand.push_back({si, x});
            stop = 2;
            break;
        }
        cand.push_back({si, x});
        off.push_back(si + 4);
        sw.push_back(x);
        si += 4 + (size_t)size;
    }
    const size_t nb = sw.size();
    std::vector<int32_t> len(nb);
    DevBuf d_src, d_off, d_sw, d_out, d_len;
    if (nb) {
        HIPCHK(d_src.alloc(n));
        HIPCHK(d_off.alloc(nb * 8));
        HIPCHK(d_sw.alloc(nb * 4));
        HIPCHK(d_out.alloc(nb * (size_t)cap));
        HIPCHK(d_len.alloc(nb * 4));
        HIPCHK(hcopy(d_src.p, src, n, hipMemcpyHostToDevice));
        HIPCHK(hcopy(d_off.p, off.data(), nb * 8, hipMemcpyHostToDevice));
        HIPCHK(hcopy(d_sw.p, sw.data(), nb * 4, hipMemcpyHostToDevice));
        HIPCHK(launch_decode(d_src.as<uint8_t>(), d_off.as<uint64_t>(), d_sw.as<uint32_t>(), nb,
                                 d_out.as<uint8_t>(), cap, cap, 0, d_len.as<int32_t>(), kHostStream));
        HIPCHK(hcopy(len.data(), d_len.p, nb * 4, hipMemcpyDeviceToHost));
    }
    size_t total = 0;
    for (size_t b = 0; b < cand.size(); b++) {
        if (cand[b].word == (uint32_t)total) return (int64_t)total;  // :289-293
        if (b >= nb)  // the word that ended the index is a real block header
            return stop == 1 ? fail(LZ4HIP_ERR_BLOCK_SIZE, "legacy block larger than 8 MiB")
                             : fail(LZ4HIP_ERR_FRAME, "truncated frame");
        if (len[b] < 0) return fail(LZ4HIP_ERR_BLOCK, "invalid source or destination buffer too short");
        if (total + (size_t)len[b] > dst_cap) return fail(LZ4HIP_ERR_SHORT, "dst_cap too small");
        if (len[b])
            HIPCHK(hcopy(dst + total, d_out.as<uint8_t>() + b * (size_t)cap, (size_t)len[b],
                             hipMemcpyDeviceToHost));
        total += (size_t)len[b];
        if (lens) lens->push_back((uint32_t)len[b]);
    }
    return (int64_t)total;
}

static int64_t frame_uncompress(const uint8_t *src, size_t n, uint8_t *dst, size_t dst_cap,
                                std::vector<uint32_t> *lens) {
    if (!src || (dst_cap && !dst)) return fail(LZ4HIP_ERR_ARG, "null pointer");
    size_t si = 0;
    auto rd32 = [&](size_t at) {
        uint32_t v;
        memcpy(&v, src + at, 4);
        return v;
    };
    // ---- headers: frame.go:80-111,169-200 (host; a few bytes) ----
    uint32_t bs = 0, hflags = 0;
    {
        const int rc = lz4hip_frame_parse_header(src, n, &bs, &hflags, nullptr, &si);
        if (rc == LZ4HIP_ERR_SHORT) return fail(LZ4HIP_ERR_FRAME, "truncated frame");
        if (rc != LZ4HIP_OK) return rc;
    }
    if (hflags & LZ4HIP_F_LEGACY) return legacy_uncompress(src, n, si, dst, dst_cap, lens);
    const bool bsum = hflags & LZ4HIP_F_BLOCK_CHECKSUM, csum = hflags & LZ4HIP_F_CONTENT_CHECKSUM,
               independent = !(hflags & LZ4HIP_F_LINKED);

    // ---- block index: lz4stream/block.go:279-317 (host walk over size words) ----
    std::vector<uint64_t> off;
    std::vector<uint32_t> sw, want;
    for (;;) {
        if (n - si < 4) return fail(LZ4HIP_ERR_FRAME, "truncated frame");
        const uint32_t w = rd32(si);
        si += 4;
        if (w == 0) break;
        const uint32_t size = w & 0x7FFFFFFFu;
        if (size > bs) return fail(LZ4HIP_ERR_BLOCK_SIZE, "block larger than the frame's block size");
        if (n - si < size) return fail(LZ4HIP_ERR_FRAME, "truncated frame");
        off.push_back(si);
        sw.push_back(w);
        si += size;
        if (bsum) {
            if (n - si < 4) return fail(LZ4HIP_ERR_FRAME, "truncated frame");
            want.push_back(rd32(si));
            si += 4;
        }
    }
    uint32_t want_cs = 0;
    if (csum) {
        if (n - si < 4) return fail(LZ4HIP_ERR_FRAME, "truncated frame");
        want_cs = rd32(si);
    }
    const size_t nb = sw.size();
    if (nb == 0) {
        if (csum && want_cs != 0x02CC5D05u)
            return fail(LZ4HIP_ERR_FRAME_CHECKSUM, "invalid frame checksum");
        return 0;
    }

    // Device memory is bounded by the caller's dst_cap, not by n_blocks * block_size: a
    // frame of many short blocks (a Writer flushed often) must not ask for n_blocks full
    // slots.  Independent blocks decode a chunk at a time into <= 1 GiB of fixed-stride
    // slots and are packed into the contiguous stream; linked blocks decode straight
    // into it.  Errors keep stream order (lz4stream/block.go:187-194): chunks run in
    // order and each chunk is checked front to back before the next starts.
    DevBuf d_src, d_off, d_sw, d_slots, d_flat, d_len, d_xxh, d_cs, d_foff, d_state;
    // stream buffer: no more than the blocks can decode to (a compressed block grows at
    // most 255-fold, a stored one not at all), no more than the caller can take, plus one
    // block of slack so that only the final check decides "dst_cap too small"
    size_t need = 0;
    for (size_t b = 0; b < nb; b++) {
        const size_t sz = sw[b] & 0x7FFFFFFFu;
        const size_t grow = (sw[b] >> 31) ? sz : sz * 255;
        need += grow < bs ? grow : bs;
    }
    const size_t flat_cap = (need < dst_cap ? need : dst_cap) + (size_t)bs;
    const size_t chunk = independent ? std::max<size_t>(1, std::min<size_t>(nb, (1ull << 30) / bs)) : nb;
    HIPCHK(d_src.alloc(n));
    HIPCHK(d_off.alloc(nb * 8));
    HIPCHK(d_sw.alloc(nb * 4));
    HIPCHK(d_flat.alloc(flat_cap + 16));
    HIPCHK(d_len.alloc(chunk * 4));
    HIPCHK(d_xxh.alloc(chunk * 4));
    HIPCHK(d_foff.alloc(chunk * 8));
    HIPCHK(d_cs.alloc(4));
    if (independent) HIPCHK(d_slots.alloc(chunk * (size_t)bs));
    HIPCHK(hcopy(d_src.p, src, n, hipMemcpyHostToDevice));
    HIPCHK(hcopy(d_off.p, off.data(), nb * 8, hipMemcpyHostToDevice));
    HIPCHK(hcopy(d_sw.p, sw.data(), nb * 4, hipMemcpyHostToDevice));
    std::vector<int32_t> len(chunk);
    std::vector<uint32_t> got(chunk);
    std::vector<uint64_t> foff(chunk);
    size_t total = 0, hashed = 0;
    bool started = false;
    SideStream sum;  // declared after the buffers: synchronised and destroyed before they are freed
    if (csum) {
        HIPCHK(d_state.alloc(6 * sizeof(uint32_t)));
        HIPCHK(hipStreamSynchronize(kHostStream));  // the allocations are ordered on it
        HIPCHK(sum.create());
    }
    for (size_t b0 = 0; b0 < nb; b0 += chunk) {
        const size_t c = std::min(chunk, nb - b0);
        if (independent) {
            HIPCHK(launch_decode(d_src.as<uint8_t>(), d_off.as<uint64_t>() + b0,
                                     d_sw.as<uint32_t>() + b0, c, d_slots.as<uint8_t>(), bs, bs, 0,
                                     d_len.as<int32_t>(), kHostStream));
            if (bsum)  // verified on the DECODED bytes: lz4stream/block.go:330-335
                HIPCHK(launch_xxh32_blocks(d_slots.as<uint8_t>(), c * (size_t)bs, bs, nullptr,
                                           d_len.as<int32_t>(), c, d_xxh.as<uint32_t>(), kHostStream));
        } else {  // reader.go:86-90,176-189: dependent blocks decode in order, output packed
            HIPCHK(launch_decode_linked(d_src.as<uint8_t>(), d_off.as<uint64_t>(), d_sw.as<uint32_t>(),
                                         nb, d_flat.as<uint8_t>(), bs, flat_cap, d_len.as<int32_t>(),
                                         d_foff.as<uint64_t>(), kHostStream));
            if (bsum)
                HIPCHK(launch_xxh32_blocks(d_flat.as<uint8_t>(), flat_cap, bs, d_foff.as<uint64_t>(),
                                           d_len.as<int32_t>(), nb, d_xxh.as<uint32_t>(), kHostStream));
        }
        HIPCHK(hcopy(len.data(), d_len.p, c * 4, hipMemcpyDeviceToHost));
        if (bsum) HIPCHK(hcopy(got.data(), d_xxh.p, c * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < c; i++) {
            if (len[i] < 0) {
                // a linked block that found no room left in the stream buffer
                if (!independent && total + bs > flat_cap)
                    return fail(LZ4HIP_ERR_SHORT, "dst_cap too small");
                return fail(LZ4HIP_ERR_BLOCK, "invalid source or destination buffer too short");
            }
            if (bsum && got[i] != want[b0 + i])
                return fail(LZ4HIP_ERR_BLOCK_CHECKSUM, "invalid block checksum");
            foff[i] = total;
            total += (size_t)len[i];
            if (total > dst_cap) return fail(LZ4HIP_ERR_SHORT, "dst_cap too small");
            if (lens) lens->push_back((uint32_t)len[i]);
        }
        if (independent) {
            HIPCHK(hcopy(d_foff.p, foff.data(), c * 8, hipMemcpyHostToDevice));
            HIPCHK(launch_pack_blocks(d_slots.as<uint8_t>(), bs, d_len.as<int32_t>(),
                                      d_foff.as<uint64_t>(), c, d_flat.as<uint8_t>(), kHostStream));
        }
        // frame.go:117-131: the content checksum is one serial walk over the decoded stream
        // and bounds the call, so it follows the chunks on its own stream (XXHZero.Write per
        // piece; pieces before the last are cut to the 16-byte stripe) while the next chunk
        // decodes and while the output goes down to the host.
        if (csum) {
            const bool last = b0 + c >= nb;
            const size_t upto = last ? total : total & ~(size_t)15;
            if (last || upto > hashed) {
                HIPCHK(hipStreamSynchronize(kHostStream));  // chunk packed
                HIPCHK(launch_xxh32_update(d_flat.as<uint8_t>() + hashed, upto - hashed,
                                           d_state.as<uint32_t>(), !started, last,
                                           d_cs.as<uint32_t>(), sum.s));
                hashed = upto;
                started = true;
            }
        }
    }
    if (total) HIPCHK(hcopy(dst, d_flat.p, total, hipMemcpyDeviceToHost));
    if (csum) {
        uint32_t cs = 0;
        HIPCHK(hipStreamSynchronize(sum.s));
        HIPCHK(hcopy(&cs, d_cs.p, 4, hipMemcpyDeviceToHost));
        if (cs != want_cs) return fail(LZ4HIP_ERR_FRAME_CHECKSUM, "invalid frame checksum");
    }
    return (int64_t)total;
}

int64_t lz4hip_frame_uncompress(const uint8_t *src, size_t n, uint8_t *dst, size_t dst_cap) {
    return frame_uncompress(src, n, dst, dst_cap, nullptr);
}

int64_t lz4hip_frame_uncompress_lens(const uint8_t *src, size_t n, uint8_t *dst, size_t dst_cap,
                                     uint32_t *block_lens, size_t max_blocks, size_t *n_blocks) {
    std::vector<uint32_t> lens;
    const int64_t k = frame_uncompress(src, n, dst, dst_cap, &lens);
    if (k < 0) return k;
    if (n_blocks) *n_blocks = lens.size();
    if (block_lens && max_blocks)
        memcpy(block_lens, lens.data(), std::min(max_blocks, lens.size()) * sizeof(uint32_t));
    return k;
}

}  // extern "C"
