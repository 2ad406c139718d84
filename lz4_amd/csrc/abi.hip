// abi.hip -- the extern "C" surface of liblz4hip.so, part 1 (see include/lz4hip.h): the
// device entry points and the small host-only helpers (bound, header, trailer, header
// parse).  The host-pointer entry points and the lz4hip_stream handle are in host_api.hip.
//
// Device entry points validate shapes on the host before any launch (a faulting
// kernel can reset the whole node), are asynchronous and allocate nothing.
#include "abi_common.h"

namespace lz4hip {
thread_local std::string g_err;
}
using namespace lz4hip;

namespace {

int block_size_index(uint32_t bs) {  // blocks.go:22-36
    switch (bs) {
    case LZ4HIP_BLOCK_64KB: return 4;
    case LZ4HIP_BLOCK_256KB: return 5;
    case LZ4HIP_BLOCK_1MB: return 6;
    case LZ4HIP_BLOCK_4MB: return 7;
    }
    return 0;
}

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

// ---------------------------------------------------------------------------------
// Scratch of lz4hip_compress_fast_blocks_noalloc: the Fast kernel's workspace (hash tables
// of the resident waves + the block counter of the dynamic schedule) as a file-scope device
// array instead of a caller's allocation.  Every GPU the library runs on holds its own copy
// (64 MiB + 256 B) at its own address until the process ends.  `scratch` is only a
// host-side stand-in for the symbol: the memory is reached through the per-device address
// that hipGetSymbolAddress returns, never through the name.
// ---------------------------------------------------------------------------------
constexpr size_t kNoAllocSlots = 512;  // two resident waves per CU
constexpr size_t kNoAllocBytes = kNoAllocSlots * kFastSlotBytes + kFastCounterBytes;
constexpr size_t kNoAllocMaxInput = 4ull << 30;  // LZ4HIP_NOALLOC_MAX_BYTES
static_assert(kNoAllocMaxInput == LZ4HIP_NOALLOC_MAX_BYTES, "header and library disagree");
static __device__ __align__(256) uint8_t scratch[kNoAllocBytes];

constexpr int kMaxDevices = 64;
std::mutex g_scratch_mu;
void *g_scratch_ptr[kMaxDevices];  // per device; nullptr = not resolved yet

// Address of the CURRENT device's copy of `scratch` (resolved once per device).
int scratch_address(void **out) {
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) return fail(LZ4HIP_ERR_ARG, "device index out of range");
    std::lock_guard<std::mutex> g(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void *p = nullptr;
        HIPCHK(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return LZ4HIP_OK;
}

}  // namespace

extern "C" {

const char *lz4hip_last_error(void) { return g_err.c_str(); }
const char *lz4hip_version(void) { return "lz4hip 0.1 (gfx950)"; }

int lz4hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// The reference pools its per-call state (sync.Pool: block.go:96,310, blocks.go:42-87).
// Here the per-call state of the host-pointer entry points is device memory taken from the
// device's stream-ordered pool: with the release threshold raised, memory freed by one call
// is kept for the next instead of being returned to the driver at every synchronisation.
int lz4hip_init(int device) {
    int n = 0;
    HIPCHK(hipGetDeviceCount(&n));
    if (device >= n) return fail(LZ4HIP_ERR_ARG, "no such device");
    for (int d = device < 0 ? 0 : device; d < (device < 0 ? n : device + 1); d++) {
        hipMemPool_t pool;
        HIPCHK(hipDeviceGetDefaultMemPool(&pool, d));
        uint64_t keep = UINT64_MAX;
        HIPCHK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep));
    }
    // the scratch of the entry point that allocates nothing, resolved per device with that
    // device current (lz4hip_compress_fast_blocks_noalloc resolves it itself otherwise)
    int prev = 0;
    HIPCHK(hipGetDevice(&prev));
    int rc = LZ4HIP_OK;
    for (int d = device < 0 ? 0 : device; d < (device < 0 ? n : device + 1) && rc == LZ4HIP_OK; d++) {
        void *p = nullptr;
        hipError_t e = hipSetDevice(d);
        rc = e == hipSuccess ? scratch_address(&p) : hip_fail(e, "hipSetDevice");
    }
    (void)hipSetDevice(prev);
    return rc;
}

int lz4hip_shutdown(void) {
    Stager::release_all();  // pinned staging buffers of the host-pointer entry points
    int n = 0;
    HIPCHK(hipGetDeviceCount(&n));
    for (int d = 0; d < n; d++) {
        hipMemPool_t pool;
        HIPCHK(hipDeviceGetDefaultMemPool(&pool, d));
        uint64_t keep = 0;
        HIPCHK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep));
        HIPCHK(hipMemPoolTrimTo(pool, 0));
    }
    return LZ4HIP_OK;
}

size_t lz4hip_compress_bound(size_t n) { return n + n / 255 + 16; }

size_t lz4hip_fast_workspace_bytes(size_t n_blocks) {
    if (n_blocks < kFastGlobalMinBlocks) return 0;  // team kernel (table in LDS), no workspace needed
    // 12 waves per CU on 256 CUs: the measured plateau.  3329..4096 blocks: a table
    // for every block, so that all of them run at once (16 waves per CU, compact LDS layout)
    // instead of a full round and a third of one.
    const size_t want = n_blocks > 3328 && n_blocks <= 4096 ? n_blocks : 3072;
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
    // Few blocks (up to two per CU): a team of waves per block with the table in LDS finishes
    // each block sooner.  Many blocks: global tables and many waves per CU give the higher
    // throughput (crossover measured at 512 blocks: profiles/r3_fast_team.txt).
    if (nb < kFastGlobalMinBlocks) slots = 0;
    HIPCHK(launch_fast((const uint8_t *)d_src, n, block_size, nb, dst_mode, 0, (uint8_t *)d_dst,
                       dst_stride, d_result, (uint16_t *)d_workspace, slots, next_block,
                       (hipStream_t)stream));
    return LZ4HIP_OK;
}

int lz4hip_compress_fast_blocks_noalloc(const void *d_src, size_t n, uint32_t block_size,
                                        int dst_mode, void *d_dst, size_t dst_stride,
                                        int32_t *d_result, void *stream) {
    if (n > kNoAllocMaxInput)
        return fail(LZ4HIP_ERR_ARG, "n exceeds LZ4HIP_NOALLOC_MAX_BYTES (4 GiB): split the call or use "
                                    "lz4hip_compress_fast_blocks with a workspace");
    void *ws = nullptr;
    const int rc = scratch_address(&ws);
    if (rc != LZ4HIP_OK) return rc;
    // Same kernels and launch as the entry point with a caller's workspace; the launch
    // re-arms the block counter in the scratch tail (hipMemsetAsync on `stream`), the tables
    // need no clearing (the kernel's in-use bitmap lives in LDS).
    return lz4hip_compress_fast_blocks(d_src, n, block_size, dst_mode, d_dst, dst_stride, d_result,
                                       ws, kNoAllocBytes, stream);
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

int lz4hip_uncompress_blocks(const void *d_src, size_t src_len, const uint64_t *d_src_off,
                             const uint32_t *d_size_words, size_t n_blocks, void *d_dst,
                             size_t dst_stride, uint32_t dst_len, const void *d_dict,
                             size_t dict_len, int32_t *d_out_len, void *stream) {
    if (n_blocks == 0) return LZ4HIP_OK;
    if (!d_src || !d_src_off || !d_size_words || !d_dst || !d_out_len)
        return fail(LZ4HIP_ERR_ARG, "null pointer");
    if (dict_len && !d_dict) return fail(LZ4HIP_ERR_ARG, "dict_len without d_dict");
    if (dst_stride < dst_len) return fail(LZ4HIP_ERR_ARG, "dst_stride smaller than dst_len");
    if (dst_len > kMaxSingleBlock) return fail(LZ4HIP_ERR_ARG, "dst_len too large");
    if (n_blocks > 0x7fffffffu) return fail(LZ4HIP_ERR_ARG, "too many blocks");
    // only the last 64 KiB of a dictionary are reachable (offsets are 16-bit)
    const uint8_t *dict = static_cast<const uint8_t *>(d_dict);
    if (dict_len > 65536) {
        dict += dict_len - 65536;
        dict_len = 65536;
    }
    HIPCHK(launch_decode((const uint8_t *)d_src, src_len, d_src_off, d_size_words, n_blocks,
                         (uint8_t *)d_dst, dst_stride, dst_len, (uint32_t)dict_len,
                         dict_len ? dict + dict_len : nullptr, d_out_len, (hipStream_t)stream));
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

int lz4hip_frame_index(const void *d_frame, size_t n, uint64_t first_off, uint32_t block_size,
                       int block_checksums, size_t max_blocks, uint64_t *d_src_off,
                       uint32_t *d_size_words, uint32_t *d_want_xxh, int64_t *d_info, void *stream) {
    if (!valid_block_size(block_size)) return fail(LZ4HIP_ERR_ARG, "invalid block size");
    if (!d_frame || !d_src_off || !d_size_words || !d_info || (block_checksums && !d_want_xxh))
        return fail(LZ4HIP_ERR_ARG, "null pointer");
    HIPCHK(launch_frame_index((const uint8_t *)d_frame, n, first_off, block_size, block_checksums,
                              max_blocks, d_src_off, d_size_words, d_want_xxh, d_info,
                              (hipStream_t)stream));
    return LZ4HIP_OK;
}

int lz4hip_pack_blocks(const void *d_slots, size_t stride, const int32_t *d_len,
                       const uint64_t *d_flat_off, size_t n_blocks, void *d_flat, void *stream) {
    if (n_blocks == 0) return LZ4HIP_OK;
    if (!d_slots || !d_len || !d_flat_off || !d_flat) return fail(LZ4HIP_ERR_ARG, "null pointer");
    HIPCHK(launch_pack_blocks((const uint8_t *)d_slots, stride, d_len, d_flat_off, n_blocks,
                              (uint8_t *)d_flat, (hipStream_t)stream));
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

}  // extern "C"
