// host_api.hip -- the extern "C" surface of liblz4hip.so, part 2: the host-pointer entry
// points (what the cgo shim binds) and the lz4hip_stream handle.
//
// Every call stages its blocks through device memory, runs the same kernels as the
// device entry points and synchronises before it returns.  The CONTENT checksum
// (XXHZero over the whole stream, lz4stream/block.go:257-259, frame.go:69-71,117-131) is
// computed on the host, on a helper thread beside the upload / kernels / download: the
// bytes are in host memory on these paths anyway and the hash is one serial chain, which
// a host core runs several times faster than a GPU lane does.  Block checksums
// and everything per block stay on the device.
#include "abi_common.h"

using namespace lz4hip;

namespace {

inline uint32_t rd32(const uint8_t *p) {
    uint32_t v;
    memcpy(&v, p, 4);
    return v;
}

// Device memory of one batch of blocks on the write side.
struct CompressBufs {
    uint8_t *src = nullptr, *comp = nullptr, *frame = nullptr;
    int32_t *res = nullptr;
    uint32_t *xxh = nullptr, *sw = nullptr;
    uint64_t *off = nullptr;
    void *ws = nullptr;  // Fast tables + counter, or HC tables + counter
    size_t ws_bytes = 0;
};

struct BatchShape {
    size_t n, nb_data, nb;  // source bytes, data blocks, blocks incl. ReadFrom's empty one
    uint32_t bs, level;
    bool legacy, bsum;
};

// A large Fast batch is compressed in sub-batches of kPipeBytes while the rest of the source
// is still on its way up: sub-batch j's kernel waits only for its own bytes and runs on one
// of kPipeStreams side streams with a third of the tables, so that up to three of them fill
// the chip together.  0 = this batch is not pipelined.
constexpr size_t kPipeBytes = 256u << 20, kPipeSlots = 1024, kPipePart = kPipeSlots * kFastSlotBytes + kFastCounterBytes;
constexpr int kPipeStreams = 3;

size_t pipe_blocks(uint32_t level, bool legacy, uint32_t bs, size_t nb_data, size_t ws_bytes) {
    const size_t sub = std::max(kPipeSlots, kPipeBytes / bs);
    return level == 0 && !legacy && nb_data >= kPipeStreams * sub && ws_bytes >= kPipeStreams * kPipePart ? sub : 0;
}

struct Event {
    hipEvent_t e = nullptr;
    hipError_t create() { return hipEventCreateWithFlags(&e, hipEventDisableTiming); }
    ~Event() {
        if (e) (void)hipEventDestroy(e);
    }
};

size_t batch_ws_bytes(uint32_t level, size_t nb_data) {
    if (!nb_data) return 0;
    if (level == 0)  // + the block counters of the pipelined sub-batches' own table ranges
        return lz4hip_fast_workspace_bytes(nb_data) + kPipeStreams * kFastCounterBytes;
    return (nb_data < kHcSlots ? nb_data : kHcSlots) * kHcSlotBytes + kFastCounterBytes;
}

// FrameDataBlock.Compress + Write for a batch (lz4stream/block.go:224-276): uploads the
// source, compresses every block, takes the block checksums, scans the entry sizes from
// first_off and writes the entries "u32 size | data | [u32 checksum]" to B.frame (whose
// byte 0 is frame offset frame_base).  On return the stream is idle, *end_off is the
// offset behind the last entry and, if sw_out, the size words are on the host.
int finish_batch(const CompressBufs &B, const BatchShape &S, uint64_t first_off, uint64_t frame_base,
                 hipStream_t st, uint64_t *end_off, uint32_t *sw_out);

int compress_batch(const CompressBufs &B, const uint8_t *src, const BatchShape &S,
                   uint64_t first_off, uint64_t frame_base, hipStream_t st, uint64_t *end_off,
                   uint32_t *sw_out) {
    HIPCHK(hipMemsetAsync(B.res, 0, (S.nb + 1) * sizeof(int32_t), st));
    // dst_len of every block: len(src) in a standard frame, cap(data) = 8 MiB in a legacy one
    const int dmode = S.legacy ? kDstExplicit : LZ4HIP_DST_FRAME;
    const uint32_t dlen_x = S.legacy ? kLegacyBlock : 0;
    Stager::Lease stage;
    const size_t sub = pipe_blocks(S.level, S.legacy, S.bs, S.nb_data, B.ws_bytes);
    if (sub) {
        // declared after the caller's buffers: the side streams are drained before those die
        SideStream side[kPipeStreams];
        Event up[kPipeStreams], done[kPipeStreams];
        for (int p = 0; p < kPipeStreams; p++) {
            HIPCHK(side[p].create());
            HIPCHK(up[p].create());
            HIPCHK(done[p].create());
        }
        size_t j = 0;
        for (size_t b0 = 0; b0 < S.nb_data; b0 += sub, j++) {
            const int p = (int)(j % kPipeStreams);
            const size_t c = std::min(sub, S.nb_data - b0), o = b0 * S.bs, len = std::min(c * S.bs, S.n - o);
            HIPCHK(stage->up(B.src + o, src + o, len, st));
            HIPCHK(hipEventRecord(up[p].e, st));
            HIPCHK(hipStreamWaitEvent(side[p].s, up[p].e, 0));
            uint8_t *w = static_cast<uint8_t *>(B.ws) + (size_t)p * kPipePart;
            size_t slots = 0;
            uint32_t *next_block = fast_ws_split(w, kPipePart, slots);
            HIPCHK(launch_fast(B.src + o, len, S.bs, c, dmode, dlen_x, B.comp + o, S.bs, B.res + b0,
                               (uint16_t *)w, slots, next_block, side[p].s));
        }
        for (int p = 0; p < kPipeStreams; p++) {
            HIPCHK(hipEventRecord(done[p].e, side[p].s));
            HIPCHK(hipStreamWaitEvent(st, done[p].e, 0));
        }
        // finish_batch leaves st idle, and with it the side streams it waits for
        return finish_batch(B, S, first_off, frame_base, st, end_off, sw_out);
    }
    if (S.n) HIPCHK(stage->up(B.src, src, S.n, st));
    if (S.nb_data) {
        if (S.level == 0) {
            size_t slots = 0;
            uint32_t *next_block = fast_ws_split(B.ws, B.ws_bytes, slots);
            if (S.nb_data < kFastGlobalMinBlocks) slots = 0;
            HIPCHK(launch_fast(B.src, S.n, S.bs, S.nb_data, dmode, dlen_x, B.comp, S.bs, B.res,
                               (uint16_t *)B.ws, slots, next_block, st));
        } else {
            size_t slots = B.ws_bytes / kHcSlotBytes;
            if (slots > kHcSlots) slots = kHcSlots;
            uint32_t *next_block =
                B.ws_bytes - slots * kHcSlotBytes >= sizeof(uint32_t)
                    ? reinterpret_cast<uint32_t *>(static_cast<uint8_t *>(B.ws) + slots * kHcSlotBytes)
                    : nullptr;
            HIPCHK(launch_hc(B.src, S.n, S.bs, S.nb_data, S.level, dmode, dlen_x, B.comp, S.bs,
                             B.res, (uint32_t *)B.ws, slots, next_block, st));
        }
    }
    return finish_batch(B, S, first_off, frame_base, st, end_off, sw_out);
}

// The per-block tail of a batch once every block is compressed: block checksums, the scan of
// the entry offsets, the entries written into the frame; leaves st idle.
int finish_batch(const CompressBufs &B, const BatchShape &S, uint64_t first_off, uint64_t frame_base,
                 hipStream_t st, uint64_t *end_off, uint32_t *sw_out) {
    if (S.bsum)  // over the UNCOMPRESSED bytes: lz4stream/block.go:248-250
        HIPCHK(launch_xxh32_blocks(B.src, S.n, S.bs, nullptr, nullptr, S.nb, B.xxh, st));
    HIPCHK(launch_frame_layout(B.res, S.n, S.bs, S.nb, S.bsum ? B.xxh : nullptr, first_off, B.off,
                               B.sw, st));
    HIPCHK(launch_frame_scatter(B.src, S.n, S.bs, B.comp, S.bs, B.res, S.bsum ? B.xxh : nullptr,
                                B.off, S.nb, B.frame, frame_base, st));
    HIPCHK(hipMemcpyAsync(end_off, B.off + S.nb, 8, hipMemcpyDeviceToHost, st));
    if (sw_out && S.nb)
        HIPCHK(hipMemcpyAsync(sw_out, B.sw, S.nb * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return LZ4HIP_OK;
}

}  // namespace

/* ---------------- single blocks ---------------- */

// Return n, 0, -1 (the kernel's verdict), LZ4HIP_ERR_BADARG or LZ4HIP_ERR_HIP: a failed call
// is never folded into the codec's -1 (LZ4HIP_ERR_ARG has that value, so it is not used here).
static int64_t compress_one(const uint8_t *src, size_t n, uint8_t *dst, size_t dst_len,
                            bool hc, uint32_t depth) {
    if (n > kMaxSingleBlock || dst_len > 0x7fffffffu)
        return fail(LZ4HIP_ERR_BADARG, "block larger than 1 GiB");
    if ((n && !src) || (dst_len && !dst)) return fail(LZ4HIP_ERR_BADARG, "null pointer");
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

static int64_t uncompress_one(const uint8_t *src, size_t n, uint8_t *dst, size_t dst_len,
                              const uint8_t *dict, size_t dict_len) {
    if (n == 0) return 0;  // block.go:45-47
    if (n > 0x7fffffffu || dst_len > kMaxSingleBlock)
        return fail(LZ4HIP_ERR_BADARG, "block larger than 1 GiB");
    if (!src || (dst_len && !dst) || (dict_len && !dict))
        return fail(LZ4HIP_ERR_BADARG, "null pointer");
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
    HIPCHK(launch_decode(d_src.as<uint8_t>(), n, (const uint64_t *)m, (const uint32_t *)(m + 8), 1,
                         d_out.as<uint8_t>() + dict_len, 0, (uint32_t)dst_len, (uint32_t)dict_len,
                         nullptr, (int32_t *)(m + 12), kHostStream));
    HIPCHK(hcopy(&meta, d_meta.p, sizeof meta, hipMemcpyDeviceToHost));
    if (meta.out > 0)
        HIPCHK(hcopy(dst, d_out.as<uint8_t>() + dict_len, (size_t)meta.out, hipMemcpyDeviceToHost));
    return meta.out;
}

static int xxh32_one(const uint8_t *p, size_t n, uint32_t *out) {
    if ((n && !p) || !out) return fail(LZ4HIP_ERR_BADARG, "null pointer");
    DevBuf d_src, d_out, d_len;
    HIPCHK(d_src.alloc(n + 16));
    HIPCHK(d_out.alloc(4));
    if (n) HIPCHK(hcopy(d_src.p, p, n, hipMemcpyHostToDevice));
    if (n <= 0x7fffffffu) {
        const int32_t len = (int32_t)n;
        HIPCHK(d_len.alloc(4));
        HIPCHK(hcopy(d_len.p, &len, 4, hipMemcpyHostToDevice));
        HIPCHK(launch_xxh32_blocks(d_src.as<uint8_t>(), n, n ? n : 1, nullptr, d_len.as<int32_t>(), 1,
                                   d_out.as<uint32_t>(), kHostStream));
    } else {
        HIPCHK(launch_xxh32_stream(d_src.as<uint8_t>(), n, d_out.as<uint32_t>(), kHostStream));
    }
    HIPCHK(hcopy(out, d_out.p, 4, hipMemcpyDeviceToHost));
    return LZ4HIP_OK;
}

/* ---------------- whole frames ---------------- */

static int64_t frame_compress(const uint8_t *src, size_t n, uint32_t block_size, uint32_t flags,
                              uint32_t level, uint64_t content_size, uint8_t *dst, size_t dst_cap,
                              std::vector<uint32_t> *lens) {
    const bool legacy = flags & LZ4HIP_F_LEGACY;
    if (legacy) block_size = kLegacyBlock;  // frame.go:47-50
    else if (!valid_block_size(block_size)) return fail(LZ4HIP_ERR_BLOCK_SIZE, "invalid block size");
    if ((n && !src) || !dst) return fail(LZ4HIP_ERR_ARG, "null pointer");
    if (dst_cap < lz4hip_frame_bound(n, block_size, flags))
        return fail(LZ4HIP_ERR_SHORT, "dst_cap < lz4hip_frame_bound()");
    const bool csum = (flags & LZ4HIP_F_CONTENT_CHECKSUM) && !legacy;  // frame.go:63-65
    BatchShape S;
    S.n = n;
    S.bs = block_size;
    S.level = level;
    S.legacy = legacy;
    S.bsum = flags & LZ4HIP_F_BLOCK_CHECKSUM;
    S.nb_data = (n + block_size - 1) / block_size;
    // Close() flushes only pending bytes; ReadFrom writes data[:rn] even when rn == 0
    // (writer.go:161-167 vs :220-233).  A standard frame stores that empty block raw
    // (dst = data[:0]); a legacy frame compresses it into the 8 MiB dst, which yields
    // the one-byte block 00 (lz4stream/block.go:226-227, block.go:258-292).
    const bool empty_tail = (flags & LZ4HIP_F_READFROM) && n % block_size == 0;
    S.nb = S.nb_data + ((empty_tail && !legacy) ? 1 : 0);

    uint8_t hdr[16], trl[12];
    size_t hlen;
    if (legacy) {  // magic only: frame.go:150-165
        const uint32_t m = 0x184C2102u;
        memcpy(hdr, &m, 4);
        hlen = 4;
    } else {
        hlen = lz4hip_frame_header(hdr, block_size, flags, content_size);
    }

    // The content checksum starts at once on its own host thread (XXHZero.Write over the
    // source in stream order) and runs beside the upload, the kernels and the download.
    HostXXH xxh;
    std::unique_ptr<HashFollower> hasher;
    if (csum && n) {
        hasher.reset(new HashFollower(&xxh, src));
        hasher->publish(n);
    }

    DevBuf d_src, d_comp, d_res, d_xxh, d_off, d_sw, d_frame, d_ws;
    CompressBufs B;
    HIPCHK(d_src.alloc(n + 16));
    HIPCHK(d_comp.alloc(S.nb_data * (size_t)block_size));
    HIPCHK(d_res.alloc((S.nb + 1) * sizeof(int32_t)));
    HIPCHK(d_xxh.alloc((S.nb + 1) * sizeof(uint32_t)));
    HIPCHK(d_sw.alloc((S.nb + 1) * sizeof(uint32_t)));
    HIPCHK(d_off.alloc((S.nb + 1) * sizeof(uint64_t)));
    // the frame's own bound, not the caller's capacity (a generous or reused dst costs nothing)
    HIPCHK(d_frame.alloc(lz4hip_frame_bound(n, block_size, flags)));
    B.ws_bytes = batch_ws_bytes(level, S.nb_data);
    if (B.ws_bytes) HIPCHK(d_ws.alloc(B.ws_bytes));
    B.src = d_src.as<uint8_t>(), B.comp = d_comp.as<uint8_t>(), B.frame = d_frame.as<uint8_t>();
    B.res = d_res.as<int32_t>(), B.xxh = d_xxh.as<uint32_t>(), B.sw = d_sw.as<uint32_t>();
    B.off = d_off.as<uint64_t>(), B.ws = d_ws.p;

    // Everything runs on the calling thread's stream, the one the buffers were allocated
    // on and will be freed on: no exit path can free a buffer that a kernel still uses.
    uint64_t end_off = 0;
    std::vector<uint32_t> sw(lens ? S.nb : 0);
    const int rc = compress_batch(B, src, S, hlen, 0, kHostStream, &end_off, lens ? sw.data() : nullptr);
    if (rc != LZ4HIP_OK) return rc;
    if (end_off > dst_cap) return fail(LZ4HIP_ERR_SHORT, "frame larger than dst_cap");
    memcpy(dst, hdr, hlen);
    if (end_off > hlen) {
        Stager::Lease stage;
        HIPCHK(stage->down(dst + hlen, B.frame + hlen, end_off - hlen, kHostStream));
    }
    size_t tlen;
    if (legacy) {  // no end mark (frame.go:63-65); ReadFrom's empty block: size 1, token 00
        tlen = 0;
        if (empty_tail) {
            const uint8_t one[5] = {1, 0, 0, 0, 0};
            memcpy(trl, one, 5);
            tlen = 5;
            if (S.bsum) {  // xxh32("") = 0x02CC5D05 follows even here (lz4stream/block.go:248-250)
                const uint32_t e = 0x02CC5D05u;
                memcpy(trl + 5, &e, 4);
                tlen = 9;
            }
        }
    } else {
        if (hasher) hasher->finish();
        tlen = lz4hip_frame_trailer(trl, flags, xxh.sum32());
    }
    if (end_off + tlen > dst_cap) return fail(LZ4HIP_ERR_SHORT, "frame larger than dst_cap");
    memcpy(dst + end_off, trl, tlen);
    if (lens) {  // len(block.Data) of every block = its size word without the raw bit
        for (size_t b = 0; b < S.nb; b++) lens->push_back(sw[b] & 0x7FFFFFFFu);
        if (legacy && empty_tail) lens->push_back(1);
    }
    return (int64_t)(end_off + tlen);
}

// Legacy frame (magic 0x184C2102, frame.go:20,46-51,169-174): 8 MiB independent
// blocks, no checksums, no end mark.  block.Read's legacy rules
// (lz4stream/block.go:284-294): a size word equal to the magic starts a
// concatenated frame; a size word equal to the bytes decoded so far ends the
// stream (the Linux kernel appends the total size).  That last rule needs the
// decoded sizes, so the stream is taken a chunk of candidate blocks at a time: index up
// to kLegacyChunk of them on the host, decode them in one batch, apply the rule in
// stream order, and index the next chunk only if none of them ended the stream -- what
// follows the end word (padding after a kernel image) is never decoded or given memory.
static int64_t legacy_uncompress(const uint8_t *src, size_t n, size_t si, uint8_t *dst,
                                 size_t dst_cap, std::vector<uint32_t> *lens) {
    const uint32_t cap = kLegacyBlock;
    constexpr size_t kLegacyChunk = 64;  // 512 MiB of slots
    struct Cand {
        uint32_t word;
        int slot;  // index into the device batch; -1: empty block, nothing to decode
    };
    DevBuf d_src, d_off, d_sw, d_out, d_len;
    bool uploaded = false;
    size_t total = 0;
    for (;;) {
        std::vector<Cand> cand;
        std::vector<uint64_t> off;
        std::vector<uint32_t> sw;
        int stop = 0;  // why the index ended: 0 = chunk full, 1 = oversize word, 2 = truncated, 3 = end of input
        while (cand.size() < kLegacyChunk) {
            if (n - si < 4) {
                // io.ReadFull of the next size word: 0 bytes left = io.EOF (clean end),
                // 1..3 = io.ErrUnexpectedEOF (lz4stream/block.go:342-345)
                stop = 3;
                break;
            }
            const uint32_t x = rd32(src + si);
            if (x == 0x184C2102u) {
                si += 4;
                continue;
            }
            const uint32_t size = x & 0x7FFFFFFFu;
            if (size > cap) {
                cand.push_back({x, -1});
                stop = 1;
                break;
            }
            if (n - si - 4 < size) {
                cand.push_back({x, -1});
                stop = 2;
                break;
            }
            if (size == 0) {
                cand.push_back({x, -1});  // decodes to nothing (block.go:45-47)
            } else {
                cand.push_back({x, (int)sw.size()});
                off.push_back(si + 4);
                sw.push_back(x);
            }
            si += 4 + (size_t)size;
        }
        const size_t nb = sw.size();
        std::vector<int32_t> len(nb);
        if (nb) {
            if (!uploaded) {
                HIPCHK(d_src.alloc(n));
                HIPCHK(d_off.alloc(kLegacyChunk * 8));
                HIPCHK(d_sw.alloc(kLegacyChunk * 4));
                HIPCHK(d_len.alloc(kLegacyChunk * 4));
                Stager::Lease stage;  // (the lease's end waits for the last piece to leave its buffers)
                HIPCHK(stage->up(d_src.p, src, n, kHostStream));
                uploaded = true;
            }
            if (!d_out.p) {
                // slots for this chunk and, if the stream goes on, the following ones
                const size_t want = stop == 0 ? kLegacyChunk : nb;
                HIPCHK(d_out.alloc(want * (size_t)cap));
            }
            HIPCHK(hcopy(d_off.p, off.data(), nb * 8, hipMemcpyHostToDevice));
            HIPCHK(hcopy(d_sw.p, sw.data(), nb * 4, hipMemcpyHostToDevice));
            HIPCHK(launch_decode(d_src.as<uint8_t>(), n, d_off.as<uint64_t>(), d_sw.as<uint32_t>(), nb,
                                 d_out.as<uint8_t>(), cap, cap, 0, nullptr, d_len.as<int32_t>(),
                                 kHostStream));
            HIPCHK(hcopy(len.data(), d_len.p, nb * 4, hipMemcpyDeviceToHost));
        }
        for (size_t b = 0; b < cand.size(); b++) {
            if (cand[b].word == (uint32_t)total) return (int64_t)total;  // :289-293
            const bool last = b + 1 == cand.size();
            if (last && stop == 1) return fail(LZ4HIP_ERR_BLOCK_SIZE, "legacy block larger than 8 MiB");
            if (last && stop == 2) return fail(LZ4HIP_ERR_FRAME, "truncated frame");
            const int32_t l = cand[b].slot < 0 ? 0 : len[cand[b].slot];
            if (l < 0) return fail(LZ4HIP_ERR_BLOCK, "invalid source or destination buffer too short");
            if (total + (size_t)l > dst_cap) return fail(LZ4HIP_ERR_SHORT, "dst_cap too small");
            if (l)
                HIPCHK(hcopy(dst + total, d_out.as<uint8_t>() + (size_t)cand[b].slot * cap, (size_t)l,
                             hipMemcpyDeviceToHost));
            total += (size_t)l;
            if (lens) lens->push_back((uint32_t)l);
        }
        if (stop == 3) {
            if (n - si != 0) return fail(LZ4HIP_ERR_FRAME, "truncated frame");
            return (int64_t)total;
        }
    }
}

static int64_t frame_uncompress(const uint8_t *src, size_t n, uint8_t *dst, size_t dst_cap,
                                std::vector<uint32_t> *lens) {
    if (!src || (dst_cap && !dst)) return fail(LZ4HIP_ERR_ARG, "null pointer");
    size_t si = 0;
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
    // A structural error found here (oversize word, truncation) belongs to the position in
    // the stream where the reader meets it: the blocks before it are decoded and verified
    // first, and an error of theirs wins (first error in stream order, :187-194).
    std::vector<uint64_t> off;
    std::vector<uint32_t> sw, want;
    int index_rc = LZ4HIP_OK;
    const char *index_msg = "";
    auto index_fail = [&](int rc, const char *msg) {
        index_rc = rc;
        index_msg = msg;
    };
    for (;;) {
        if (n - si < 4) {
            index_fail(LZ4HIP_ERR_FRAME, "truncated frame");
            break;
        }
        const uint32_t w = rd32(src + si);
        si += 4;
        if (w == 0) break;
        const uint32_t size = w & 0x7FFFFFFFu;
        // :301-304: larger than the block buffer of the frame's block size
        if (size > bs) {
            index_fail(LZ4HIP_ERR_BLOCK_SIZE, "block larger than the frame's block size");
            break;
        }
        if (n - si < size || (bsum && n - si - size < 4)) {
            index_fail(LZ4HIP_ERR_FRAME, "truncated frame");
            break;
        }
        off.push_back(si);
        sw.push_back(w);
        si += size;
        if (bsum) {
            want.push_back(rd32(src + si));
            si += 4;
        }
    }
    uint32_t want_cs = 0;
    if (csum && index_rc == LZ4HIP_OK) {
        if (n - si < 4)
            index_fail(LZ4HIP_ERR_FRAME, "truncated frame");
        else
            want_cs = rd32(src + si);
    }
    const size_t nb = sw.size();
    if (nb == 0) {
        if (index_rc != LZ4HIP_OK) return fail(index_rc, index_msg);
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
    DevBuf d_src, d_off, d_sw, d_slots, d_flat, d_len, d_xxh, d_foff;
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
    if (independent) HIPCHK(d_slots.alloc(chunk * (size_t)bs));
    Stager::Lease stage;
    HIPCHK(stage->up(d_src.p, src, n, kHostStream));
    HIPCHK(hcopy(d_off.p, off.data(), nb * 8, hipMemcpyHostToDevice));
    HIPCHK(hcopy(d_sw.p, sw.data(), nb * 4, hipMemcpyHostToDevice));
    std::vector<int32_t> len(chunk);
    std::vector<uint32_t> got(chunk);
    std::vector<uint64_t> foff(chunk);
    size_t total = 0, down = 0;
    // frame.go:117-131: the content checksum is one serial walk over the decoded stream; it
    // follows the pieces as they arrive in the caller's buffer, on its own host thread,
    // while the next chunk decodes and the next piece comes down.
    HostXXH xxh;
    std::unique_ptr<HashFollower> hasher;
    if (csum) hasher.reset(new HashFollower(&xxh, dst));
    for (size_t b0 = 0; b0 < nb; b0 += chunk) {
        const size_t c = std::min(chunk, nb - b0);
        if (independent) {
            HIPCHK(launch_decode(d_src.as<uint8_t>(), n, d_off.as<uint64_t>() + b0,
                                 d_sw.as<uint32_t>() + b0, c, d_slots.as<uint8_t>(), bs, bs, 0, nullptr,
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
        // this chunk's bytes go down in pieces; the hash follows each piece
        if (down < total) {
            const size_t base = down;
            HIPCHK(stage->down(dst + base, d_flat.as<uint8_t>() + base, total - base, kHostStream,
                               [&](size_t upto) {
                                   if (hasher) hasher->publish(base + upto);
                               }));
            down = total;
        }
    }
    if (index_rc != LZ4HIP_OK) return fail(index_rc, index_msg);
    if (csum) {
        hasher->finish();
        if (xxh.sum32() != want_cs) return fail(LZ4HIP_ERR_FRAME_CHECKSUM, "invalid frame checksum");
    }
    return (int64_t)total;
}

/* ---------------- the stream handle ---------------- */

// One per Writer or Reader.  What the reference keeps per frame -- the content checksum
// state (Frame.checksum, frame.go:27) and pooled block buffers (blocks.go:42-87) -- plus
// what only a GPU path needs: device buffers for one batch of ConcurrencyOption(n) blocks
// and a HIP stream, allocated once and reused by every call.
struct lz4hip_stream {
    uint32_t bs = 0, flags = 0, level = 0;
    size_t max_blocks = 0;
    int device = 0;
    HostXXH xxh;
    hipStream_t st = nullptr;
    // write side
    CompressBufs W;
    size_t frame_cap = 0;
    // read side
    uint8_t *r_src = nullptr, *r_slots = nullptr, *r_flat = nullptr;
    uint64_t *r_off = nullptr, *r_foff = nullptr;
    uint32_t *r_sw = nullptr, *r_xxh = nullptr;
    int32_t *r_len = nullptr;
    size_t r_src_cap = 0;
    std::vector<void *> owned;

    hipError_t take(void **p, size_t n) {
        hipError_t e = hipMalloc(p, n ? n : 16);
        if (e == hipSuccess) owned.push_back(*p);
        return e;
    }
    ~lz4hip_stream() {
        if (st) {
            (void)hipStreamSynchronize(st);
            (void)hipStreamDestroy(st);
        }
        for (void *p : owned) (void)hipFree(p);
    }
};

namespace {

struct DeviceGuard {  // makes the handle's device current for the duration of a call
    int prev = -1;
    hipError_t enter(int dev) {
        hipError_t e = hipGetDevice(&prev);
        if (e != hipSuccess) return e;
        return prev == dev ? hipSuccess : hipSetDevice(dev);
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

size_t entry_bound(size_t n, size_t nb) { return n + nb * 8; }  // size word + data + checksum each

int stream_alloc(lz4hip_stream *s) {
    const size_t mb = s->max_blocks, bs = s->bs;
    HIPCHK(hipStreamCreateWithFlags(&s->st, hipStreamNonBlocking));
    // write side
    s->frame_cap = entry_bound(mb * bs, mb + 1) + 16;
    HIPCHK(s->take((void **)&s->W.src, mb * bs + 16));
    HIPCHK(s->take((void **)&s->W.comp, mb * bs));
    HIPCHK(s->take((void **)&s->W.frame, s->frame_cap));
    HIPCHK(s->take((void **)&s->W.res, (mb + 2) * 4));
    HIPCHK(s->take((void **)&s->W.xxh, (mb + 2) * 4));
    HIPCHK(s->take((void **)&s->W.sw, (mb + 2) * 4));
    HIPCHK(s->take((void **)&s->W.off, (mb + 2) * 8));
    s->W.ws_bytes = batch_ws_bytes(s->level, mb);
    if (s->W.ws_bytes) HIPCHK(s->take(&s->W.ws, s->W.ws_bytes));
    // read side: the same memory serves it (a handle is a Writer's or a Reader's)
    s->r_src = s->W.frame;
    s->r_src_cap = s->frame_cap;
    s->r_slots = s->W.comp;
    s->r_flat = s->W.src;
    s->r_off = s->W.off;
    s->r_sw = s->W.sw;
    s->r_xxh = s->W.xxh;
    s->r_len = s->W.res;
    HIPCHK(s->take((void **)&s->r_foff, (mb + 2) * 8));
    return LZ4HIP_OK;
}

}  // namespace

extern "C" {

int64_t lz4hip_compress_block(const uint8_t *src, size_t n, uint8_t *dst, size_t dst_len) {
    return compress_one(src, n, dst, dst_len, false, 0);
}

int64_t lz4hip_compress_block_hc(const uint8_t *src, size_t n, uint8_t *dst, size_t dst_len,
                                 uint32_t depth) {
    return compress_one(src, n, dst, dst_len, true, depth);
}

int64_t lz4hip_uncompress_block(const uint8_t *src, size_t n, uint8_t *dst, size_t dst_len,
                                const uint8_t *dict, size_t dict_len) {
    return uncompress_one(src, n, dst, dst_len, dict, dict_len);
}

uint32_t lz4hip_xxh32(const uint8_t *p, size_t n) {
    uint32_t out = 0;
    return xxh32_one(p, n, &out) == LZ4HIP_OK ? out : 0;
}

int lz4hip_xxh32_checked(const uint8_t *p, size_t n, uint32_t *out) {
    return xxh32_one(p, n, out);
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

// The content checksum of device-resident bytes, computed on the host.  The chain
// v <- rol13(v + x*P2)*P1 is serial; one GPU lane steps it every ~15 cycles (2.4 GB/s), a host
// core every ~5.  The bytes come down through two pinned buffers, one on the bus
// while the other is hashed on the calling thread, so the call runs at the host's hash rate.
int lz4hip_xxh32_stream_update_host(const void *d_src, size_t n, uint32_t *h_state, int first,
                                    int final_piece, uint32_t *h_digest, void *stream) {
    if (!h_state || (!d_src && n) || (final_piece && !h_digest))
        return fail(LZ4HIP_ERR_ARG, "null pointer");
    if (!final_piece && (n & 15))
        return fail(LZ4HIP_ERR_ARG, "a non-final piece must be a multiple of 16 bytes");
    HostXXH x;
    if (!first) {
        memcpy(x.v, h_state, 16);
        x.total = (uint64_t)h_state[4] | ((uint64_t)h_state[5] << 32);
        // a state handed over with totalLen 0 would be reset by the next Write
        // (xxh32zero.go:59-61), exactly as the reference's would
    }
    // The pinned buffers and events are a leased Stager's (per device, kept on the free list
    // between calls and released by lz4hip_shutdown): nothing is tied to the calling thread.
    {
        Stager::Lease stage;
        HIPCHK(stage->drain(d_src, n, (hipStream_t)stream,
                            [&](const uint8_t *p, size_t len) { x.write(p, len); }));
    }
    memcpy(h_state, x.v, 16);
    h_state[4] = (uint32_t)x.total;
    h_state[5] = (uint32_t)(x.total >> 32);
    if (final_piece) *h_digest = x.sum32();
    return LZ4HIP_OK;
}

// The same chain step over bytes that already are in host memory (a shard staged into pinned
// memory while its rank waited for the state): XXHZero.Write / Sum32 and nothing else.
int lz4hip_xxh32_host_update(const void *h_src, size_t n, uint32_t *h_state, int first,
                             int final_piece, uint32_t *h_digest) {
    if (!h_state || (!h_src && n) || (final_piece && !h_digest))
        return fail(LZ4HIP_ERR_ARG, "null pointer");
    if (!final_piece && (n & 15))
        return fail(LZ4HIP_ERR_ARG, "a non-final piece must be a multiple of 16 bytes");
    HostXXH x;
    if (!first) {
        memcpy(x.v, h_state, 16);
        x.total = (uint64_t)h_state[4] | ((uint64_t)h_state[5] << 32);
    }
    x.write(static_cast<const uint8_t *>(h_src), n);
    memcpy(h_state, x.v, 16);
    h_state[4] = (uint32_t)x.total;
    h_state[5] = (uint32_t)(x.total >> 32);
    if (final_piece) *h_digest = x.sum32();
    return LZ4HIP_OK;
}

lz4hip_stream *lz4hip_stream_create(uint32_t block_size, uint32_t flags, uint32_t level,
                                    size_t max_blocks) {
    if (flags & LZ4HIP_F_LEGACY) block_size = kLegacyBlock;
    else if (!valid_block_size(block_size)) {
        fail(LZ4HIP_ERR_BLOCK_SIZE, "invalid block size");
        return nullptr;
    }
    if (max_blocks == 0 || max_blocks > (1u << 20) || max_blocks * (size_t)block_size > (64ull << 30)) {
        fail(LZ4HIP_ERR_ARG, "max_blocks out of range");
        return nullptr;
    }
    lz4hip_stream *s = new lz4hip_stream;
    s->bs = block_size;
    s->flags = flags;
    s->level = level;
    s->max_blocks = max_blocks;
    if (hipGetDevice(&s->device) != hipSuccess || stream_alloc(s) != LZ4HIP_OK) {
        if (g_err.empty()) g_err = "no HIP device";
        delete s;
        return nullptr;
    }
    return s;
}

void lz4hip_stream_destroy(lz4hip_stream *s) {
    if (!s) return;
    DeviceGuard g;
    (void)g.enter(s->device);
    delete s;
}

void lz4hip_stream_reset(lz4hip_stream *s) {
    if (s) s->xxh.reset();
}

uint32_t lz4hip_stream_checksum(const lz4hip_stream *s) { return s ? s->xxh.sum32() : 0; }

int64_t lz4hip_stream_compress_blocks(lz4hip_stream *s, const uint8_t *src, size_t n, uint8_t *dst,
                                      size_t dst_cap, uint32_t *block_lens) {
    if (!s || (n && !src) || !dst) return fail(LZ4HIP_ERR_ARG, "null pointer");
    if (n > s->max_blocks * (size_t)s->bs)
        return fail(LZ4HIP_ERR_ARG, "more than max_blocks blocks in one call");
    BatchShape S;
    S.n = n;
    S.bs = s->bs;
    S.level = s->level;
    S.legacy = s->flags & LZ4HIP_F_LEGACY;
    S.bsum = s->flags & LZ4HIP_F_BLOCK_CHECKSUM;
    S.nb_data = (n + s->bs - 1) / s->bs;
    // Writer.write(data[:0]) -- the block io.Copy leaves behind a block multiple
    // (writer.go:220-233) -- is one empty block, stored raw in a standard frame
    S.nb = S.nb_data ? S.nb_data : 1;
    if (S.legacy && n == 0) {  // compressed into the 8 MiB dst: the one-byte block 00
        const size_t k = S.bsum ? 9 : 5;
        if (dst_cap < k) return fail(LZ4HIP_ERR_SHORT, "dst_cap too small");
        const uint8_t one[9] = {1, 0, 0, 0, 0, 0x05, 0x5D, 0xCC, 0x02};
        memcpy(dst, one, k);
        if (block_lens) block_lens[0] = 1;
        return (int64_t)k;
    }
    const size_t bound = entry_bound(n, S.nb) + (S.legacy ? n % s->bs / 255 + 16 : 0);
    if (dst_cap < bound) return fail(LZ4HIP_ERR_SHORT, "dst_cap < n + 8 * n_blocks (+ n/255 + 16 for a legacy frame)");
    DeviceGuard g;
    HIPCHK(g.enter(s->device));
    const bool csum = (s->flags & LZ4HIP_F_CONTENT_CHECKSUM) && !S.legacy;
    std::unique_ptr<HashFollower> hasher;
    if (csum && n) {  // Write folds src into the frame's checksum in block order: :257-259
        hasher.reset(new HashFollower(&s->xxh, src));
        hasher->publish(n);
    }
    uint64_t end_off = 0;
    const int rc = compress_batch(s->W, src, S, 0, 0, s->st, &end_off, block_lens);
    if (rc != LZ4HIP_OK) return rc;
    if (end_off > dst_cap) return fail(LZ4HIP_ERR_SHORT, "dst_cap too small");
    if (end_off) {
        Stager::Lease stage;
        HIPCHK(stage->down(dst, s->W.frame, end_off, s->st));
    }
    if (block_lens)
        for (size_t b = 0; b < S.nb; b++) block_lens[b] &= 0x7FFFFFFFu;
    return (int64_t)end_off;
}

int64_t lz4hip_stream_uncompress_blocks(lz4hip_stream *s, const uint8_t *src, size_t n, uint8_t *dst,
                                        size_t dst_cap, size_t *consumed, int *end_mark,
                                        uint32_t *block_lens, size_t *n_blocks) {
    if (!s || (n && !src) || (dst_cap && !dst) || !consumed || !end_mark)
        return fail(LZ4HIP_ERR_ARG, "null pointer");
    if (s->flags & (LZ4HIP_F_LEGACY | LZ4HIP_F_LINKED))
        return fail(LZ4HIP_ERR_ARG, "legacy and linked frames do not decode in batches (reader.go:86-90)");
    const bool bsum = s->flags & LZ4HIP_F_BLOCK_CHECKSUM, csum = s->flags & LZ4HIP_F_CONTENT_CHECKSUM;
    const uint32_t bs = s->bs;
    *consumed = 0;
    *end_mark = 0;
    if (n_blocks) *n_blocks = 0;
    // ---- FrameDataBlock.Read per entry (lz4stream/block.go:279-317), whole entries only ----
    std::vector<uint64_t> off;
    std::vector<uint32_t> sw, want;
    size_t si = 0;
    while (off.size() < s->max_blocks && n - si >= 4) {
        const uint32_t w = rd32(src + si);
        if (w == 0) {  // :295-298
            si += 4;
            *end_mark = 1;
            break;
        }
        const uint32_t size = w & 0x7FFFFFFFu;
        if (size > bs) {  // :301-304 -- once the entries before it have been delivered
            if (off.empty()) return fail(LZ4HIP_ERR_BLOCK_SIZE, "block larger than the frame's block size");
            break;
        }
        const size_t entry = 4 + (size_t)size + (bsum ? 4 : 0);
        if (n - si < entry) break;  // the caller has not read this entry completely yet
        off.push_back(si + 4);
        sw.push_back(w);
        if (bsum) want.push_back(rd32(src + si + 4 + size));
        si += entry;
    }
    const size_t nb = sw.size();
    if (si > s->r_src_cap) return fail(LZ4HIP_ERR_ARG, "entries larger than the handle's buffers");
    int64_t total = 0;
    if (nb) {
        DeviceGuard g;
        HIPCHK(g.enter(s->device));
        hipStream_t st = s->st;
        Stager::Lease stage;
        HIPCHK(stage->up(s->r_src, src, si, st));
        HIPCHK(hipMemcpyAsync(s->r_off, off.data(), nb * 8, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(s->r_sw, sw.data(), nb * 4, hipMemcpyHostToDevice, st));
        HIPCHK(launch_decode(s->r_src, si, s->r_off, s->r_sw, nb, s->r_slots, bs, bs, 0, nullptr,
                             s->r_len, st));
        if (bsum)  // on the DECODED bytes: lz4stream/block.go:330-335
            HIPCHK(launch_xxh32_blocks(s->r_slots, nb * (size_t)bs, bs, nullptr, s->r_len, nb, s->r_xxh, st));
        std::vector<int32_t> len(nb);
        std::vector<uint32_t> got(nb);
        std::vector<uint64_t> foff(nb);
        HIPCHK(hipMemcpyAsync(len.data(), s->r_len, nb * 4, hipMemcpyDeviceToHost, st));
        if (bsum) HIPCHK(hipMemcpyAsync(got.data(), s->r_xxh, nb * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        // the first failing block in stream order decides the error: lz4stream/block.go:187-194
        for (size_t b = 0; b < nb; b++) {
            if (len[b] < 0) return fail(LZ4HIP_ERR_BLOCK, "invalid source or destination buffer too short");
            if (bsum && got[b] != want[b]) return fail(LZ4HIP_ERR_BLOCK_CHECKSUM, "invalid block checksum");
            foff[b] = (uint64_t)total;
            total += len[b];
            if (block_lens) block_lens[b] = (uint32_t)len[b];
        }
        if ((size_t)total > dst_cap) return fail(LZ4HIP_ERR_SHORT, "dst_cap too small");
        HIPCHK(hipMemcpyAsync(s->r_foff, foff.data(), nb * 8, hipMemcpyHostToDevice, st));
        HIPCHK(launch_pack_blocks(s->r_slots, bs, s->r_len, s->r_foff, nb, s->r_flat, st));
        std::unique_ptr<HashFollower> hasher;
        if (csum) hasher.reset(new HashFollower(&s->xxh, dst));  // the collector's Write: :173-176
        HIPCHK(stage->down(dst, s->r_flat, (size_t)total, st, [&](size_t upto) {
            if (hasher) hasher->publish(upto);
        }));
    }
    *consumed = si;
    if (n_blocks) *n_blocks = nb;
    return total;
}

}  // extern "C"
