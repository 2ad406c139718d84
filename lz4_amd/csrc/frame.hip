// frame.hip -- frame assembly on device.
//
// Restates FrameDataBlock.Compress's raw fallback and size word
// (internal/lz4stream/block.go:238-245), FrameDataBlock.Write (:254-276: u32le
// size | payload | [u32le checksum unless 0]) for a batch of blocks: an
// exclusive scan of the entry sizes gives every block its offset in the frame,
// then one workgroup per block copies its entry there.
#include "common.h"

namespace lz4hip {

__device__ __forceinline__ uint32_t block_len(size_t n, uint32_t bs, size_t b) {
    const size_t o = b * (size_t)bs;
    if (o >= n) return 0;  // trailing empty block of Writer.ReadFrom (writer.go:220-233)
    const size_t left = n - o;
    return (uint32_t)(left < bs ? left : bs);
}

// Single workgroup, 1024 threads: chunked inclusive scan with a running carry.
__global__ __launch_bounds__(1024) void k_frame_layout(const int32_t *__restrict__ result, size_t n,
                                                       uint32_t bs, size_t nb,
                                                       const uint32_t *__restrict__ xxh,
                                                       uint64_t first_off,
                                                       uint64_t *__restrict__ entry_off,
                                                       uint32_t *__restrict__ size_words) {
    __shared__ uint64_t wave_sum[16];
    __shared__ uint64_t carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) carry_s = first_off;
    __syncthreads();
    for (size_t base = 0; base < nb; base += 1024) {
        const size_t b = base + tid;
        uint64_t e = 0;
        if (b < nb) {
            const int32_t r = result[b];
            const uint32_t len = block_len(n, bs, b);
            const uint32_t payload = r > 0 ? (uint32_t)r : len;
            e = 4 + (uint64_t)payload + ((xxh && xxh[b] != 0) ? 4 : 0);  // :270-272
            if (size_words) size_words[b] = r > 0 ? payload : (payload | 0x80000000u);
        }
        uint64_t v = e;  // inclusive scan inside the wave
        for (int o = 1; o < 64; o <<= 1) {
            const uint64_t t = __shfl_up(v, o);
            if (lane >= o) v += t;
        }
        if (lane == 63) wave_sum[wid] = v;
        __syncthreads();
        uint64_t pre = carry_s;
        for (int w = 0; w < wid; w++) pre += wave_sum[w];
        if (b < nb) entry_off[b] = pre + v - e;
        __syncthreads();
        if (tid == 1023) carry_s = pre + v;
        __syncthreads();
    }
    if (tid == 0) entry_off[nb] = carry_s;
}

__device__ __forceinline__ void put32(uint8_t *p, uint32_t v) {
    p[0] = (uint8_t)v;
    p[1] = (uint8_t)(v >> 8);
    p[2] = (uint8_t)(v >> 16);
    p[3] = (uint8_t)(v >> 24);
}

// Workgroup-wide copy with 16-byte stores on the (arbitrarily aligned) destination.
__device__ __forceinline__ void wg_copy(uint8_t *d, const uint8_t *s, size_t n, int tid, int nt) {
    size_t head = (16 - (reinterpret_cast<uintptr_t>(d) & 15)) & 15;
    if (head > n) head = n;
    for (size_t k = tid; k < head; k += nt) d[k] = s[k];
    const size_t body = (n - head) / 16;
    uint4 *d4 = reinterpret_cast<uint4 *>(d + head);
    const uint8_t *sb = s + head;
    for (size_t k = tid; k < body; k += nt) {
        uint4 v;
        __builtin_memcpy(&v, sb + k * 16, 16);  // unaligned global load
        d4[k] = v;
    }
    const size_t done = head + body * 16;
    for (size_t k = done + tid; k < n; k += nt) d[k] = s[k];
}

__global__ __launch_bounds__(256) void k_frame_scatter(
    const uint8_t *__restrict__ src, size_t n, uint32_t bs, const uint8_t *__restrict__ comp,
    size_t comp_stride, const int32_t *__restrict__ result, const uint32_t *__restrict__ xxh,
    const uint64_t *__restrict__ entry_off, size_t nb, uint8_t *__restrict__ frame,
    uint64_t frame_base) {
    const size_t b = blockIdx.x;
    if (b >= nb) return;
    const int tid = threadIdx.x;
    const int32_t r = result[b];
    const uint32_t len = block_len(n, bs, b);
    const uint32_t payload = r > 0 ? (uint32_t)r : len;
    uint8_t *e = frame + (entry_off[b] - frame_base);
    if (tid == 0) put32(e, r > 0 ? payload : (payload | 0x80000000u));
    const uint8_t *from = r > 0 ? comp + b * comp_stride : src + b * (size_t)bs;
    wg_copy(e + 4, from, payload, tid, 256);
    if (xxh && tid == 0) {
        const uint32_t c = xxh[b];
        if (c != 0) put32(e + 4 + payload, c);
    }
}

// Reader side: decoded blocks sit in fixed-stride slots; pack the ones that are not full
// into the contiguous stream (block b's out_len bytes go to flat + flat_off[b]).
__global__ __launch_bounds__(256) void k_pack_blocks(const uint8_t *__restrict__ slots,
                                                     size_t stride,
                                                     const int32_t *__restrict__ out_len,
                                                     const uint64_t *__restrict__ flat_off,
                                                     size_t nb, uint8_t *__restrict__ flat) {
    const size_t b = blockIdx.x;
    if (b >= nb) return;
    const int32_t l = out_len[b];
    if (l > 0) wg_copy(flat + flat_off[b], slots + b * stride, (size_t)l, threadIdx.x, 256);
}

// Reader side, block index of a standard frame that sits in device memory
// (FrameDataBlock.Read per block, lz4stream/block.go:279-317): every size word's place
// depends on the sizes before it, so one lane walks them -- a dependent 4-byte load per
// block, ~50 ms for the 32,768 blocks of a 128 GiB frame -- and records payload offset,
// size word and expected checksum per block.  info[0] = blocks found, info[1] = offset
// behind the end mark (or where the walk stopped), info[2] = 0 or the LZ4HIP_ERR_* the
// reader meets at that point (as int64), info[3] = 1 when max_blocks was too small.
__global__ __launch_bounds__(64) void k_frame_index(const uint8_t *__restrict__ frame, size_t n,
                                                    uint64_t first_off, uint32_t bs, int bsum,
                                                    size_t max_blocks, uint64_t *__restrict__ off,
                                                    uint32_t *__restrict__ sw,
                                                    uint32_t *__restrict__ want,
                                                    int64_t *__restrict__ info) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    size_t si = first_off, b = 0;
    int64_t status = 0, full = 0;
    for (;;) {
        if (si > n || n - si < 4) {
            status = LZ4HIP_ERR_FRAME;  // io.ErrUnexpectedEOF
            break;
        }
        const uint32_t w = ld32(frame + si);
        if (w == 0) {  // end mark: :295-298
            si += 4;
            break;
        }
        const uint32_t size = w & 0x7FFFFFFFu;
        if (size > bs) {  // :301-304
            status = LZ4HIP_ERR_BLOCK_SIZE;
            break;
        }
        if (n - si - 4 < size || (bsum && n - si - 4 - size < 4)) {
            status = LZ4HIP_ERR_FRAME;
            break;
        }
        if (b == max_blocks) {
            full = 1;
            break;
        }
        off[b] = si + 4;
        sw[b] = w;
        si += 4 + (size_t)size;
        if (bsum) {
            want[b] = ld32(frame + si);
            si += 4;
        }
        b++;
    }
    info[0] = (int64_t)b;
    info[1] = (int64_t)si;
    info[2] = status;
    info[3] = full;
}

hipError_t launch_frame_index(const uint8_t *frame, size_t n, uint64_t first_off, uint32_t bs,
                              int bsum, size_t max_blocks, uint64_t *off, uint32_t *sw,
                              uint32_t *want, int64_t *info, hipStream_t st) {
    hipLaunchKernelGGL(k_frame_index, dim3(1), dim3(64), 0, st, frame, n, first_off, bs, bsum,
                       max_blocks, off, sw, want, info);
    return hipGetLastError();
}

hipError_t launch_pack_blocks(const uint8_t *slots, size_t stride, const int32_t *out_len,
                              const uint64_t *flat_off, size_t nb, uint8_t *flat, hipStream_t st) {
    if (nb == 0) return hipSuccess;
    hipLaunchKernelGGL(k_pack_blocks, dim3((unsigned)nb), dim3(256), 0, st, slots, stride, out_len,
                       flat_off, nb, flat);
    return hipGetLastError();
}

hipError_t launch_frame_layout(const int32_t *result, size_t n, uint32_t bs, size_t nb,
                               const uint32_t *xxh, uint64_t first_off, uint64_t *entry_off,
                               uint32_t *size_words, hipStream_t st) {
    hipLaunchKernelGGL(k_frame_layout, dim3(1), dim3(1024), 0, st, result, n, bs, nb, xxh,
                       first_off, entry_off, size_words);
    return hipGetLastError();
}

hipError_t launch_frame_scatter(const uint8_t *src, size_t n, uint32_t bs, const uint8_t *comp,
                                size_t comp_stride, const int32_t *result, const uint32_t *xxh,
                                const uint64_t *entry_off, size_t nb, uint8_t *frame,
                                uint64_t frame_base, hipStream_t st) {
    if (nb == 0) return hipSuccess;
    hipLaunchKernelGGL(k_frame_scatter, dim3((unsigned)nb), dim3(256), 0, st, src, n, bs, comp,
                       comp_stride, result, xxh, entry_off, nb, frame, frame_base);
    return hipGetLastError();
}

}  // namespace lz4hip
