// common.h -- shared device helpers and launch declarations of liblz4hip (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/lz4hip.h"

namespace lz4hip {

constexpr int kWave = 64;
constexpr int kMinMatch = 4;
constexpr int kWinSize = 1 << 16;
constexpr int kWinMask = kWinSize - 1;
constexpr int kMfLimit = 14;  // block.go:25
constexpr int kAdaptSkipLog = 7;
// HC chain table: links per entry (hc.hip) and the bytes of one workspace slot
// (hashTable u32[65536] + chainTable kHcLinks x u16 [65536] + one byte per window position that
// says whether the position is in the tables, hc.hip).
#ifndef LZ4HIP_HC_LINKS
#define LZ4HIP_HC_LINKS 32
#endif
constexpr int kHcLinks = LZ4HIP_HC_LINKS;
constexpr size_t kHcSlotBytes = 65536 * 4 + (size_t)65536 * kHcLinks * 2 + 65536;
constexpr int kDstExplicit = 2;  // internal dst_mode: dst_len given explicitly (single-block API)

__host__ __device__ inline uint32_t compress_bound32(uint32_t n) { return n + n / 255 + 16; }

// Unaligned little-endian loads.  gfx950 global memory runs in unaligned access
// mode, so these lower to single global_load_dword/dwordx2.
__device__ __forceinline__ uint32_t ld32(const uint8_t *p) {
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
__device__ __forceinline__ uint64_t ld64(const uint8_t *p) {
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}

__device__ __forceinline__ uint32_t rfl(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ uint64_t ballot(bool p) { return __ballot(p); }

// Inclusive prefix sum over the 64 lanes with DPP row shifts and row broadcasts (six
// v_add_u32 with a DPP operand; the shuffle form costs six LDS permutes plus selects).
__device__ __forceinline__ int wave_incl_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);  // row_bcast:31 -> rows 2, 3
    return v;
}

// block.go:35-38
__device__ __forceinline__ uint32_t hash6(uint64_t x) {
    return (uint32_t)(((x << 16) * 227718039650203ULL) >> 48);
}
// block.go:298-301
__device__ __forceinline__ uint32_t hash4(uint32_t x) { return (x * 2654435761u) >> 16; }

// Emits "length - 15" as 255-run + remainder starting at d[di]; returns new di, or
// -1 when a byte would land at or beyond dlen (block.go:211-219,239-246 and the
// index panics of :424-428,443-447).  Lane 0 stores; every lane tracks di.
__device__ __forceinline__ int emit_len_tail(uint8_t *d, int di, int dlen, int v, int lane) {
    int nff = v / 255;
    // bytes di .. di+nff-1 are 0xFF, byte di+nff is v % 255
    if (di + nff >= dlen) return -1;
    for (int k = lane; k < nff; k += kWave) d[di + k] = 0xFF;
    if (lane == 0) d[di + nff] = (uint8_t)(v - nff * 255);
    return di + nff + 1;
}

// Wave-wide byte copy d[0..n) = s[0..n) (no overlap).
__device__ __forceinline__ void wave_copy(uint8_t *d, const uint8_t *s, int n, int lane) {
    for (int k = lane; k < n; k += kWave) d[k] = s[k];
}

// Wave-wide copy of a long span, 16 bytes per lane (either pointer may be unaligned);
// the last n % 16 bytes go byte-wise.
__device__ __forceinline__ void wave_copy16(uint8_t *d, const uint8_t *s, int n, int lane) {
    const int body = n & ~15;
    for (int k = lane * 16; k < body; k += kWave * 16) {
        const uint64_t a = ld64(s + k), b = ld64(s + k + 8);
        __builtin_memcpy(d + k, &a, 8);
        __builtin_memcpy(d + k + 8, &b, 8);
    }
    wave_copy(d + body, s + body, n - body, lane);
}

// kernel launchers (defined in the .hip files)
// next_block: one zero-initialisable uint32 in device memory for the dynamic block
// schedule of the global-table kernel (nullptr: blocks are dealt round-robin).
hipError_t launch_fast(const uint8_t *src, size_t n, uint32_t bs, size_t nb, int dst_mode,
                       uint32_t dst_len_x, uint8_t *dst, size_t dst_stride, int32_t *result,
                       uint16_t *ws, size_t slots, uint32_t *next_block, hipStream_t st);
// Few blocks: a team of four waves per block, table in LDS (fast_team.hip).
hipError_t launch_fast_team(const uint8_t *src, size_t n, uint32_t bs, size_t nb, int dst_mode,
                            uint32_t dst_len_x, uint8_t *dst, size_t dst_stride, int32_t *result,
                            hipStream_t st);
hipError_t launch_hc(const uint8_t *src, size_t n, uint32_t bs, size_t nb, uint32_t depth,
                     int dst_mode,
                     uint32_t dst_len_x, uint8_t *dst, size_t dst_stride, int32_t *result, uint32_t *ws,
                     size_t slots, uint32_t *next_block, hipStream_t st);
// src_len: bytes readable at src (a block whose span leaves them yields -1 unread).
// History of every block: `prefix` bytes ending at dict_end, or, with dict_end == nullptr,
// the `prefix` bytes in memory right in front of the block's dst.
hipError_t launch_decode(const uint8_t *src, size_t src_len, const uint64_t *off, const uint32_t *sw,
                         size_t nb, uint8_t *dst, size_t dst_stride, uint32_t dst_len,
                         uint32_t prefix, const uint8_t *dict_end, int32_t *out_len, hipStream_t st);
hipError_t launch_decode_linked(const uint8_t *src, const uint64_t *off, const uint32_t *sw,
                                size_t nb, uint8_t *dst, uint32_t block_cap, size_t dst_cap,
                                int32_t *out_len, uint64_t *out_off, hipStream_t st);
hipError_t launch_xxh32_blocks(const uint8_t *base, size_t n, size_t stride, const uint64_t *off,
                               const int32_t *len, size_t n_spans, uint32_t *out, hipStream_t st);
hipError_t launch_xxh32_stream(const uint8_t *src, size_t n, uint32_t *out, hipStream_t st);
hipError_t launch_xxh32_update(const uint8_t *src, size_t n, uint32_t *state, int first,
                               int final_piece, uint32_t *digest, hipStream_t st);
hipError_t launch_frame_layout(const int32_t *result, size_t n, uint32_t bs, size_t nb,
                               const uint32_t *xxh, uint64_t first_off, uint64_t *entry_off,
                               uint32_t *size_words, hipStream_t st);
hipError_t launch_frame_index(const uint8_t *frame, size_t n, uint64_t first_off, uint32_t bs,
                              int bsum, size_t max_blocks, uint64_t *off, uint32_t *sw,
                              uint32_t *want, int64_t *info, hipStream_t st);
hipError_t launch_pack_blocks(const uint8_t *slots, size_t stride, const int32_t *out_len,
                              const uint64_t *flat_off, size_t nb, uint8_t *flat, hipStream_t st);
hipError_t launch_frame_scatter(const uint8_t *src, size_t n, uint32_t bs, const uint8_t *comp,
                                size_t comp_stride, const int32_t *result, const uint32_t *xxh,
                                const uint64_t *entry_off, size_t nb, uint8_t *frame,
                                uint64_t frame_base, hipStream_t st);

}  // namespace lz4hip
