"""Device-resident batched codec: torch tensors in HBM in, torch tensors out.

torch is plumbing here (device memory, streams); every byte of compute happens
in liblz4hip.so through the C ABI of include/lz4hip.h.  A "block" is one
independent LZ4 block of the frame path (lz4stream/block.go:224-252,319-340).
"""
import torch

from . import _lib


def _stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def _u8(t):
    if t.dtype != torch.uint8 or not t.is_cuda or not t.is_contiguous():
        raise TypeError("expected a contiguous CUDA uint8 tensor")
    return t


def n_blocks(n, block_size):
    return (n + block_size - 1) // block_size


_ws_cache = {}


def fast_workspace(nb, device):
    """Cached hash-table workspace of the Fast compressor (None when the batch is small
    enough for the team kernel, whose tables live in LDS).  One per (device, stream): the workspace holds the
    tables and the block counter of a launch, so two launches that may overlap -- different
    streams -- must not share one.  A workspace that is outgrown is kept alive until the
    stream it served has drained (record_stream), never freed under a running kernel."""
    need = _lib.load().lz4hip_fast_workspace_bytes(nb)
    if need == 0:
        return None
    stream = torch.cuda.current_stream(device)
    key = (device.type, device.index, stream.cuda_stream)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < need:
        if ws is not None:
            ws.record_stream(stream)
        ws = torch.empty(need, dtype=torch.uint8, device=device)
        _ws_cache[key] = ws
    return ws


def compress_blocks(src, block_size, level=0, dst_mode=_lib.DST_FRAME, out=None, result=None,
                    workspace=None):
    """Compress every block_size piece of `src` (1-D CUDA uint8, 16-byte aligned).

    level 0 = Fast (block.go:105-295), else HC chain depth (block.go:319-481).
    Returns (comp [nb, stride] uint8, result [nb] int32) with the Go return
    convention per block: n > 0, 0 = incompressible, -1 = short buffer.
    """
    _u8(src)
    lib = _lib.load()
    n = src.numel()
    nb = n_blocks(n, block_size)
    stride = block_size if dst_mode == _lib.DST_FRAME else (lib.lz4hip_compress_bound(block_size) + 15) // 16 * 16
    if out is None:
        out = torch.empty((max(nb, 1), stride), dtype=torch.uint8, device=src.device)
    if result is None:
        result = torch.zeros(max(nb, 1), dtype=torch.int32, device=src.device)
    if level == 0:
        if workspace is None:
            workspace = fast_workspace(nb, src.device)
        ws_ptr, ws_len = (workspace.data_ptr(), workspace.numel()) if workspace is not None else (None, 0)
        _lib.check(lib.lz4hip_compress_fast_blocks(src.data_ptr(), n, block_size, dst_mode,
                                                   out.data_ptr(), stride, result.data_ptr(),
                                                   ws_ptr, ws_len, _stream_ptr()))
    else:
        if workspace is None:  # a caller's own (possibly smaller) workspace is the ABI's to judge
            workspace = torch.empty(lib.lz4hip_hc_workspace_bytes(nb), dtype=torch.uint8,
                                    device=src.device)
        _lib.check(lib.lz4hip_compress_hc_blocks(src.data_ptr(), n, block_size, level, dst_mode,
                                                 out.data_ptr(), stride, result.data_ptr(),
                                                 workspace.data_ptr(), workspace.numel(),
                                                 _stream_ptr()))
    return out[:nb], result[:nb]


def uncompress_blocks(src, src_off, size_words, dst_len, out=None, out_len=None, dst_stride=None,
                      dict=None):
    """Decode n independent blocks (block.go:44-52 per block).

    src: CUDA uint8 holding the payloads; src_off int64/uint64 [nb] byte offsets;
    size_words int32/uint32 [nb] (bit 31 = stored raw).  dict: optional CUDA uint8
    dictionary shared by every block (lz4.UncompressBlockWithDict, lz4.go:47-49).
    Returns (out [nb, stride], out_len [nb] int32; -1 = ErrInvalidSourceShortBuffer;
    a block whose span leaves `src` is rejected without being read).
    """
    _u8(src)
    lib = _lib.load()
    nb = size_words.numel()
    stride = dst_stride or dst_len
    if out is None:
        out = torch.empty((max(nb, 1), stride), dtype=torch.uint8, device=src.device)
    if out_len is None:
        out_len = torch.empty(max(nb, 1), dtype=torch.int32, device=src.device)
    if dict is not None:
        _u8(dict)
    _lib.check(lib.lz4hip_uncompress_blocks(src.data_ptr(), src.numel(), src_off.data_ptr(),
                                            size_words.data_ptr(), nb, out.data_ptr(), stride,
                                            dst_len, dict.data_ptr() if dict is not None else None,
                                            dict.numel() if dict is not None else 0,
                                            out_len.data_ptr(), _stream_ptr()))
    return out[:nb], out_len[:nb]


def xxh32_blocks(base, stride, n_spans, n=None, off=None, length=None, out=None):
    """xxh32.ChecksumZero of n_spans spans of `base` (xxh32zero.go:142-184)."""
    _u8(base)
    lib = _lib.load()
    if out is None:
        out = torch.empty(max(n_spans, 1), dtype=torch.int32, device=base.device)
    _lib.check(lib.lz4hip_xxh32_blocks(base.data_ptr(), base.numel() if n is None else n, stride,
                                       off.data_ptr() if off is not None else None,
                                       length.data_ptr() if length is not None else None,
                                       n_spans, out.data_ptr(), _stream_ptr()))
    return out[:n_spans]


def xxh32_stream(src, n=None):
    """Content checksum of the first n bytes of src (XXHZero, xxh32zero.go:58-139)."""
    _u8(src)
    out = torch.empty(1, dtype=torch.int32, device=src.device)
    _lib.check(_lib.load().lz4hip_xxh32_stream(src.data_ptr(), src.numel() if n is None else n,
                                               out.data_ptr(), _stream_ptr()))
    return out


def xxh32_stream_update(src, n, state, first, final):
    """Chained content checksum over block-range shards; state: int32[6] on device.
    Returns the digest tensor (int32[1]) when final, else None."""
    digest = torch.empty(1, dtype=torch.int32, device=state.device) if final else None
    _lib.check(_lib.load().lz4hip_xxh32_stream_update(
        src.data_ptr() if n else None, n, state.data_ptr(), int(first), int(final),
        digest.data_ptr() if final else None, _stream_ptr()))
    return digest


def xxh32_stream_update_host(src, n, state, first, final):
    """The same chained checksum with the serial chain on a host core (the bytes come
    down through pinned buffers); state: HOST int32[6] tensor.  Returns the digest as an
    int when final, else None."""
    import ctypes
    digest = ctypes.c_uint32()
    _lib.check(_lib.load().lz4hip_xxh32_stream_update_host(
        src.data_ptr() if n else None, n, state.data_ptr(), int(first), int(final),
        ctypes.addressof(digest) if final else None, _stream_ptr()))
    return digest.value if final else None


def xxh32_host_update(h_src, n, state, first, final):
    """One step of the chained checksum over bytes that already are in HOST memory (h_src:
    CPU uint8 tensor, e.g. a shard staged into pinned memory); state: HOST int32[6]."""
    import ctypes
    digest = ctypes.c_uint32()
    _lib.check(_lib.load().lz4hip_xxh32_host_update(
        h_src.data_ptr() if n else None, n, state.data_ptr(), int(first), int(final),
        ctypes.addressof(digest) if final else None))
    return digest.value if final else None


def frame_index(frame, n, first_off, block_size, block_checksums, max_blocks):
    """Block index of a standard frame in device memory (lz4stream/block.go:279-317).
    Returns (src_off int64 [max_blocks], size_words int32, want_xxh int32, info int64[4]);
    info = [blocks, offset behind the end mark, status, max_blocks too small]."""
    _u8(frame)
    dev = frame.device
    off = torch.empty(max(max_blocks, 1), dtype=torch.int64, device=dev)
    sw = torch.empty(max(max_blocks, 1), dtype=torch.int32, device=dev)
    want = torch.zeros(max(max_blocks, 1), dtype=torch.int32, device=dev)
    info = torch.zeros(4, dtype=torch.int64, device=dev)
    _lib.check(_lib.load().lz4hip_frame_index(frame.data_ptr(), n, first_off, block_size,
                                              int(bool(block_checksums)), max_blocks, off.data_ptr(),
                                              sw.data_ptr(), want.data_ptr(), info.data_ptr(),
                                              _stream_ptr()))
    return off, sw, want, info


def pack_blocks(slots, stride, out_len, flat_off, nb, flat):
    """Copies block b's out_len[b] bytes from slots + b*stride to flat + flat_off[b]."""
    _lib.check(_lib.load().lz4hip_pack_blocks(slots.data_ptr(), stride, out_len.data_ptr(),
                                              flat_off.data_ptr(), nb, flat.data_ptr(), _stream_ptr()))
    return flat


def frame_layout(result, n, block_size, nb, block_xxh=None, first_off=0):
    """Exclusive scan of entry sizes (lz4stream/block.go:254-276).
    Returns (entry_off [nb+1] int64, size_words [nb] int32)."""
    dev = result.device
    entry_off = torch.empty(nb + 1, dtype=torch.int64, device=dev)
    size_words = torch.empty(max(nb, 1), dtype=torch.int32, device=dev)
    _lib.check(_lib.load().lz4hip_frame_layout(result.data_ptr(), n, block_size, nb,
                                               block_xxh.data_ptr() if block_xxh is not None else None,
                                               first_off, entry_off.data_ptr(),
                                               size_words.data_ptr(), _stream_ptr()))
    return entry_off, size_words[:nb]


def frame_scatter(src, block_size, comp, result, entry_off, nb, frame, frame_base=0,
                  block_xxh=None, n=None):
    _lib.check(_lib.load().lz4hip_frame_scatter(
        src.data_ptr(), src.numel() if n is None else n, block_size, comp.data_ptr(),
        comp.stride(0) if comp.dim() == 2 else block_size, result.data_ptr(),
        block_xxh.data_ptr() if block_xxh is not None else None, entry_off.data_ptr(), nb,
        frame.data_ptr(), frame_base, _stream_ptr()))
    return frame


def frame_compress(src, block_size=1 << 22, flags=_lib.F_CONTENT_CHECKSUM, level=0,
                   content_size=0):
    """Whole LZ4 frame of a device-resident `src`, assembled on device.
    Equivalent to Writer.Write(src)+Close() (writer.go:74-184) or io.Copy with
    F_READFROM.  Returns a CUDA uint8 tensor holding exactly the frame."""
    _u8(src)
    lib = _lib.load()
    n = src.numel()
    nb_data = n_blocks(n, block_size)
    nb = nb_data + (1 if (flags & _lib.F_READFROM) and n % block_size == 0 else 0)
    bsum = bool(flags & _lib.F_BLOCK_CHECKSUM)
    hdr = torch.zeros(16, dtype=torch.uint8)
    hlen = lib.lz4hip_frame_header(hdr.data_ptr(), block_size, flags, content_size)
    result = torch.zeros(nb + 1, dtype=torch.int32, device=src.device)
    if n == 0:  # keep device pointers non-NULL for the empty stream
        src = torch.zeros(16, dtype=torch.uint8, device=src.device)
    comp, _ = compress_blocks(src[:n], block_size, level, _lib.DST_FRAME, result=result)
    xxh = xxh32_blocks(src, block_size, nb, n=n) if bsum else None
    cs = xxh32_stream(src, n) if flags & _lib.F_CONTENT_CHECKSUM else None
    entry_off, _ = frame_layout(result, n, block_size, nb, xxh, hlen)
    end = int(entry_off[nb].item())
    frame = torch.empty(end + 8, dtype=torch.uint8, device=src.device)
    frame[:hlen] = hdr[:hlen].to(src.device)
    frame_scatter(src, block_size, comp, result, entry_off, nb, frame, 0, xxh, n=n)
    trl = torch.zeros(8, dtype=torch.uint8)
    tlen = lib.lz4hip_frame_trailer(trl.data_ptr(), flags,
                                    (int(cs.item()) & 0xFFFFFFFF) if cs is not None else 0)
    frame[end:end + tlen] = trl[:tlen].to(src.device)
    return frame[:end + tlen]
