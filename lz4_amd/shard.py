"""Frame path sharded over the GPUs of one node, one process per GPU.

Frames written by the reference always have independent blocks
(internal/lz4stream/frame.go:139-142), so the block index range shards
naturally: rank g owns the contiguous blocks [g*B/G, (g+1)*B/G).  The only
exchange steps are the ones the frame format forces:
  1. all-gather of each rank's span length (8 bytes per rank) -> span offsets
     (exclusive scan across ranks of the per-block entry sizes);
  2. the XXHZero content-checksum state handed from rank to rank in stream order
     (24 bytes; the hash is serial by construction, lz4stream/block.go:257-259);
  3. the gather of the assembled spans to the rank that owns the frame (RCCL
     send/recv: RCCL has no gatherv; each sender uses its own xGMI link to the root).
Compute (compress, checksums, layout, scatter) is the device path of device.py;
`ops` is injectable so the exchange logic can be rehearsed on CPU with gloo.
"""
import torch
import torch.distributed as dist

from . import _lib


def block_range(n_blocks, rank, world):
    """Contiguous block range of `rank`: [lo, hi)."""
    return n_blocks * rank // world, n_blocks * (rank + 1) // world


class DeviceOps:
    """The product compute path: lz4_amd.device on the current CUDA device."""

    def __init__(self):
        from . import device
        self.d = device

    def compress_blocks(self, src, bs, level):
        return self.d.compress_blocks(src, bs, level)

    def xxh32_blocks(self, src, bs, nb, n):
        return self.d.xxh32_blocks(src, bs, nb, n=n)

    def frame_layout(self, result, n, bs, nb, xxh, first_off):
        return self.d.frame_layout(result, n, bs, nb, xxh, first_off)

    def frame_scatter(self, src, n, bs, comp, result, entry_off, nb, span, base, xxh):
        return self.d.frame_scatter(src, bs, comp, result, entry_off, nb, span, base, xxh, n=n)

    def xxh32_update(self, src, n, state, first, final):
        return self.d.xxh32_stream_update(src, n, state, first, final)

    def header(self, bs, flags, content_size):
        hdr = torch.zeros(16, dtype=torch.uint8)
        k = _lib.load().lz4hip_frame_header(hdr.data_ptr(), bs, flags, content_size)
        return hdr[:k]


def frame_compress_sharded(src_local, n_total, block_size, flags=_lib.F_CONTENT_CHECKSUM, level=0,
                           content_size=0, group=None, ops=None, root=0):
    """Every rank passes the bytes of ITS block range (`src_local`); the root
    returns the whole frame (uint8 tensor on src_local's device), others None.

    Byte-identical to the single-process frame (ordering contract of
    lz4stream/block.go:31-57)."""
    ops = ops or DeviceOps()
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    dev = src_local.device
    nb_data = (n_total + block_size - 1) // block_size
    lo, hi = block_range(nb_data, rank, world)
    n_local = min(n_total, hi * block_size) - min(n_total, lo * block_size)
    assert src_local.numel() == n_local, (src_local.numel(), n_local)
    last = rank == world - 1
    # the trailing empty block of Writer.ReadFrom (writer.go:220-233) belongs to the last rank
    extra = 1 if (last and (flags & _lib.F_READFROM) and n_total % block_size == 0) else 0
    nb = (hi - lo) + extra
    bsum, csum = bool(flags & _lib.F_BLOCK_CHECKSUM), bool(flags & _lib.F_CONTENT_CHECKSUM)

    result = torch.zeros(nb + 1, dtype=torch.int32, device=dev)
    comp = None
    if hi > lo:
        comp, res = ops.compress_blocks(src_local, block_size, level)
        result[:hi - lo] = res
    xxh = ops.xxh32_blocks(src_local, block_size, nb, n_local) if bsum else None

    # 1. span lengths -> span offsets
    off0, _ = ops.frame_layout(result, n_local, block_size, nb, xxh, 0)
    span_len = off0[nb:nb + 1].to(torch.int64).clone()
    all_len = [torch.zeros_like(span_len) for _ in range(world)]
    dist.all_gather(all_len, span_len, group=group)
    lens = [int(t.item()) for t in all_len]
    hdr = ops.header(block_size, flags, content_size)
    first = len(hdr) + sum(lens[:rank])
    entry_off, size_words = ops.frame_layout(result, n_local, block_size, nb, xxh, first)
    span = torch.zeros(lens[rank] + 16, dtype=torch.uint8, device=dev)
    if nb:
        ops.frame_scatter(src_local, n_local, block_size, comp if comp is not None else span,
                          result, entry_off, nb, span, first, xxh)

    # 2. content checksum: state travels rank 0 -> 1 -> ... -> last, digest back to root
    digest = None
    if csum:
        state = torch.zeros(6, dtype=torch.int32, device=dev)
        if rank > 0:
            dist.recv(state, src=rank - 1, group=group)
        d = ops.xxh32_update(src_local, n_local, state, rank == 0, last)
        if not last:
            dist.send(state, dst=rank + 1, group=group)
        if last and rank != root:
            dist.send(d, dst=root, group=group)
        if rank == root:
            if last:
                digest = d
            else:
                digest = torch.zeros(1, dtype=torch.int32, device=dev)
                dist.recv(digest, src=world - 1, group=group)

    # 3. gather the spans at the root
    if rank != root:
        if lens[rank]:
            dist.send(span[:lens[rank]].contiguous(), dst=root, group=group)
        return None
    total = len(hdr) + sum(lens) + 4 + (4 if csum else 0)
    frame = torch.zeros(total, dtype=torch.uint8, device=dev)
    frame[:len(hdr)] = hdr.to(dev)
    at = len(hdr)
    for r in range(world):
        if lens[r]:
            if r == rank:
                frame[at:at + lens[r]] = span[:lens[r]]
            else:
                buf = torch.empty(lens[r], dtype=torch.uint8, device=dev)
                dist.recv(buf, src=r, group=group)
                frame[at:at + lens[r]] = buf
        at += lens[r]
    # end mark + content checksum: frame.go:59-74
    if csum:
        dg = int(digest.item()) & 0xFFFFFFFF
        frame[at + 4:at + 8] = torch.tensor([dg & 0xFF, dg >> 8 & 0xFF, dg >> 16 & 0xFF, dg >> 24],
                                            dtype=torch.uint8, device=dev)
    return frame
