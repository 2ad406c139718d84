"""N>1 path on CPU: world_size-2 (and 3) gloo rehearsal of lz4_amd.shard.

The exchange logic (block-range partition, span-length all-gather, checksum
state chain, span gather) is the product code; the per-rank compute is swapped
for oracle-backed CPU ops (test infrastructure) because there is no GPU here.
The assembled frame must equal the single-process oracle frame byte for byte."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class CpuOps:
    device = torch.device("cpu")

    def __init__(self):
        from oracle import pyoracle
        self.o = pyoracle

    def compress_blocks(self, src, bs, level):
        pl, sw, _ = self.o.compress_blocks(src.numpy(), bs, level, 1, False)
        res = np.where(sw >> 31, 0, sw & 0x7FFFFFFF).astype(np.int32)
        return torch.from_numpy(pl.copy()), torch.from_numpy(res)

    def xxh32_blocks(self, src, bs, nb, n):
        a = src.numpy()
        out = [self.o.ChecksumZero(a[b * bs:min(n, (b + 1) * bs)]) for b in range(nb)]
        return torch.from_numpy(np.array(out, dtype=np.uint32).view(np.int32))

    def frame_layout(self, result, n, bs, nb, xxh, first_off):
        r = result.numpy()
        e, sw = [], []
        for b in range(nb):
            ln = max(0, min(bs, n - b * bs))
            payload = int(r[b]) if r[b] > 0 else ln
            e.append(4 + payload + (4 if xxh is not None and int(xxh[b]) != 0 else 0))
            sw.append(payload if r[b] > 0 else payload | 0x80000000)
        off = np.concatenate([[0], np.cumsum(e)]).astype(np.int64) + first_off
        return torch.from_numpy(off), torch.from_numpy(np.array(sw, dtype=np.uint32).view(np.int32))

    def frame_scatter(self, src, n, bs, comp, result, entry_off, nb, span, base, xxh):
        r, s = result.numpy(), src.numpy()
        for b in range(nb):
            at = int(entry_off[b]) - base
            ln = max(0, min(bs, n - b * bs))
            if r[b] > 0:
                payload, word = comp[b, :r[b]].numpy(), int(r[b])
            else:
                payload, word = s[b * bs:b * bs + ln], ln | 0x80000000
            span[at:at + 4] = torch.from_numpy(np.frombuffer(np.uint32(word).tobytes(), dtype=np.uint8).copy())
            span[at + 4:at + 4 + len(payload)] = torch.from_numpy(payload.copy())
            if xxh is not None and int(xxh[b]) != 0:
                c = np.frombuffer(np.int32(int(xxh[b])).tobytes(), dtype=np.uint8).copy()
                span[at + 4 + len(payload):at + 8 + len(payload)] = torch.from_numpy(c)

    def xxh32_update(self, src, n, state, first, final):
        h = self.o.XXHZero()
        if not first:
            h._s[0:16] = np.frombuffer(state[:4].numpy().tobytes(), dtype=np.uint8)
            h._s[16:24] = np.frombuffer(state[4:6].numpy().tobytes(), dtype=np.uint8)
        h.Write(src.numpy()[:n])
        state[:4] = torch.from_numpy(h._s[0:16].copy().view(np.int32))
        state[4:6] = torch.from_numpy(h._s[16:24].copy().view(np.int32))
        return int(h.Sum32()) if final else None

    # ---- read side ----
    def parse_header(self, head):
        from lz4_amd.shard import DeviceOps      # host-only helper of the library: no GPU needed
        return DeviceOps.parse_header(self, head)

    def frame_index(self, frame, n, first_off, bs, bsum, max_blocks):
        f = frame.numpy()
        off, sw, want = [], [], []
        si, status, full = first_off, 0, 0
        while True:
            if n - si < 4:
                status = -4
                break
            w = int(np.frombuffer(f[si:si + 4].tobytes(), dtype=np.uint32)[0])
            if w == 0:
                si += 4
                break
            size = w & 0x7FFFFFFF
            if size > bs:
                status = -6
                break
            if n - si - 4 < size or (bsum and n - si - 4 - size < 4):
                status = -4
                break
            if len(off) == max_blocks:
                full = 1
                break
            off.append(si + 4)
            sw.append(w)
            si += 4 + size
            if bsum:
                want.append(int(np.frombuffer(f[si:si + 4].tobytes(), dtype=np.uint32)[0]))
                si += 4
        m = max(max_blocks, 1)

        def pad(a, dt):
            return np.array(a + [0] * (m - len(a)), dtype=dt)
        return (torch.from_numpy(pad(off, np.int64)), torch.from_numpy(pad(sw, np.uint32).view(np.int32)),
                torch.from_numpy(pad(want, np.uint32).view(np.int32)),
                torch.tensor([len(off), si, status, full], dtype=torch.int64))

    def uncompress_blocks(self, src, src_off, sw, bs):
        s, nb = src.numpy(), sw.numel()
        out = np.zeros((max(nb, 1), bs), dtype=np.uint8)
        ol = np.zeros(max(nb, 1), dtype=np.int32)
        for b in range(nb):
            w = int(sw[b]) & 0xFFFFFFFF
            size, o = w & 0x7FFFFFFF, int(src_off[b])
            blk = s[o:o + size]
            if w >> 31:
                out[b, :size] = blk
                ol[b] = size
            else:
                try:
                    d = self.o.UncompressBlock(blk, bs)
                    out[b, :len(d)] = np.frombuffer(d, dtype=np.uint8)
                    ol[b] = len(d)
                except self.o.ErrInvalidSourceShortBuffer:
                    ol[b] = -1
        return torch.from_numpy(out[:nb]), torch.from_numpy(ol[:nb])

    def xxh32_decoded(self, out, bs, nb, out_len):
        a = out.numpy()
        x = [self.o.ChecksumZero(a[b, :max(int(out_len[b]), 0)]) for b in range(nb)]
        return torch.from_numpy(np.array(x, dtype=np.uint32).view(np.int32))

    def pack_blocks(self, out, bs, out_len, flat_off, nb, flat):
        for b in range(nb):
            k = int(out_len[b])
            if k > 0:
                flat[int(flat_off[b]):int(flat_off[b]) + k] = out[b, :k]
        return flat

    def header(self, bs, flags, content_size):
        return torch.from_numpy(np.frombuffer(self.o.frame_header(bs, flags & 7, content_size),
                                              dtype=np.uint8).copy())


def _worker(rank, world, port, n, bs, flags, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lz4_amd import corpus, shard
    data = corpus.generate("mixed", 5, n) if n else np.zeros(0, dtype=np.uint8)
    nb = (n + bs - 1) // bs
    lo, hi = shard.block_range(nb, rank, world)
    local = torch.from_numpy(data[min(n, lo * bs):min(n, hi * bs)].copy())
    frame = shard.frame_compress_sharded(local, n, bs, flags, 0, ops=CpuOps())
    if rank == 0:
        q.put(frame.numpy().tobytes())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n,flags", [(2, 5 * 65536 + 999, 3), (2, 4 * 65536, 3 | 8),
                                           (3, 2 * 65536 + 1, 2), (2, 1000, 1), (2, 0, 2 | 8)])
def test_sharded_frame_equals_single_process_frame(world, n, flags):
    from lz4_amd import corpus
    from oracle import pyoracle as o
    bs = 1 << 16
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + n + flags) % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, bs, flags, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    data = corpus.generate("mixed", 5, n) if n else np.zeros(0, dtype=np.uint8)
    assert got == o.frame_compress(data, bs, flags, 0)


def _reader_worker(rank, world, port, frame, gather, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lz4_amd import shard
    f = torch.from_numpy(np.frombuffer(frame, dtype=np.uint8).copy()) if rank == 0 else None
    try:
        flat, start, total = shard.frame_uncompress_sharded(f, len(frame), ops=CpuOps(), gather=gather)
        q.put((rank, "ok", flat.numpy().tobytes(), start, total))
    except shard.ShardError as e:
        q.put((rank, "err", e.code, e.block, 0))
    dist.barrier()
    dist.destroy_process_group()


def _run_reader(world, frame, gather=False):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() + len(frame) + world) % 2000
    procs = [ctx.Process(target=_reader_worker, args=(r, world, port, frame, gather, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return got


@pytest.mark.parametrize("world,n,flags", [(2, 5 * 65536 + 999, 3), (3, 7 * 65536, 3 | 8), (2, 3 * 65536 + 5, 0),
                                           (3, 1000, 2), (2, 0, 2)])
def test_sharded_reader_equals_source(world, n, flags):
    """frame_uncompress_sharded: the ranks' pieces, in rank order, are the source."""
    from lz4_amd import corpus
    from oracle import pyoracle as o
    bs = 1 << 16
    data = corpus.generate("mixed", 9, n).tobytes() if n else b""
    frame = o.frame_compress(data, bs, flags, 0)
    got = _run_reader(world, frame)
    assert all(g[1] == "ok" for g in got), got
    assert b"".join(g[2] for g in got) == data
    assert [g[3] for g in got] == [sum(len(h[2]) for h in got[:i]) for i in range(world)]
    assert all(g[4] == len(data) for g in got)
    whole = _run_reader(world, frame, gather=True)
    assert whole[0][2] == data


def test_sharded_reader_first_error_in_stream_order_wins():
    """lz4stream/block.go:187-194: a bad checksum in block 1 (rank 0's) beats a corrupt
    payload in block 4 (rank 1's) and a truncated tail; every rank raises the same verdict."""
    import struct
    from lz4_amd import _lib, corpus
    from oracle import pyoracle as o
    bs = 1 << 16
    data = corpus.generate("text", 3, 6 * bs).tobytes()
    frame = bytearray(o.frame_compress(data, bs, 3, 0))
    offs, si = [], 7
    while struct.unpack_from("<I", frame, si)[0]:
        offs.append(si)
        si += 8 + (struct.unpack_from("<I", frame, si)[0] & 0x7FFFFFFF)
    bad = bytearray(frame)
    bad[offs[4] + 4:offs[4] + 8] = b"\xff\xff\xff\xff"
    bad[offs[2] - 1] ^= 1                                  # checksum word of block 1
    got = _run_reader(2, bytes(bad))
    assert [g[1:4] for g in got] == [("err", _lib.ERR_BLOCK_CHECKSUM, 1)] * 2
    bad = bytearray(frame)
    bad[offs[4] + 4:offs[4] + 8] = b"\xff\xff\xff\xff"      # only the corrupt payload
    got = _run_reader(2, bytes(bad))
    assert all(g[1] == "err" and g[3] == 4 for g in got), got
    got = _run_reader(2, bytes(frame[:offs[5] + 100]))     # truncated inside block 5: blocks 0-4 fine
    assert [g[1:4] for g in got] == [("err", _lib.ERR_FRAME, 5)] * 2
    bad = bytearray(frame)
    bad[-1] ^= 1                                           # content checksum
    got = _run_reader(3, bytes(bad))
    assert [g[1:3] for g in got] == [("err", _lib.ERR_FRAME_CHECKSUM)] * 3
    # ragged frame: short blocks mid-stream (a Writer that was flushed) -- shards are not
    # stripe multiples and the content checksum takes the gathered path
    parts = [data[:bs], data[bs:bs + 1001], data[bs + 1001:2 * bs + 77], data[2 * bs + 77:3 * bs]]
    f = bytearray(o.frame_header(bs, 2))
    for p in parts:
        k, c = o.CompressBlock(p, len(p)) if len(p) > 64 else (0, b"")
        f += (struct.pack("<I", k) + c) if k else (struct.pack("<I", len(p) | 0x80000000) + p)
    f += struct.pack("<II", 0, o.ChecksumZero(b"".join(parts)))
    got = _run_reader(2, bytes(f))
    assert all(g[1] == "ok" for g in got) and b"".join(g[2] for g in got) == b"".join(parts)


def test_block_range_partition_is_contiguous_and_complete():
    from lz4_amd.shard import block_range
    for nb in (0, 1, 7, 8, 9, 4096, 32768):
        for world in (1, 2, 3, 8):
            r = [block_range(nb, g, world) for g in range(world)]
            assert r[0][0] == 0 and r[-1][1] == nb
            assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1
