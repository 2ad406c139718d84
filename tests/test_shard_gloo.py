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
        if final:
            return torch.from_numpy(np.array([h.Sum32()], dtype=np.uint32).view(np.int32))
        return None

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


def test_block_range_partition_is_contiguous_and_complete():
    from lz4_amd.shard import block_range
    for nb in (0, 1, 7, 8, 9, 4096, 32768):
        for world in (1, 2, 3, 8):
            r = [block_range(nb, g, world) for g in range(world)]
            assert r[0][0] == 0 and r[-1][1] == nb
            assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1
