"""Frame path sharded over the GPUs of one node, one process per GPU, both directions.

Frames written by the reference always have independent blocks
(internal/lz4stream/frame.go:139-142), so the block index range shards
naturally: rank g owns the contiguous blocks [g*B/G, (g+1)*B/G).  The only
exchange steps are the ones the frame format forces:

write (Writer, writer.go:74-184; ordered output contract lz4stream/block.go:31-57)
  1. all-gather of each rank's span length (8 bytes per rank) -> span offsets
     (exclusive scan across ranks of the per-block entry sizes);
  2. the gather of the assembled spans to the rank that owns the frame: ONE group of
     point-to-point operations, every sender's span received straight into its slice of
     the frame (RCCL has no gatherv; each sender has its own xGMI link to the root, so
     the 7 transfers run side by side and nothing is copied twice);
  3. the XXHZero content-checksum state handed from rank to rank in stream order
     (24 bytes; the hash is serial by construction, lz4stream/block.go:257-259).

read (Reader, reader.go:81-196; lz4stream/block.go:95-194)
  1. the root indexes the frame's blocks on its GPU (FrameDataBlock.Read per block,
     :279-317) and broadcasts the index (16 bytes per block);
  2. ONE group of point-to-point operations scatters the compressed spans by block range;
  3. every rank decodes and verifies its blocks; a 1-word MIN all-reduce carries the first
     failing block of the stream (first error wins, :187-194);
  4. the content checksum runs over the decoded shards in stream order, state relayed as on
     the write side, and is compared with the frame's (frame.go:117-131).
  The decoded bytes stay sharded (rank g holds its block range) unless gather=True.

Compute is the device path of device.py; `ops` is injectable so that the exchange logic
can be rehearsed on CPU with gloo (tests/test_shard_gloo.py).
"""
import threading

import torch
import torch.distributed as dist

from . import _lib

I64_MAX = (1 << 63) - 1


class ShardError(RuntimeError):
    """A frame error found by the sharded reader: .code is the LZ4HIP_ERR_* of the
    single-process reader, .block the failing block's index in the stream (or None)."""

    def __init__(self, code, block=None):
        super().__init__("lz4 sharded reader: error %d%s" % (code, "" if block is None else " in block %d" % block))
        self.code, self.block = code, block


def block_range(n_blocks, rank, world):
    """Contiguous block range of `rank`: [lo, hi)."""
    return n_blocks * rank // world, n_blocks * (rank + 1) // world


class DeviceOps:
    """The product compute path: lz4_amd.device on the current CUDA device.

    checksum_on: where the serial XXHZero chain of the CONTENT checksum runs -- "host"
    (bytes come down through pinned buffers, a host core steps the chain ~4x faster than a
    GPU lane) or "device" (k_xxh32_stream_update, nothing leaves HBM)."""

    def __init__(self, checksum_on="host"):
        from . import device
        self.d = device
        self.checksum_on = checksum_on
        self.device = torch.device("cuda", torch.cuda.current_device())   # where this rank computes

    def compress_blocks(self, src, bs, level):
        return self.d.compress_blocks(src, bs, level)

    def xxh32_blocks(self, src, bs, nb, n):
        return self.d.xxh32_blocks(src, bs, nb, n=n)

    def frame_layout(self, result, n, bs, nb, xxh, first_off):
        return self.d.frame_layout(result, n, bs, nb, xxh, first_off)

    def frame_scatter(self, src, n, bs, comp, result, entry_off, nb, span, base, xxh):
        return self.d.frame_scatter(src, bs, comp, result, entry_off, nb, span, base, xxh, n=n)

    def xxh32_update(self, src, n, state, first, final):
        """state: CPU int32[6] (XXHZero: v1..v4, totalLen lo, hi), updated in place.
        Returns the digest (int) when final."""
        if self.checksum_on == "host":
            return self.d.xxh32_stream_update_host(src, n, state, first, final)
        st = state.to(src.device)
        dg = self.d.xxh32_stream_update(src, n, st, first, final)
        state.copy_(st.cpu())
        return (int(dg.item()) & 0xFFFFFFFF) if final else None

    def stage_ahead(self, src, n):
        """Starts bringing src[:n] down into pinned host memory on a side stream and returns at
        once.  For a rank that has to wait for the XXHZero state of the ranks before it
        (lz4stream/block.go:257-259: the chain is serial): when the state arrives its turn is
        the hash alone, not download + hash."""
        if self.checksum_on != "host" or not src.is_cuda or n == 0:
            return None
        try:
            return _StageAhead(src, n)
        except RuntimeError:      # no pinned memory to be had: the turn brings the bytes down itself
            return None

    def xxh32_update_staged(self, staged, src, n, state, first, final):
        """xxh32_update over a shard whose first staged.m bytes are in host memory."""
        m = staged.wait()
        rest = n - m
        dg = self.d.xxh32_host_update(staged.buf, m, state, first, final and rest == 0)
        if rest:
            dg = self.d.xxh32_stream_update_host(src[m:], rest, state, False, final)
        return dg

    def header(self, bs, flags, content_size):
        hdr = torch.zeros(16, dtype=torch.uint8)
        k = _lib.load().lz4hip_frame_header(hdr.data_ptr(), bs, flags, content_size)
        return hdr[:k]

    # ---- read side ----
    def parse_header(self, head):
        """head: CPU uint8 tensor with the first bytes of the frame.
        Returns (rc, block_size, flags, header_len)."""
        import ctypes as C
        bs, fl, cs, hl = C.c_uint32(), C.c_uint32(), C.c_uint64(), C.c_size_t()
        rc = _lib.load().lz4hip_frame_parse_header(head.data_ptr(), head.numel(), C.addressof(bs),
                                                   C.addressof(fl), C.addressof(cs), C.addressof(hl))
        return rc, bs.value, fl.value, hl.value

    def frame_index(self, frame, n, first_off, bs, bsum, max_blocks):
        return self.d.frame_index(frame, n, first_off, bs, bsum, max_blocks)

    def uncompress_blocks(self, src, src_off, sw, bs):
        return self.d.uncompress_blocks(src, src_off, sw, bs)

    def xxh32_decoded(self, out, bs, nb, out_len):
        return self.d.xxh32_blocks(out, bs, nb, length=out_len)

    def pack_blocks(self, out, bs, out_len, flat_off, nb, flat):
        return self.d.pack_blocks(out, bs, out_len, flat_off, nb, flat)


_pinned = {}        # device index -> pinned uint8 staging tensor, kept between calls


def reserve_staging(n, device=None):
    """Pinned host memory for stage_ahead, allocated (page-locking gigabytes takes seconds) outside
    the caller's timed region; capped at LZ4_SHARD_STAGE_GIB (default 4) GiB per process -- eight
    ranks of a node pin 32 GiB between them; what lies beyond the cap comes down piecewise in the
    rank's turn, as without staging."""
    import os
    cap = int(float(os.environ.get("LZ4_SHARD_STAGE_GIB", "4")) * (1 << 30)) & ~15
    m = min(n, cap) & ~15 if n > cap else n
    key = torch.cuda.current_device() if device is None else device.index
    buf = _pinned.get(key)
    if buf is None or buf.numel() < m:
        _pinned.pop(key, None)
        buf = torch.empty(max(m, 16), dtype=torch.uint8, pin_memory=True)
        _pinned[key] = buf
    return buf, m


class _StageAhead:
    def __init__(self, src, n):
        self.buf, self.m = reserve_staging(n, src.device)
        self.side = torch.cuda.Stream(device=src.device)
        self.side.wait_stream(torch.cuda.current_stream(src.device))     # src is complete there
        with torch.cuda.stream(self.side):
            self.buf[:self.m].copy_(src[:self.m], non_blocking=True)
        self.done = torch.cuda.Event()
        self.done.record(self.side)

    def wait(self):
        self.done.synchronize()
        return self.m


class _Wire:
    """Point-to-point plumbing.  With RCCL ("nccl") device tensors travel as they are; with
    gloo (CPU rehearsal, or several ranks rehearsing on ONE GPU) device tensors are staged
    through the host."""

    def __init__(self, group):
        self.group = group
        self.staged = dist.get_backend(group) != "nccl"
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)

    def small(self, t, dev):
        """A small tensor in the form collectives can carry."""
        return t.cpu() if self.staged else t.to(dev)

    def exchange(self, sends, recvs):
        """sends: [(tensor, dst)], recvs: [(tensor slice to fill, src)] -- one group."""
        ops, fix = [], []
        for t, dst in sends:
            w = t.cpu().contiguous() if self.staged and t.is_cuda else t.contiguous()
            ops.append(dist.P2POp(dist.isend, w, dst, self.group))
        for t, src in recvs:
            if self.staged and t.is_cuda:
                tmp = torch.empty(t.numel(), dtype=t.dtype)
                fix.append((t, tmp))
                ops.append(dist.P2POp(dist.irecv, tmp, src, self.group))
            else:
                assert t.is_contiguous()
                ops.append(dist.P2POp(dist.irecv, t, src, self.group))
        if ops:
            for r in dist.batch_isend_irecv(ops):
                r.wait()
        for t, tmp in fix:
            t.copy_(tmp)

    def all_gather_i64(self, value, dev):
        """One int64 per rank -> python list (one small collective, one host read)."""
        mine = self.small(torch.tensor([value], dtype=torch.int64), dev)
        parts = [torch.zeros_like(mine) for _ in range(self.world)]
        dist.all_gather(parts, mine, group=self.group)
        return [int(x) for x in torch.cat(parts).cpu().tolist()]

    def relay_state(self, state, first_rank, last_rank, step):
        """Runs `step(state) -> digest|None` on every rank in rank order, handing the CPU
        int32[6] state from rank to rank; returns the last rank's digest on every rank."""
        dev = torch.device("cuda", torch.cuda.current_device()) if not self.staged else None
        if self.rank > first_rank:
            buf = self.small(state, dev)
            dist.recv(buf, src=self.rank - 1, group=self.group)
            state.copy_(buf.cpu())
        digest = step(state)
        if self.rank < last_rank:
            dist.send(self.small(state, dev), dst=self.rank + 1, group=self.group)
        dg = torch.tensor([digest if digest is not None else 0], dtype=torch.int64)
        dg = self.small(dg, dev)
        dist.broadcast(dg, src=last_rank, group=self.group)
        return int(dg.item()) & 0xFFFFFFFF


class _HashAhead(threading.Thread):
    """The head of the content-checksum chain.  The first rank needs no state from anyone, so
    its share is hashed from the start of the call, beside its own compression (on the product
    path: a side stream brings the bytes down, a host core steps the chain, the GIL is free
    during the call)."""

    def __init__(self, ops, src, n, final):
        super().__init__(daemon=True)
        self.ops, self.src, self.n, self.final = ops, src, n, final
        self.state = torch.zeros(6, dtype=torch.int32)
        self.digest, self.err, self.side = None, None, None
        if src.is_cuda:
            self.side = torch.cuda.Stream(device=src.device)
            self.side.wait_stream(torch.cuda.current_stream(src.device))   # src is complete there
        self.start()

    def run(self):
        try:
            if self.side is not None:
                torch.cuda.set_device(self.src.device)
                with torch.cuda.stream(self.side):
                    self.digest = self.ops.xxh32_update(self.src, self.n, self.state, True, self.final)
                    self.side.synchronize()
            else:
                self.digest = self.ops.xxh32_update(self.src, self.n, self.state, True, self.final)
        except BaseException as e:   # handed to the caller's thread by result()
            self.err = e

    def result(self, state):
        self.join()
        if self.err is not None:
            raise self.err
        state.copy_(self.state)
        return self.digest


class _Clock:
    """Phase timer for bench.py: stats[name] += seconds between marks.  Synchronises the
    device at every mark, so it is only used when a caller asks for stats."""

    def __init__(self, stats, dev):
        import time
        self.t, self.stats, self.cuda = time.perf_counter, stats, dev is not None and dev.type == "cuda"
        self.last = self._now() if stats is not None else 0.0

    def _now(self):
        if self.cuda:
            torch.cuda.synchronize()
        return self.t()

    def mark(self, name):
        if self.stats is None:
            return
        now = self._now()
        self.stats[name] = self.stats.get(name, 0.0) + now - self.last
        self.last = now


def frame_compress_sharded(src_local, n_total, block_size, flags=_lib.F_CONTENT_CHECKSUM, level=0,
                           content_size=0, group=None, ops=None, root=0, stats=None):
    """Every rank passes the bytes of ITS block range (`src_local`); the root
    returns the whole frame (uint8 tensor on src_local's device), others None.

    Byte-identical to the single-process frame (ordering contract of
    lz4stream/block.go:31-57).  stats (optional dict): seconds per phase -- compute_s,
    gather_s (+ gather_bytes at the root), checksum_s."""
    ops = ops or DeviceOps()
    wire = _Wire(group)
    clock = _Clock(stats, src_local.device)
    rank, world = wire.rank, wire.world
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

    ahead = _HashAhead(ops, src_local, n_local, last) if csum and rank == 0 else None
    # the other ranks wait for the state of those before them: their bytes go down meanwhile
    staged = ops.stage_ahead(src_local, n_local) if csum and rank > 0 and hasattr(ops, "stage_ahead") else None
    result = torch.zeros(nb + 1, dtype=torch.int32, device=dev)
    comp = None
    if hi > lo:
        comp, res = ops.compress_blocks(src_local, block_size, level)
        result[:hi - lo] = res
    xxh = ops.xxh32_blocks(src_local, block_size, nb, n_local) if bsum else None

    # 1. span lengths -> span offsets (one small all-gather, one host read)
    off0, _ = ops.frame_layout(result, n_local, block_size, nb, xxh, 0)
    mine = wire.small(off0[nb:nb + 1].to(torch.int64).clone(), dev)
    parts = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    lens = [int(x) for x in torch.cat(parts).cpu().tolist()]
    hdr = ops.header(block_size, flags, content_size)
    first = len(hdr) + sum(lens[:rank])
    entry_off = off0 + first

    # 2. entries: the root writes its own straight into the frame, the others into a span;
    #    then one group of transfers, every span landing in its slice of the frame
    total = len(hdr) + sum(lens) + 4 + (4 if csum else 0)
    if rank == root:
        frame = torch.zeros(total + 16, dtype=torch.uint8, device=dev)
        frame[:len(hdr)] = hdr.to(dev)
        target, base = frame, 0
    else:
        frame = None
        target, base = torch.zeros(lens[rank] + 16, dtype=torch.uint8, device=dev), first
    if nb:
        ops.frame_scatter(src_local, n_local, block_size, comp if comp is not None else target,
                          result, entry_off, nb, target, base, xxh)
    clock.mark("compute_s")
    if rank == root:
        at = len(hdr)
        recvs = []
        for r in range(world):
            if r != root and lens[r]:
                recvs.append((frame[at:at + lens[r]], r))
            at += lens[r]
        wire.exchange([], recvs)
        if stats is not None:
            stats["gather_bytes"] = stats.get("gather_bytes", 0) + sum(lens) - lens[root]
    elif lens[rank]:
        wire.exchange([(target[:lens[rank]], root)], [])
    clock.mark("gather_s")

    # 3. content checksum: state travels rank 0 -> 1 -> ... -> last (frame.go:69-71)
    if csum:
        state = torch.zeros(6, dtype=torch.int32)
        def step(st):
            if ahead is not None:
                return ahead.result(st)
            if staged is not None:
                return ops.xxh32_update_staged(staged, src_local, n_local, st, False, last)
            return ops.xxh32_update(src_local, n_local, st, rank == 0, last)
        dg = wire.relay_state(state, 0, world - 1, step)
        if rank == root:
            at = total - 4
            frame[at:at + 4] = torch.tensor([dg & 0xFF, dg >> 8 & 0xFF, dg >> 16 & 0xFF, dg >> 24],
                                            dtype=torch.uint8).to(dev)
    clock.mark("checksum_s")
    return frame[:total] if rank == root else None


def frame_uncompress_sharded(frame, n, group=None, ops=None, root=0, gather=False, stats=None):
    """Sharded Reader.  The root passes the frame (uint8 tensor, n bytes); the other ranks
    pass None.  Every rank returns (decoded bytes of ITS block range, offset of that range in
    the stream, total decoded length); with gather=True the root's first element is the
    whole decoded stream.  Raises ShardError (the same verdict on every rank) where the
    single-process reader fails: first failing block in stream order wins."""
    ops = ops or DeviceOps()
    wire = _Wire(group)
    rank, world = wire.rank, wire.world
    cdev = torch.device("cuda", torch.cuda.current_device()) if not wire.staged else torch.device("cpu")
    clock = _Clock(stats, torch.device("cuda") if torch.cuda.is_available() else None)
    # ---- root: headers (frame.go:80-111,169-200) and the block index ----
    meta = torch.zeros(8, dtype=torch.int64)
    idx = None
    if rank == root:
        head = frame[:min(n, 32)].cpu().contiguous()
        rc, bs, flags, hlen = ops.parse_header(head)
        if rc == _lib.ERR_SHORT:
            rc = _lib.ERR_FRAME
        if rc == 0 and flags & (_lib.F_LEGACY | _lib.F_LINKED):
            rc = _lib.ERR_ARG      # serial formats: reader.go:86-90; not a sharded job
        nb = end = status = 0
        if rc == 0:
            cap = max(1, min((n - hlen) // 4, 1 << 16))
            while True:
                idx = ops.frame_index(frame, n, hlen, bs, bool(flags & _lib.F_BLOCK_CHECKSUM), cap)
                nb, end, status, full = [int(x) for x in idx[3].cpu().tolist()]
                if not full:
                    break
                cap *= 8
        want_cs = 0
        if rc == 0 and status == 0 and flags & _lib.F_CONTENT_CHECKSUM:
            if n - end < 4:
                status = _lib.ERR_FRAME
            else:
                w = frame[end:end + 4].cpu().tolist()
                want_cs = w[0] | w[1] << 8 | w[2] << 16 | w[3] << 24
        meta = torch.tensor([rc, bs, flags, nb, end, status, want_cs, hlen], dtype=torch.int64)
    meta = wire.small(meta, cdev)
    dist.broadcast(meta, src=root, group=group)
    rc, bs, flags, nb, end, status, want_cs, hlen = [int(x) for x in meta.cpu().tolist()]
    if rc != 0:
        raise ShardError(rc)
    bsum, csum = bool(flags & _lib.F_BLOCK_CHECKSUM), bool(flags & _lib.F_CONTENT_CHECKSUM)
    dev = frame.device if rank == root else ops.device

    # ---- the index travels to every rank (16 bytes per block) ----
    if rank == root:
        off, sw, want = idx[0][:nb].contiguous(), idx[1][:nb].contiguous(), idx[2][:nb].contiguous()
    else:
        off = torch.zeros(nb, dtype=torch.int64, device=dev)
        sw = torch.zeros(nb, dtype=torch.int32, device=dev)
        want = torch.zeros(nb, dtype=torch.int32, device=dev)
    if nb:
        for t in ((off, sw, want) if bsum else (off, sw)):
            w = wire.small(t, cdev)
            dist.broadcast(w, src=root, group=group)
            if w is not t:
                t.copy_(w)
    off_h = off.cpu()
    # where the entries of rank g start and end in the frame (size word included)
    ranges = [block_range(nb, g, world) for g in range(world)]
    stop = (end - 4) if status == 0 else end      # end mark, or where the index gave up

    def span(g):
        lo, hi = ranges[g]
        if hi == lo:
            return 0, 0
        a = int(off_h[lo]) - 4
        b = (int(off_h[hi]) - 4) if hi < nb else stop
        return a, b

    clock.mark("index_s")
    # ---- one group of transfers: compressed spans by block range ----
    lo, hi = ranges[rank]
    a, b = span(rank)
    if rank == root:
        local = frame[a:b]
        sends = []
        for g in range(world):
            ga, gb = span(g)
            if g != root and gb > ga:
                sends.append((frame[ga:gb], g))
        wire.exchange(sends, [])
    else:
        local = torch.zeros(b - a + 16, dtype=torch.uint8, device=dev)[:b - a]
        if b > a:
            wire.exchange([], [(local, root)])

    clock.mark("scatter_s")
    # ---- decode + verify this rank's blocks; first failing block of the stream wins ----
    nbl = hi - lo
    first_bad = I64_MAX
    out = out_len = None
    if nbl:
        out, out_len = ops.uncompress_blocks(local, (off[lo:hi] - a).contiguous(), sw[lo:hi].contiguous(), bs)
        bad = out_len < 0
        kind = torch.zeros(nbl, dtype=torch.int64, device=bad.device)     # 0 = invalid block
        if bsum:
            got = ops.xxh32_decoded(out, bs, nbl, out_len)
            mism = (got != want[lo:hi]) & ~bad
            kind = torch.where(mism, torch.ones_like(kind), kind)          # 1 = block checksum
            bad = bad | mism
        if bool(bad.any()):
            i = int(torch.nonzero(bad)[0])
            first_bad = (lo + i) * 2 + int(kind[i])
    fb = wire.small(torch.tensor([first_bad], dtype=torch.int64), cdev)
    dist.all_reduce(fb, op=dist.ReduceOp.MIN, group=group)
    fb = int(fb.item())
    if fb != I64_MAX:
        raise ShardError(_lib.ERR_BLOCK_CHECKSUM if fb & 1 else _lib.ERR_BLOCK, fb >> 1)
    if status != 0:      # the structural error behind the good blocks (stream order)
        raise ShardError(status, nb)

    # ---- this rank's decoded bytes as one contiguous piece of the stream ----
    lens_h = out_len.cpu().tolist() if nbl else []
    n_local = sum(lens_h)
    if nbl and all(x == bs for x in lens_h[:-1]):
        flat = out.reshape(-1)[:n_local]
    elif nbl:
        foff = torch.tensor([sum(lens_h[:i]) for i in range(nbl)], dtype=torch.int64).to(out.device)
        flat = torch.zeros(n_local + 16, dtype=torch.uint8, device=out.device)
        ops.pack_blocks(out, bs, out_len, foff, nbl, flat)
        flat = flat[:n_local]
    else:
        flat = torch.zeros(0, dtype=torch.uint8, device=dev)
    sizes = wire.all_gather_i64(n_local, cdev)
    start, total = sum(sizes[:rank]), sum(sizes)
    clock.mark("compute_s")

    # ---- content checksum over the decoded stream, in stream order (frame.go:117-131) ----
    if csum:
        state = torch.zeros(6, dtype=torch.int32)
        last = world - 1
        # every non-final piece must be a stripe multiple: a rank whose shard is not (ragged
        # frames) passes its tail on with the state instead -- block sizes are multiples of
        # 16, so this only happens for frames a Writer was flushed into
        if any(s % 16 for s in sizes[:-1]):
            dg = _relay_ragged(wire, ops, flat, sizes, rank)
        else:
            staged = ops.stage_ahead(flat, n_local) if rank > 0 and hasattr(ops, "stage_ahead") else None

            def step(st):
                if staged is not None:
                    return ops.xxh32_update_staged(staged, flat, n_local, st, False, rank == last)
                return ops.xxh32_update(flat, n_local, st, rank == 0, rank == last)
            dg = wire.relay_state(state, 0, last, step)
        if dg != want_cs:
            raise ShardError(_lib.ERR_FRAME_CHECKSUM)
    clock.mark("checksum_s")

    if gather:
        if rank == root:
            whole = torch.zeros(total + 16, dtype=torch.uint8, device=dev)[:total]
            at, recvs = 0, []
            for g in range(world):
                if g == root:
                    whole[at:at + sizes[g]] = flat
                elif sizes[g]:
                    recvs.append((whole[at:at + sizes[g]], g))
                at += sizes[g]
            wire.exchange([], recvs)
            return whole, 0, total
        if n_local:
            wire.exchange([(flat, root)], [])
    return flat, start, total


def _relay_ragged(wire, ops, flat, sizes, rank):
    """Content checksum when shard lengths are not stripe multiples: shards are gathered on
    rank 0 and hashed there in one piece (rare: a frame with short blocks mid-stream)."""
    total = sum(sizes)
    if rank == 0:
        whole = torch.zeros(total + 16, dtype=torch.uint8, device=flat.device)[:total]
        at, recvs = 0, []
        for g in range(wire.world):
            if g == 0:
                whole[:sizes[0]] = flat
            elif sizes[g]:
                recvs.append((whole[at:at + sizes[g]], g))
            at += sizes[g]
        wire.exchange([], recvs)
        state = torch.zeros(6, dtype=torch.int32)
        dg = ops.xxh32_update(whole, total, state, True, True)
    else:
        if sizes[rank]:
            wire.exchange([(flat, 0)], [])
        dg = 0
    t = wire.small(torch.tensor([dg], dtype=torch.int64),
                   torch.device("cuda", torch.cuda.current_device()) if not wire.staged else None)
    dist.broadcast(t, src=0, group=wire.group)
    return int(t.item()) & 0xFFFFFFFF
