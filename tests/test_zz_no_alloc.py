"""lz4hip_compress_fast_blocks_noalloc (the entry point that allocates nothing: its hash
tables and block counter live in a file-scope device array of the library) against
lz4hip_compress_fast_blocks with a caller's workspace, through the C ABI: same result
words and the same compressed bytes, on calls in a row that share the scratch -- the
second and third with smaller inputs than the first.  A seeded sample of blocks of every
call is also compared with the oracle directly (result word and bytes), and one test captures
the call in a HIP graph and replays it."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _both(torch, lib, src, bs, dst_mode):
    from lz4_amd import _lib, device as dev
    n = src.numel()
    nb = (n + bs - 1) // bs
    stride = bs if dst_mode == _lib.DST_FRAME else (lib.lz4hip_compress_bound(bs) + 15) // 16 * 16
    st = torch.cuda.current_stream().cuda_stream
    # existing entry point (torch allocates its workspace)
    want_c, want_r = dev.compress_blocks(src, bs, 0, dst_mode)
    # the new one: inputs and outputs only
    got_c = torch.full((nb, stride), 0xEE, dtype=torch.uint8, device=src.device)
    got_r = torch.full((nb,), -77, dtype=torch.int32, device=src.device)
    torch.cuda.synchronize()
    before = torch.cuda.memory_allocated()
    rc = lib.lz4hip_compress_fast_blocks_noalloc(src.data_ptr(), n, bs, dst_mode, got_c.data_ptr(),
                                                 stride, got_r.data_ptr(), st)
    assert rc == 0, lib.lz4hip_last_error()
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() == before
    wr, gr = want_r.cpu().numpy(), got_r.cpu().numpy()
    assert (wr == gr).all()
    wc, gc = want_c.cpu().numpy(), got_c.cpu().numpy()
    for b in range(nb):
        if wr[b] > 0:
            assert wc[b, :wr[b]].tobytes() == gc[b, :wr[b]].tobytes(), b
    _against_oracle(src, bs, dst_mode, gr, gc)
    return int((wr > 0).sum())


def _against_oracle(src, bs, dst_mode, gr, gc, picks=12):
    """The no-alloc entry's own output against the oracle, not only against its sibling."""
    from lz4_amd import _lib
    from oracle import pyoracle as orc
    n = src.numel()
    nb = (n + bs - 1) // bs
    rng = np.random.default_rng(nb)
    for b in sorted(set(rng.integers(0, nb, picks).tolist()) | {0, nb - 1}):
        blk = src[b * bs:min(n, (b + 1) * bs)].cpu().numpy()
        dlen = len(blk) if dst_mode == _lib.DST_FRAME else orc.CompressBlockBound(len(blk))
        try:
            want_n, want = orc.CompressBlock(blk, dlen)
        except orc.ErrInvalidSourceShortBuffer:
            want_n, want = -1, b""
        assert int(gr[b]) == want_n, (b, int(gr[b]), want_n)
        if want_n > 0:
            assert gc[b, :want_n].tobytes() == want, b


def test_noalloc_matches_workspace_entry_point():
    import torch
    from lz4_amd import _lib, corpus
    lib = _lib.load()
    _lib.require_gpu()
    bs = 1 << 16

    def dev_src(kind, start, n):
        data = corpus.generate(kind, start, n)
        t = torch.zeros(n + 16, dtype=torch.uint8, device="cuda")
        t[:n] = torch.from_numpy(data).cuda()
        return t[:n]

    # 1. more blocks than scratch tables (700 > 512): persistent waves + block counter
    assert _both(torch, lib, dev_src("mixed", 0, 700 * bs + 4321), bs, _lib.DST_FRAME) > 300
    # 2. right after it, a smaller input on the same scratch (560 blocks: still more than tables)
    assert _both(torch, lib, dev_src("text", 12345, 560 * bs), bs, _lib.DST_FRAME) == 560
    # 3. bound-sized dst, 256 KiB blocks, again more blocks than tables
    assert _both(torch, lib, dev_src("binary", 999, 520 * (1 << 18) + 17), 1 << 18, _lib.DST_BOUND) > 500
    # 4. a handful of blocks (the team kernel, tables in LDS: no scratch at all), then many again
    assert _both(torch, lib, dev_src("text", 5, 3 * bs + 100), bs, _lib.DST_FRAME) >= 3
    assert _both(torch, lib, dev_src("mixed", 1 << 20, 600 * bs), bs, _lib.DST_FRAME) > 250


def test_noalloc_rejects_more_than_its_maximum():
    import torch
    from lz4_amd import _lib
    lib = _lib.load()
    _lib.require_gpu()
    buf = torch.zeros(1 << 16, dtype=torch.uint8, device="cuda")
    res = torch.full((4,), 5, dtype=torch.int32, device="cuda")
    rc = lib.lz4hip_compress_fast_blocks_noalloc(buf.data_ptr(), (4 << 30) + 1, 1 << 16, 0,
                                                 buf.data_ptr(), 1 << 16, res.data_ptr(), None)
    assert rc == _lib.ERR_ARG and b"LZ4HIP_NOALLOC_MAX_BYTES" in lib.lz4hip_last_error()
    torch.cuda.synchronize()
    assert (res.cpu().numpy() == 5).all()      # nothing was launched


def test_noalloc_under_graph_capture():
    """The call captured in a HIP graph (after lz4hip_init resolved the scratch address, as
    include/lz4hip.h asks) and replayed on two different inputs written into the same buffer."""
    import torch
    from lz4_amd import _lib, corpus
    lib = _lib.load()
    _lib.require_gpu()
    assert lib.lz4hip_init(torch.cuda.current_device()) == 0, lib.lz4hip_last_error()
    bs, nb = 1 << 16, 600
    n = nb * bs
    src = torch.zeros(n + 16, dtype=torch.uint8, device="cuda")[:n]
    got_c = torch.zeros((nb, bs), dtype=torch.uint8, device="cuda")
    got_r = torch.zeros(nb, dtype=torch.int32, device="cuda")
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            rc = lib.lz4hip_compress_fast_blocks_noalloc(src.data_ptr(), n, bs, _lib.DST_FRAME,
                                                         got_c.data_ptr(), bs, got_r.data_ptr(),
                                                         torch.cuda.current_stream().cuda_stream)
    assert rc == 0, lib.lz4hip_last_error()
    torch.cuda.current_stream().wait_stream(side)
    from lz4_amd import device as dev
    for kind, start in (("mixed", 7 << 20), ("text", 1 << 16)):
        src.copy_(torch.from_numpy(corpus.generate(kind, start, n)).cuda())
        got_r.fill_(-77)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        want_c, want_r = dev.compress_blocks(src, bs, 0, _lib.DST_FRAME)
        wr, gr = want_r.cpu().numpy(), got_r.cpu().numpy()
        assert (wr == gr).all()
        wc, gc = want_c.cpu().numpy(), got_c.cpu().numpy()
        for b in range(nb):
            if wr[b] > 0:
                assert wc[b, :wr[b]].tobytes() == gc[b, :wr[b]].tobytes(), b
        _against_oracle(src, bs, _lib.DST_FRAME, gr, gc, picks=6)
