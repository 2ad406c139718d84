"""BASELINE.json's configurations at their FULL per-GPU sizes, through the C ABI,
checked by size-independent properties (the oracle cannot run 8-16 GiB in test time):

  * compress -> frame body -> decode round trip equals the source, bit for bit;
  * checksum of checksums: xxh32 of every decoded block equals xxh32 of its source block
    and the checksum word stored in the frame;
  * a seeded sample of blocks is byte-identical to the oracle's output;
  * cfg5: the content checksum over the whole stream equals canonical XXH32 computed on
    the host (python xxhash, streamed) -- the length is chosen off the T6 degeneracy.

Configs: cfg2 text 8 GiB / 64 KiB blocks (decode), cfg3 mixed
8 GiB / 256 KiB blocks Fast, cfg4 binary 4 GiB / 1 MiB blocks HC depth 131072, cfg5's
per-GPU share 16 GiB mixed / 4 MiB blocks with block + content checksums.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import torch
    return torch


def _need(T, gib):
    free, _ = T.cuda.mem_get_info()
    if free < int(gib * (1 << 30)):
        pytest.skip("needs %.0f GiB of free HBM" % gib)


def _full_size(T, oracle, kind, n, bs, level, samples, content_checksum=False):
    from lz4_amd import corpus, device as dev
    dv = T.device("cuda", 0)
    src = corpus.generate_device(kind, n, dv)
    nb = (n + bs - 1) // bs
    comp, res = dev.compress_blocks(src, bs, level)
    xxh = dev.xxh32_blocks(src, bs, nb, n=n)
    off, sw = dev.frame_layout(res, n, bs, nb, xxh, 7)
    end = int(off[nb].item())
    frame = T.empty(end + 16, dtype=T.uint8, device=dv)
    dev.frame_scatter(src, bs, comp, res, off, nb, frame, 0, xxh, n=n)

    # sampled blocks against the oracle: payload bytes, size word, checksum word
    res_h = res.cpu().numpy()
    sw_h = sw.cpu().numpy().astype(np.uint32)
    xxh_h = xxh.cpu().numpy().astype(np.uint32)
    off_h = off.cpu().numpy()
    rng = np.random.default_rng(2024)
    pick = sorted(set(rng.integers(0, nb, samples).tolist()) | {0, nb - 1})
    for b in pick:
        blk = src[b * bs:min(n, (b + 1) * bs)].cpu().numpy()
        pl, osw, oxx = oracle.compress_blocks(blk, bs, level, 1, True)
        assert sw_h[b] == osw[0], (b, hex(sw_h[b]), hex(osw[0]))
        assert xxh_h[b] == oxx[0], b
        size = int(osw[0] & 0x7FFFFFFF)
        at = int(off_h[b])
        entry = frame[at:at + 4 + size + 4].cpu().numpy()
        assert int(entry[:4].view(np.uint32)[0]) == osw[0], b
        want = blk.tobytes() if osw[0] >> 31 else pl[0, :size].tobytes()
        assert entry[4:4 + size].tobytes() == want, b
        if oxx[0] != 0:
            assert int(entry[4 + size:8 + size].copy().view(np.uint32)[0]) == oxx[0], b
    ratio = float(np.where(res_h > 0, res_h, bs).sum()) / n
    del comp
    T.cuda.empty_cache()

    # decode from the assembled body
    out, out_len = dev.uncompress_blocks(frame, (off[:nb] + 4).contiguous(), sw, bs)
    want_len = T.full((nb,), bs, dtype=T.int32, device=dv)
    want_len[nb - 1] = n - (nb - 1) * bs
    assert T.equal(out_len, want_len)
    assert T.equal(out.view(-1)[:n], src)
    xxh2 = dev.xxh32_blocks(out.view(-1), bs, nb, n=n)
    assert T.equal(xxh2, xxh)

    if content_checksum:
        import xxhash
        got = int(dev.xxh32_stream(src, n).item()) & 0xFFFFFFFF
        h = xxhash.xxh32()
        step = 1 << 30
        for a in range(0, n, step):
            h.update(src[a:min(n, a + step)].cpu().numpy().tobytes())
        assert got == h.intdigest()
    return ratio


def test_cfg2_text_8gib_64k_blocks(T, oracle):
    _need(T, 40)
    r = _full_size(T, oracle, "text", 8 << 30, 1 << 16, 0, 24)
    assert 0.6 < r < 0.7


def test_cfg3_mixed_8gib_256k_blocks(T, oracle):
    _need(T, 40)
    r = _full_size(T, oracle, "mixed", 8 << 30, 1 << 18, 0, 24)
    assert 0.70 < r < 0.75


def test_cfg4_binary_4gib_1m_blocks_hc_level9(T, oracle):
    _need(T, 24)
    _full_size(T, oracle, "binary", 4 << 30, 1 << 20, 131072, 6)


def test_cfg5_share_16gib_4m_blocks_checksums(T, oracle):
    """cfg5's per-GPU share (128 GiB over 8 GPUs), the non-degenerate length of SURVEY 8d:
    16 GiB - 4 MiB + 12,345 bytes, ragged last block."""
    _need(T, 80)
    _full_size(T, oracle, "mixed", (16 << 30) - (4 << 20) + 12345, 1 << 22, 0, 6,
               content_checksum=True)
