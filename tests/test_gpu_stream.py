"""Round-2 boundary additions through the C ABI against the oracle: the streaming
batch handle (Writer/Reader with ConcurrencyOption(n): writer.go:123-145,
lz4stream/block.go:21-58,95-185), dictionaries and source bounds on the batched decoder
(lz4.go:47-49, decode_other.go:109-121), the zero-checksum elision (T8,
lz4stream/block.go:270-272), the reader's block-size guard on standard frames (T11,
:301-304), legacy frames followed by padding, and error codes that never fold a failed
call into the codec's -1.
"""
import ctypes as C
import struct

import numpy as np
import pytest

import kats
from conftest import load

pytestmark = pytest.mark.gpu

M32 = 0xFFFFFFFF
P1, P2 = 2654435761, 2246822519


def _rol(x, k):
    return ((x << k) | (x >> (32 - k))) & M32


def _ror(x, k):
    return ((x >> k) | (x << (32 - k))) & M32


def zero_xxh32_block(data):
    """`data` (a multiple of 16, >= 32 bytes) with bytes [n-16, n-12) replaced so that
    XXH32(seed 0) of the result is 0.  With no tail bytes the digest is the avalanche of
    n + rol1(v1) + rol7(v2) + rol12(v3) + rol18(v4); the avalanche maps 0 to 0 and every
    step of the first accumulator is invertible (the primes are odd), so the word solves
    in closed form (xxh32zero.go:142-184)."""
    n = len(data)
    assert n % 16 == 0 and n >= 32
    v = [(P1 + P2) & M32, P2, 0, (-P1) & M32]
    for p in range(0, n - 16, 16):
        w = struct.unpack_from("<4I", data, p)
        v = [(_rol((v[i] + w[i] * P2) & M32, 13) * P1) & M32 for i in range(4)]
    w = struct.unpack_from("<4I", data, n - 16)
    last = [(_rol((v[i] + w[i] * P2) & M32, 13) * P1) & M32 for i in range(1, 4)]
    target = (-(n + _rol(last[0], 7) + _rol(last[1], 12) + _rol(last[2], 18))) & M32
    v1 = _ror(target, 1)                                   # rol1(v1') = target
    t = _ror((v1 * pow(P1, -1, 1 << 32)) & M32, 13)        # v1 + x*P2
    x = ((t - v[0]) * pow(P2, -1, 1 << 32)) & M32
    return data[:n - 16] + struct.pack("<I", x) + data[n - 12:]


def _np(b):
    return np.frombuffer(bytes(b), dtype=np.uint8) if not isinstance(b, np.ndarray) else b


def _ptr(a):
    return a.ctypes.data if a.size else None


def frame_compress(data, bs, flags, level=0):
    from lz4_amd import _lib
    lib = _lib.load()
    s = _np(data)
    cap = lib.lz4hip_frame_bound(s.size, bs, flags)
    d = np.zeros(cap, dtype=np.uint8)
    k = lib.lz4hip_frame_compress(_ptr(s), s.size, bs, flags, level, 0, d.ctypes.data, cap)
    assert k >= 0, (k, lib.lz4hip_last_error())
    return d[:k].tobytes()


# oracle frame-reader codes (oracle/lz4_oracle.h) -> LZ4HIP_ERR_*
ORACLE_TO_HIP = {-1: -9, -2: -4, -3: -5, -4: -6, -5: -7, -6: -8, -7: -4}


def same_reader_verdict(oracle, frame, cap):
    """The HIP reader and the oracle's reader agree on a frame that does not read back."""
    k, _ = frame_uncompress(frame, cap)
    with pytest.raises(ValueError) as e:
        oracle.frame_uncompress(frame, cap)
    assert k == ORACLE_TO_HIP[e.value.args[0]], (k, e.value.args[0])
    return k


def frame_uncompress(frame, cap):
    from lz4_amd import _lib
    lib = _lib.load()
    s = _np(frame)
    d = np.zeros(max(cap, 1), dtype=np.uint8)
    k = lib.lz4hip_frame_uncompress(_ptr(s), s.size, d.ctypes.data, cap)
    return k, d[:max(k, 0)].tobytes()


# ---------------------------------------------------------------- T8 / T11

def test_t8_four_byte_block_with_zero_checksum(oracle):
    """27 11 4b 23 has XXH32 0: with block checksums on, no checksum word follows it."""
    src = bytes([0x27, 0x11, 0x4B, 0x23])
    assert oracle.ChecksumZero(src) == 0
    for bs in (1 << 16, 1 << 22):
        f = frame_compress(src, bs, 1 | 2)
        assert f == oracle.frame_compress(src, bs, 1 | 2)
        # header 7 | size word | 4 raw bytes | end mark | content checksum: no block checksum
        assert len(f) == 7 + 4 + 4 + 4 + 4 and f[7:11] == struct.pack("<I", 4 | 0x80000000)
        # The reference's reader always expects the word (lz4stream/block.go:309-315): it takes
        # the end mark for the checksum and the content checksum for a size word, i.e. the
        # reference cannot read this frame of its own back.  Same verdict here.
        same_reader_verdict(oracle, f, 16)


@pytest.mark.parametrize("level", [0, 512])
def test_t8_zero_checksum_block_inside_a_frame(oracle, level):
    """A 5-block source whose blocks 1 and 3 hash to 0: the frame has their payloads
    without checksum words (layout + scatter take the elision branch), equals the
    oracle's bytes and reads back; the same through frame_layout / frame_scatter."""
    import torch
    from lz4_amd import device as dev
    bs = 1 << 16
    text = load("pg1661.txt")
    blocks = [text[i * bs:(i + 1) * bs] for i in range(4)] + [text[4 * bs:4 * bs + 4096]]
    blocks[1] = zero_xxh32_block(blocks[1])
    blocks[3] = zero_xxh32_block(blocks[3])
    data = b"".join(blocks)
    assert [oracle.ChecksumZero(b) == 0 for b in blocks] == [False, True, False, True, False]
    want = oracle.frame_compress(data, bs, 1 | 2, level)
    f = frame_compress(data, bs, 1 | 2, level)
    assert f == want
    # read back: block 1's missing word makes the reader take block 2's size word for it
    from lz4_amd import _lib
    assert same_reader_verdict(oracle, f, len(data)) == _lib.ERR_BLOCK_CHECKSUM
    if level == 0:  # the device entry points directly
        src = torch.zeros(len(data) + 16, dtype=torch.uint8, device="cuda")
        src[:len(data)] = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
        src = src[:len(data)]
        comp, res = dev.compress_blocks(src, bs, 0)
        xxh = dev.xxh32_blocks(src, bs, 5)
        assert [int(x) == 0 for x in xxh.cpu()] == [False, True, False, True, False]
        off, sw = dev.frame_layout(res, len(data), bs, 5, xxh, 7)
        end = int(off[5].item())
        frame = torch.zeros(end + 16, dtype=torch.uint8, device="cuda")
        dev.frame_scatter(src, bs, comp, res, off, 5, frame, 0, xxh)
        assert frame[7:end].cpu().numpy().tobytes() == want[7:end]
        assert end == len(want) - 8


def test_t11_standard_frame_block_larger_than_block_size(oracle):
    """lz4stream/block.go:301-304: a size word above the frame's block size is
    ErrOptionInvalidBlockSize on a STANDARD frame (raw or compressed flag alike)."""
    from lz4_amd import _lib
    hdr = struct.pack("<I", 0x184D2204) + kats.HEADERS[0][1]          # 64 KiB blocks
    for word in ((1 << 16) + 1, ((1 << 16) + 1) | 0x80000000, 0x7FFFFFFF):
        f = hdr + struct.pack("<I", word) + b"\x00" * ((1 << 16) + 64)
        assert frame_uncompress(f, 1 << 20)[0] == _lib.ERR_BLOCK_SIZE
        with pytest.raises(ValueError) as e:
            oracle.frame_uncompress(f, 1 << 20)
        assert e.value.args[0] == -4          # LZ4O_FE_BLOCK_SIZE
    ok = hdr + struct.pack("<I", (1 << 16) | 0x80000000) + b"z" * (1 << 16) + b"\0" * 4 \
        + struct.pack("<I", oracle.ChecksumZero(b"z" * (1 << 16)))
    assert frame_uncompress(ok, 1 << 20) == (1 << 16, b"z" * (1 << 16))


# ---------------------------------------------------------------- legacy + padding

def test_legacy_golden_followed_by_padding(oracle):
    """A kernel-style legacy frame (blocks, then the total size word) followed by zero
    padding: the walk ends at the size word (lz4stream/block.go:289-293); nothing behind
    it is decoded or given device memory.  1-3 stray bytes after the last block of a frame
    WITHOUT the size word are io.ErrUnexpectedEOF."""
    from lz4_amd import _lib
    raw = load("bzImage_lz4_isolated")
    golden = load(kats.LEGACY_FRAME)
    for pad in (4, 1 << 16, 1 << 20):
        f = golden + b"\x00" * pad
        k, out = frame_uncompress(f, len(raw) + 64)
        assert k == len(raw) and out == raw, pad
        assert oracle.frame_uncompress(f, len(raw) + 64) == raw
    # many empty blocks in a row between two blocks cost no slots; a zero word before any
    # output IS the end word (x == cum == 0, lz4stream/block.go:289-293)
    name, frame, want = kats.legacy_frames(lambda d: oracle.CompressBlock(d, len(d) + 64)[1])[0]
    cut = 8 + struct.unpack_from("<I", frame, 4)[0]
    f = frame[:cut] + b"\x00" * 4000 + frame[cut:]
    assert frame_uncompress(f, 1 << 20) == (len(want), want)
    assert oracle.frame_uncompress(f, 1 << 20) == want
    f = frame[:4] + b"\x00" * 4000 + frame[4:]
    assert frame_uncompress(f, 1 << 20) == (0, b"")
    assert oracle.frame_uncompress(f, 1 << 20) == b""
    for stray in (1, 2, 3):
        g = frame + b"\x01" * stray
        assert frame_uncompress(g, 1 << 20)[0] == _lib.ERR_FRAME, stray
        with pytest.raises(ValueError):
            oracle.frame_uncompress(g, 1 << 20)


# ---------------------------------------------------------------- error codes

def test_failed_calls_are_not_folded_into_the_codec_verdict():
    from lz4_amd import _lib
    lib = _lib.load()
    dst = np.zeros(64, dtype=np.uint8)
    # a bad argument is LZ4HIP_ERR_BADARG, never -1 (= ErrInvalidSourceShortBuffer = "store raw")
    assert lib.lz4hip_compress_block(None, 10, dst.ctypes.data, 64) == _lib.ERR_BADARG
    assert lib.lz4hip_compress_block_hc(None, 10, dst.ctypes.data, 64, 16) == _lib.ERR_BADARG
    assert lib.lz4hip_uncompress_block(None, 10, dst.ctypes.data, 64, None, 0) == _lib.ERR_BADARG
    out = C.c_uint32(77)
    assert lib.lz4hip_xxh32_checked(None, 10, C.addressof(out)) == _lib.ERR_BADARG
    src = np.frombuffer(b"abcdefgh" * 8, dtype=np.uint8)
    assert lib.lz4hip_xxh32_checked(src.ctypes.data, src.size, C.addressof(out)) == 0
    assert out.value == lib.lz4hip_xxh32(src.ctypes.data, src.size)
    # the codec's own verdicts are still -1 / 0
    assert lib.lz4hip_compress_block(src.ctypes.data, src.size, dst.ctypes.data, 3) == -1
    assert lib.lz4hip_uncompress_block(src.ctypes.data, src.size, dst.ctypes.data, 64, None, 0) == -1


def test_init_pools_device_memory():
    from lz4_amd import _lib
    lib = _lib.load()
    assert lib.lz4hip_init(-1) == 0
    data = load("pg1661.txt")
    a = frame_compress(data, 1 << 16, 2)
    b = frame_compress(data, 1 << 16, 2)          # second call: buffers from the pool
    assert a == b and frame_uncompress(a, len(data)) == (len(data), data)
    assert lib.lz4hip_shutdown() == 0
    assert frame_compress(data, 1 << 16, 2) == a
    assert lib.lz4hip_init(99) == _lib.ERR_ARG


# ---------------------------------------------------------------- the stream handle

class Stream:
    def __init__(self, bs, flags, level, max_blocks):
        from lz4_amd import _lib
        self.lib = _lib.load()
        self.h = self.lib.lz4hip_stream_create(bs, flags, level, max_blocks)
        assert self.h, self.lib.lz4hip_last_error()
        self.bs, self.max_blocks = bs, max_blocks

    def close(self):
        self.lib.lz4hip_stream_destroy(self.h)

    def compress(self, data):
        s = _np(data)
        nb = max(1, -(-s.size // self.bs))
        cap = s.size + 8 * nb + 64
        d = np.zeros(cap, dtype=np.uint8)
        lens = np.zeros(nb, dtype=np.uint32)
        k = self.lib.lz4hip_stream_compress_blocks(self.h, _ptr(s), s.size, d.ctypes.data, cap,
                                                   lens.ctypes.data)
        assert k >= 0, (k, self.lib.lz4hip_last_error())
        return d[:k].tobytes(), lens

    def uncompress(self, entries, cap):
        s = _np(entries)
        d = np.zeros(max(cap, 1), dtype=np.uint8)
        used, end, nb = C.c_size_t(), C.c_int(), C.c_size_t()
        lens = np.zeros(self.max_blocks, dtype=np.uint32)
        k = self.lib.lz4hip_stream_uncompress_blocks(self.h, _ptr(s), s.size, d.ctypes.data, cap,
                                                     C.addressof(used), C.addressof(end),
                                                     lens.ctypes.data, C.addressof(nb))
        return k, d[:max(k, 0)].tobytes(), used.value, end.value, lens[:nb.value]

    def checksum(self):
        return self.lib.lz4hip_stream_checksum(self.h)


def _writer(data, bs, flags, level, batch, readfrom=False):
    """What Writer.Write / ReadFrom + Close do with a batch hook: header, one call per
    `batch` blocks, end mark, content checksum (writer.go:74-145,173-184,199-242)."""
    from lz4_amd import _lib
    lib = _lib.load()
    hdr = np.zeros(16, dtype=np.uint8)
    out = bytearray(hdr[:lib.lz4hip_frame_header(hdr.ctypes.data, bs, flags, 0)].tobytes())
    st = Stream(bs, flags, level, batch)
    lens = []
    for a in range(0, len(data), batch * bs):
        e, l = st.compress(data[a:a + batch * bs])
        out += e
        lens += list(l)
    if readfrom and len(data) % bs == 0:       # ReadFrom writes data[:0] (writer.go:220-233)
        e, l = st.compress(b"")
        out += e
        lens += list(l)
    trl = np.zeros(8, dtype=np.uint8)
    out += trl[:lib.lz4hip_frame_trailer(trl.ctypes.data, flags, st.checksum())].tobytes()
    st.close()
    return bytes(out), lens


@pytest.mark.parametrize("bs,batch", [(1 << 16, 1), (1 << 16, 4), (1 << 16, 64), (1 << 18, 3)])
@pytest.mark.parametrize("flags", [0, 1, 2, 3])
def test_stream_writer_bytes_equal_oracle(oracle, bs, batch, flags):
    data = load("pg1661.txt") + load("random.data") + load("bzImage_lz4_isolated")[:300000]
    frame, lens = _writer(data, bs, flags, 0, batch)
    assert frame == oracle.frame_compress(data, bs, flags)
    nb = -(-len(data) // bs)
    assert len(lens) == nb and frame_uncompress(frame, len(data)) == (len(data), data)


def test_stream_writer_hc_readfrom_and_more_blocks_than_cus(oracle):
    bs = 1 << 16
    data = load("pg1661.txt")[:4 * bs]                      # a block multiple: T9
    frame, lens = _writer(data, bs, 3, 512, 2, readfrom=True)
    assert frame == oracle.frame_compress(data, bs, 3 | 8, 512)
    assert lens[-1] == 0
    from lz4_amd import corpus                              # 300 blocks per call: global-table kernel
    big = corpus.generate("mixed", 7, 300 * bs + 777).tobytes()
    frame, _ = _writer(big, bs, 3, 0, 300)
    assert frame == oracle.frame_compress(big, bs, 3)
    # Reset re-arms the checksum (Writer.Reset -> Frame.Reset)
    st = Stream(bs, 2, 0, 4)
    st.compress(data[:bs])
    st.lib.lz4hip_stream_reset(st.h)
    assert st.checksum() == 0x02CC5D05
    st.close()


@pytest.mark.parametrize("flags", [0, 1, 2, 3])
def test_stream_reader_arbitrary_feeding(oracle, flags):
    """Reader with a batch hook: the frame's block area arrives in arbitrary pieces; every
    call decodes the complete entries it holds and says how much it used."""
    from lz4_amd import _lib
    bs = 1 << 16
    data = load("pg1661.txt") + load("random.data") + load("repeat.txt")
    frame = oracle.frame_compress(data, bs, flags)
    body = frame[7:]
    rng = np.random.default_rng(flags)
    st = Stream(bs, flags, 0, 5)
    out, lens, at, buf, ended, calls = bytearray(), [], 0, b"", False, 0
    while not ended:
        calls += 1
        assert calls < 1000
        take = int(rng.integers(1, 3 * bs))
        buf += body[at:at + take]
        at += take
        k, o, used, end, l = st.uncompress(buf, 5 * bs)
        assert k >= 0, (k, _lib.last_error())
        out += o
        lens += list(l)
        buf = buf[used:]
        ended = bool(end)
    assert bytes(out) == data and sum(lens) == len(data)
    if flags & 2:   # what is left is the content checksum word; the handle has the same
        assert struct.unpack("<I", (buf + body[at:])[:4])[0] == st.checksum()
    st.close()


def test_stream_reader_errors_keep_stream_order(oracle):
    from lz4_amd import _lib
    bs = 1 << 16
    data = load("pg1661.txt")[:5 * bs]
    frame = bytearray(oracle.frame_compress(data, bs, 1))
    body = bytes(frame[7:])
    # entry offsets
    offs, si = [], 0
    while True:
        w = struct.unpack_from("<I", body, si)[0]
        if w == 0:
            break
        offs.append(si)
        si += 4 + (w & 0x7FFFFFFF) + 4
    st = Stream(bs, 1, 0, 8)
    bad = bytearray(body)
    bad[offs[3] + 4 + 10] ^= 0x55            # block 3: payload damaged
    bad[offs[1] + 4 + (struct.unpack_from("<I", body, offs[1])[0] & 0x7FFFFFFF)] ^= 1   # block 1: checksum word
    k = st.uncompress(bytes(bad), 8 * bs)[0]
    assert k == _lib.ERR_BLOCK_CHECKSUM       # block 1 comes first in the stream
    bad = bytearray(body)
    bad[offs[0] + 4:offs[0] + 8] = b"\xff\xff\xff\xff"
    assert st.uncompress(bytes(bad), 8 * bs)[0] in (_lib.ERR_BLOCK, _lib.ERR_BLOCK_CHECKSUM)
    big = struct.pack("<I", bs + 1) + body     # T11 through the handle
    assert st.uncompress(big, 8 * bs)[0] == _lib.ERR_BLOCK_SIZE
    assert st.uncompress(body, 100)[0] == _lib.ERR_SHORT
    st.close()


# ---------------------------------------------------------------- batched decode: dict, src_len

def encode_with_history(history, data, min_off=1):
    """A small greedy LZ4 encoder (test data only): sequences may point into `history`,
    which precedes `data`.  Returns one LZ4 block that decodes to `data` given the
    dictionary `history`."""
    whole = history + data
    base = len(history)
    table = {}
    for i in range(max(0, base - 65535), base - 3):
        table[whole[i:i + 4]] = i
    out, anchor, i, n = bytearray(), base, base, len(whole)
    while i + 4 <= n - 5:
        key = whole[i:i + 4]
        c = table.get(key)
        table[key] = i
        if c is not None and min_off <= i - c <= 65535:
            ml = 4
            while i + ml < n - 5 and whole[c + ml] == whole[i + ml]:
                ml += 1
            out += kats._seq(whole[anchor:i], i - c, ml)
            i += ml
            anchor = i
        else:
            i += 1
    out += kats._seq(whole[anchor:], 0, 0)
    return bytes(out)


def test_uncompress_blocks_with_dictionary(oracle):
    import torch
    from lz4_amd import device as dev
    rng = np.random.default_rng(11)
    text = load("pg1661.txt")
    dic = text[1000:1000 + 70000]                 # longer than 64 KiB: only the tail is reachable
    blocks, wants = [], []
    for i in range(40):
        a = int(rng.integers(0, 60000))
        body = text[1000 + a:1000 + a + int(rng.integers(20, 3000))]      # overlaps the dictionary's text
        noise = bytes(rng.integers(97, 101, int(rng.integers(0, 40)), dtype=np.uint8))
        runs = bytes([dic[-1]]) * int(rng.integers(0, 300))                # period-1 run fed by the dict's last byte
        data = (runs + body + noise) if i % 2 else (body + runs + noise)
        blocks.append(encode_with_history(dic[-65535:], data))
        wants.append(data)
    blocks.append(encode_with_history(b"", b"no dictionary needed " * 30))
    wants.append(b"no dictionary needed " * 30)
    # matches that start inside the dictionary and run on into the block's own output,
    # overlapping themselves (decode_other.go:109-133): offsets 1, 3, 10, 1000, all of it
    for off, ml in ((1, 300), (3, 50), (3, 4), (10, 200), (1000, 3000), (65535, 70), (65535, 4000)):
        tail = dic[-off:]
        want = bytearray()
        for k in range(ml):
            want.append((tail + bytes(want))[k])
        blocks.append(kats._seq(b"", off, ml) + kats._seq(b"tail!", 0, 0))
        wants.append(bytes(want) + b"tail!")
        if off + 2 > 65535:
            continue
        blocks.append(kats._seq(b"ab", off + 2, ml) + kats._seq(b"tail!", 0, 0))   # di = 2: two literals first
        want2 = bytearray(b"ab")
        src2 = dic[-off:] + b"ab"
        for k in range(ml):
            want2.append((src2 + bytes(want2[2:]))[k])
        wants.append(bytes(want2) + b"tail!")
    blocks.append(b"\x10a\x00\x00\x50abcde")        # offset 0: error (decode_other.go:88-90)
    wants.append(None)
    for b, w in zip(blocks, wants):               # the oracle agrees on every block
        if w is None:
            with pytest.raises(oracle.ErrInvalidSourceShortBuffer):
                oracle.UncompressBlock(b, 4096, dic)
        else:
            assert oracle.UncompressBlock(b, 4096, dic) == w
    blob = b"".join(blocks)
    offs = np.cumsum([0] + [len(b) for b in blocks[:-1]]).astype(np.int64)
    sw = np.array([len(b) for b in blocks], dtype=np.int32)
    d_blob = torch.from_numpy(np.frombuffer(blob, dtype=np.uint8).copy()).cuda()
    d_dic = torch.from_numpy(np.frombuffer(dic, dtype=np.uint8).copy()).cuda()
    out, ol = dev.uncompress_blocks(d_blob, torch.from_numpy(offs).cuda(), torch.from_numpy(sw).cuda(),
                                    4096, dict=d_dic)
    ol = ol.cpu().numpy()
    out = out.cpu().numpy()
    for b, w in enumerate(wants):
        if w is None:
            assert ol[b] == -1
        else:
            assert ol[b] == len(w) and out[b, :len(w)].tobytes() == w, b
    # the reference's own dictionary cases (decode_test.go:262-284), one batch per dictionary
    for src, d, want in kats.WITH_DICT:
        o, l = dev.uncompress_blocks(torch.from_numpy(np.frombuffer(src, dtype=np.uint8).copy()).cuda(),
                                     torch.zeros(1, dtype=torch.int64, device="cuda"),
                                     torch.tensor([len(src)], dtype=torch.int32, device="cuda"), 100,
                                     dict=torch.from_numpy(np.frombuffer(d, dtype=np.uint8).copy()).cuda())
        if want is None:
            assert int(l[0]) == -1
        else:
            assert o[0, :int(l[0])].cpu().numpy().tobytes() == want


def test_uncompress_blocks_rejects_spans_outside_src(oracle):
    import torch
    from lz4_amd import device as dev
    good = oracle.CompressBlock(b"inside the buffer " * 50)[1]
    blob = np.frombuffer(good * 3, dtype=np.uint8).copy()
    n = len(good)
    offs = np.array([0, n, 2 * n, 2 * n + 1, 3 * n, 1 << 40], dtype=np.int64)
    sw = np.array([n, n, n, n, 1, n], dtype=np.uint32).astype(np.int32)
    sw_raw = sw.view(np.uint32).copy()
    sw_raw[3] |= 0x80000000                       # a raw block that leaves the buffer
    sw_raw = sw_raw.view(np.int32)
    d_blob = torch.from_numpy(blob).cuda()
    for words in (sw, sw_raw):
        out, ol = dev.uncompress_blocks(d_blob, torch.from_numpy(offs).cuda(),
                                        torch.from_numpy(words).cuda(), 2048)
        ol = ol.cpu().numpy()
        assert list(ol[:3]) == [900, 900, 900] and list(ol[3:]) == [-1, -1, -1]
        assert out[0, :900].cpu().numpy().tobytes() == b"inside the buffer " * 50


# ---------------------------------------------------------------- staged host copies

@pytest.mark.parametrize("mib", [5, 33, 64, 70])
def test_host_frames_through_staged_copies(oracle, mib):
    """Host buffers above 4 MiB travel through two 32 MiB pinned buffers (abi_common.h Stager):
    one piece, one and a bit, exactly two, three -- frame bytes equal the oracle's Writer
    (content + block checksums), and the frame reads back through the same staging."""
    from lz4_amd import _lib, corpus
    n = mib << 20
    data = corpus.generate("mixed", 11 + mib, n)
    flags = _lib.F_CONTENT_CHECKSUM | _lib.F_BLOCK_CHECKSUM
    frame = frame_compress(data, 1 << 18, flags)
    assert frame == oracle.frame_compress(data, 1 << 18, flags)
    k, back = frame_uncompress(frame, n)
    assert k == n and back == data.tobytes()
    # a Stream handle takes the same path: one batch of the whole input
    s = Stream(1 << 18, flags, 0, n >> 18)
    try:
        entries, _ = s.compress(data)
        assert entries == frame[7:7 + len(entries)] and len(entries) == len(frame) - 7 - 8
        k, back, used, end, _ = s.uncompress(entries, n)
        assert k == n and back == data.tobytes() and used == len(entries) and end == 0
    finally:
        s.close()


@pytest.mark.parametrize("bs,mib", [(1 << 18, 800), (1 << 16, 769)])
def test_host_frame_compress_pipelined_sub_batches(oracle, bs, mib):
    """From 768 MiB up (256 KiB blocks and smaller) lz4hip_frame_compress compresses sub-batches
    of 256 MiB on side streams while the rest of the source is still being uploaded
    (host_api.hip compress_batch).  The frame equals the single-launch device path block for
    block -- entry bytes from lz4hip_compress_fast_blocks + frame_layout/scatter -- a sample
    of blocks equals the oracle's, and the frame reads back to the source with both
    checksums verified."""
    import torch as T
    from lz4_amd import _lib, corpus, device as dev
    n = (mib << 20) + 12345
    data = corpus.generate("mixed", 77, n)
    flags = _lib.F_CONTENT_CHECKSUM | _lib.F_BLOCK_CHECKSUM
    frame = frame_compress(data, bs, flags)
    nb = -(-n // bs)
    # the same blocks in one launch on the device path
    src = T.from_numpy(data).to("cuda:0")
    comp, res = dev.compress_blocks(src, bs, 0)
    xxh = dev.xxh32_blocks(src, bs, nb, n=n)
    off, sw = dev.frame_layout(res, n, bs, nb, xxh, 7)
    end = int(off[nb].item())
    body = T.empty(end + 16, dtype=T.uint8, device=src.device)
    dev.frame_scatter(src, bs, comp, res, off, nb, body, 0, xxh, n=n)
    assert len(frame) == end + 8
    assert frame[7:end] == body[7:end].cpu().numpy().tobytes()
    off_h = off.cpu().numpy()
    for b in (0, 1023, 1024, 2047, 2048, nb // 2, nb - 2, nb - 1):
        blk = data[b * bs:min(n, (b + 1) * bs)]
        _, osw, oxx = oracle.compress_blocks(blk, bs, 0, 1, True)
        at = int(off_h[b])
        assert struct.unpack_from("<I", frame, at)[0] == osw[0], b
        size = int(osw[0] & 0x7FFFFFFF)
        assert struct.unpack_from("<I", frame, at + 4 + size)[0] == oxx[0], b
    del src, comp, body
    k, back = frame_uncompress(frame, n)
    assert k == n and back == data.tobytes()


def test_host_frames_from_concurrent_threads(oracle):
    """cgo runs every goroutine's call on its own OS thread: six threads call
    lz4hip_frame_compress / _uncompress at once on 40 MiB each (staged copies: every call leases
    its own pinned buffers; per-thread streams).  Every frame equals the oracle's and reads back."""
    import threading
    from lz4_amd import _lib, corpus
    n = 40 << 20
    flags = _lib.F_CONTENT_CHECKSUM | _lib.F_BLOCK_CHECKSUM
    inputs = [corpus.generate("mixed", 1000 + 7 * t, n + 1000 * t) for t in range(6)]
    frames, backs, errs = [None] * 6, [None] * 6, []

    def work(t):
        try:
            for _ in range(2):
                frames[t] = frame_compress(inputs[t], 1 << 18, flags)
                k, back = frame_uncompress(frames[t], inputs[t].size)
                backs[t] = (k, back)
        except BaseException as e:  # surfaced in the main thread
            errs.append(e)

    ts = [threading.Thread(target=work, args=(t,)) for t in range(6)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for t in range(6):
        assert backs[t][0] == inputs[t].size and backs[t][1] == inputs[t].tobytes()
    assert frames[0] == oracle.frame_compress(inputs[0], 1 << 18, flags)
    assert frames[5] == oracle.frame_compress(inputs[5], 1 << 18, flags)


def test_legacy_upload_beside_staged_downloads(oracle):
    """A legacy frame larger than one 32 MiB staging piece is uploaded and decoded on one thread
    while two others run frame_compress / frame_uncompress on 40 MiB (staged downloads): a
    Stager goes back to the free list only when its own copies have left its pinned buffers
    (abi_common.h Lease::~Lease), so a download of the next lessee can not overwrite bytes an
    upload is still reading.  Legacy frames carry no checksum: the decoded bytes are compared."""
    import threading
    from lz4_amd import _lib, corpus
    n_leg = 72 << 20
    # mostly incompressible, so that the legacy FRAME itself is well above one piece
    raw = np.concatenate([corpus.generate("random", 5, 48 << 20), corpus.generate("text", 9, n_leg - (48 << 20))])
    legacy = frame_compress(raw, 0, _lib.F_LEGACY)
    assert len(legacy) > (40 << 20)
    flags = _lib.F_CONTENT_CHECKSUM | _lib.F_BLOCK_CHECKSUM
    others = [corpus.generate("mixed", 300 + t, (40 << 20) + 4096 * t) for t in range(2)]
    errs, done = [], []

    def leg():
        try:
            for _ in range(3):
                k, back = frame_uncompress(legacy, n_leg)
                assert k == n_leg and back == raw.tobytes()
            done.append("legacy")
        except BaseException as e:
            errs.append(e)

    def std(t):
        try:
            for _ in range(3):
                f = frame_compress(others[t], 1 << 18, flags)
                k, back = frame_uncompress(f, others[t].size)
                assert k == others[t].size and back == others[t].tobytes()
            done.append(t)
        except BaseException as e:
            errs.append(e)

    ts = [threading.Thread(target=leg)] + [threading.Thread(target=std, args=(t,)) for t in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    assert len(done) == 3


def test_stream_update_host_from_short_lived_threads(oracle):
    """lz4hip_xxh32_stream_update_host from a dozen threads that each live for one call: the
    pinned buffers are a leased Stager's (freed by lz4hip_shutdown), not the calling thread's,
    so the process's resident memory does not grow by 64 MiB per thread."""
    import threading
    import psutil
    import torch as T
    from lz4_amd import corpus, device as dev
    n = (40 << 20) + 16 * 77
    data = corpus.generate("mixed", 4242, n)
    want = oracle.ChecksumZero(data)
    src = T.from_numpy(data).cuda()
    got, errs = [], []

    def work():
        try:
            state = T.zeros(6, dtype=T.int32)
            half = (n // 2) & ~15
            dev.xxh32_stream_update_host(src, half, state, True, False)
            got.append(dev.xxh32_stream_update_host(src[half:], n - half, state, False, True))
        except BaseException as e:
            errs.append(e)

    def one():
        t = threading.Thread(target=work)
        t.start()
        t.join()

    one()
    one()
    rss0 = psutil.Process().memory_info().rss
    for _ in range(12):
        one()
    grown = psutil.Process().memory_info().rss - rss0
    assert not errs, errs
    assert got and all(g == want for g in got)
    assert grown < (200 << 20), grown       # 12 x 64 MiB = 768 MiB before the fix
