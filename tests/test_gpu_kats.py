"""The reference's exact known answers, run through the HIP path via the C ABI
(host-pointer entry points, i.e. what the cgo shim calls)."""
import io
import json
import os

import numpy as np
import pytest

import kats
from conftest import GOLDEN, load

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,src,exp", kats.DECODE, ids=[k[0] for k in kats.DECODE])
def test_block_decode_kat(name, src, exp):
    import lz4_amd as lz
    dst = np.zeros(len(exp), dtype=np.uint8)
    assert lz.UncompressBlock(src, dst) == len(exp)
    assert dst.tobytes() == exp


@pytest.mark.parametrize("name,src,size", kats.INVALID, ids=[k[0] for k in kats.INVALID])
def test_block_decode_invalid(name, src, size):
    import lz4_amd as lz
    buf = (np.arange(size + 8) & 0xFF).astype(np.uint8)
    if src == b"":  # T10: the API returns (0, nil) for an empty source (block.go:45-47)
        assert lz.UncompressBlock(src, buf[:size]) == 0
    else:
        with pytest.raises(lz.ErrInvalidSourceShortBuffer):
            lz.UncompressBlock(src, buf[:size])
    assert (buf[size:] == (np.arange(size, size + 8) & 0xFF)).all()


@pytest.mark.parametrize("src,dic,want", kats.WITH_DICT)
def test_block_decode_with_dict(src, dic, want):
    import lz4_amd as lz
    dst = np.zeros(100, dtype=np.uint8)
    if want is None:
        with pytest.raises(lz.ErrInvalidSourceShortBuffer):
            lz.UncompressBlockWithDict(src, dst, dic)
    else:
        n = lz.UncompressBlockWithDict(src, dst, dic)
        assert dst[:n].tobytes() == want


def test_fuzz_corpus_matches_oracle_and_stays_in_bounds(oracle):
    import lz4_amd as lz
    seeds = json.load(open(os.path.join(GOLDEN, "fuzz_uncompress_corpus.json")))
    for name, hx in seeds.items():
        src = bytes.fromhex(hx)
        if not src:
            continue
        for size in (0, 16, 4096):
            buf = (np.arange(size + 8) & 0xFF).astype(np.uint8)
            want, wd = oracle.decodeBlock(src, size)
            try:
                n = lz.UncompressBlock(src, buf[:size])
                assert want >= 0 and n == want and buf[:n].tobytes() == wd[:n].tobytes(), name
            except lz.ErrInvalidSourceShortBuffer:
                assert want < 0, name
            assert (buf[size:] == (np.arange(size, size + 8) & 0xFF)).all(), name


@pytest.mark.parametrize("want,data", kats.XXH32)
def test_xxh32_kat(want, data):
    import lz4_amd as lz
    assert lz.ChecksumZero(data) == want


def test_xxh32_unaligned_and_sizes(oracle):
    import lz4_amd as lz
    rng = np.random.default_rng(5)
    blob = rng.integers(0, 256, 70000, dtype=np.uint8)
    for n in (0, 1, 3, 4, 15, 16, 17, 63, 64, 65, 1000, 65536, 69999):
        for off in (0, 1, 7):
            b = blob[off:off + n].tobytes()
            assert lz.ChecksumZero(b) == oracle.ChecksumZero(b)


@pytest.mark.parametrize("frame", sorted(kats.GOLDEN_FRAMES))
def test_golden_frames_decode_to_raw(frame):
    """TestReader (reader_test.go:22-138): frames made by the C lz4 tool."""
    import lz4_amd as lz
    raw = load(kats.GOLDEN_FRAMES[frame][0])
    assert lz.Reader(io.BytesIO(load(frame))).Read() == raw


def test_legacy_golden_frame():
    """TestReaderLegacy (reader_test.go:192-236)."""
    import lz4_amd as lz
    raw = load("bzImage_lz4_isolated")
    assert lz.Reader(io.BytesIO(load(kats.LEGACY_FRAME))).Read() == raw


@pytest.mark.parametrize("i", range(9))
def test_legacy_synthetic(i, oracle):
    import lz4_amd as lz
    name, frame, want = kats.legacy_frames(lambda d: oracle.CompressBlock(d, len(d) + 64)[1])[i]
    if want is None:
        with pytest.raises(Exception) as e:
            lz.frame.uncompress_frame(frame)
        with pytest.raises(ValueError):
            oracle.frame_uncompress(frame, 1 << 20)
        assert not isinstance(e.value, (AssertionError, TypeError)), name
    else:
        assert lz.frame.uncompress_frame(frame) == want, name
        assert oracle.frame_uncompress(frame, 1 << 20) == want


@pytest.mark.parametrize("name", sorted(kats.LONG_FRAMES))
def test_long_and_linked_goldens(name, oracle):
    """reader_test.go:40-46; the linked frame decodes serially on the device."""
    import lz4_amd as lz
    size, cs = kats.LONG_FRAMES[name]
    out = lz.Reader(io.BytesIO(load(name))).Read()
    assert len(out) == size and lz.ChecksumZero(out) == cs
    assert out == oracle.frame_uncompress(load(name), size + 16)


@pytest.mark.parametrize("i", range(7))
def test_linked_synthetic(i, oracle):
    import lz4_amd as lz
    name, frame, want = kats.linked_frames(oracle.ChecksumZero)[i]
    if want is None:
        with pytest.raises(ValueError):
            oracle.frame_uncompress(frame, 1 << 20)
        with pytest.raises(Exception) as e:
            lz.frame.uncompress_frame(frame)
        assert not isinstance(e.value, (AssertionError, TypeError)), name
    else:
        assert lz.frame.uncompress_frame(frame) == want, name


@pytest.mark.parametrize("i", range(9))
def test_legacy_write(i, oracle):
    """TestWriterLegacy (writer_test.go:256-299): bytes equal the oracle's, and read back."""
    import lz4_amd as lz
    name, data, readfrom, level, bsum = kats.legacy_write_cases(load)[i]
    out = io.BytesIO()
    zw = lz.Writer(out)
    zw.Apply(lz.LegacyOption(True), lz.CompressionLevelOption(level), lz.BlockChecksumOption(bsum))
    if readfrom:
        zw.ReadFrom(io.BytesIO(data))
    else:
        zw.Write(data)
    zw.Close()
    flags = (oracle.F_LEGACY | (oracle.F_READFROM if readfrom else 0)
             | (oracle.F_BLOCK_CHECKSUM if bsum else 0))
    assert out.getvalue() == oracle.frame_compress(data, flags=flags, level=level), name
    if not bsum:
        assert lz.Reader(io.BytesIO(out.getvalue())).Read() == data


@pytest.mark.parametrize("i", range(2))
def test_ragged_frames(i, oracle):
    """Short blocks in the middle of a frame: decoded slots are packed on the device."""
    import lz4_amd as lz

    def cb(d):
        try:
            return oracle.CompressBlock(d, len(d))[1]
        except oracle.ErrInvalidSourceShortBuffer:
            return b""
    name, frame, want = kats.ragged_frames(cb, oracle.ChecksumZero)[i]
    assert lz.frame.uncompress_frame(frame) == want, name
    bad = bytearray(frame)
    bad[-1] ^= 1
    if i == 1:
        with pytest.raises(lz.ErrInvalidFrameChecksum):
            lz.frame.uncompress_frame(bytes(bad))


def test_many_short_blocks_in_a_4mb_frame(oracle):
    """A Writer flushed every few bytes: 20,000 blocks of a 4 MiB-block frame.  Device
    memory must follow the data (a few MB), not n_blocks * block_size (80 GB)."""
    import struct
    import lz4_amd as lz
    hdr = bytes([0x64, 0x70])
    f = bytearray(struct.pack("<I", 0x184D2204) + hdr + bytes([(oracle.ChecksumZero(hdr) >> 8) & 0xFF]))
    want = bytearray()
    for i in range(20000):
        d = (b"short block %5d " % i) * (1 + i % 5)
        k, c = oracle.CompressBlock(d, oracle.CompressBlockBound(len(d)))
        if i % 3 == 0 or k >= len(d):
            f += struct.pack("<I", len(d) | 0x80000000) + d
        else:
            f += struct.pack("<I", k) + c
        want += d
    f += struct.pack("<I", 0) + struct.pack("<I", oracle.ChecksumZero(bytes(want)))
    assert lz.frame.uncompress_frame(bytes(f)) == bytes(want)
    assert oracle.frame_uncompress(bytes(f), len(want) + 16) == bytes(want)


def test_malformed_block_checksum():
    """reader_test.go: malformed.block.lz4 => invalid block checksum."""
    import lz4_amd as lz
    with pytest.raises(lz.ErrInvalidBlockChecksum):
        lz.Reader(io.BytesIO(load("malformed.block.lz4"))).Read()


def test_bad_header_checksum_and_magic():
    import lz4_amd as lz
    f = bytearray(load("gettysburg.txt.lz4"))
    f[6] ^= 1
    with pytest.raises(lz.ErrInvalidHeaderChecksum):
        lz.Reader(io.BytesIO(bytes(f))).Read()
    with pytest.raises(lz.ErrInvalidFrame):
        lz.Reader(io.BytesIO(b"\x00\x01\x02\x03\x04\x05\x06\x07")).Read()
    f = bytearray(load("gettysburg.txt.lz4"))
    f[-1] ^= 1
    with pytest.raises(lz.ErrInvalidFrameChecksum):
        lz.Reader(io.BytesIO(bytes(f))).Read()


@pytest.mark.parametrize("name", ["pg1661.txt", "random.data", "repeat.txt", "empty.txt",
                                  "bzImage_lz4_isolated", "issue102.data"])
@pytest.mark.parametrize("opts", [dict(), dict(block_checksum=True), dict(content_size=123),
                                  dict(block_size=1 << 16, block_checksum=True),
                                  dict(block_size=1 << 18, content_checksum=False),
                                  dict(block_size=1 << 16, readfrom=True),
                                  dict(block_size=1 << 16, level=512)])
def test_writer_frames_byte_identical_to_oracle(name, opts, oracle):
    """TestWriter (writer_test.go:19-92) option matrix; compares the frame BYTES
    with the oracle's Writer restatement, then reads it back."""
    import lz4_amd as lz
    if name in ("bzImage_lz4_isolated", "issue102.data") and opts.get("level"):
        pytest.skip("HC on 1 MB binary is covered by test_gpu_batch at test size")
    data = load(name)
    w = io.BytesIO()
    zw = lz.Writer(w)
    zw.Apply(lz.BlockSizeOption(opts.get("block_size", 1 << 22)),
             lz.BlockChecksumOption(opts.get("block_checksum", False)),
             lz.ChecksumOption(opts.get("content_checksum", True)),
             lz.SizeOption(opts.get("content_size", 0)),
             lz.CompressionLevelOption(opts.get("level", 0)))
    if opts.get("readfrom"):
        zw.ReadFrom(io.BytesIO(data))
    else:
        zw.Write(data)
    zw.Close()
    flags = ((1 if opts.get("block_checksum") else 0) | (2 if opts.get("content_checksum", True) else 0)
             | (4 if opts.get("content_size") else 0) | (8 if opts.get("readfrom") else 0))
    want = oracle.frame_compress(data, opts.get("block_size", 1 << 22), flags, opts.get("level", 0),
                                 opts.get("content_size", 0))
    assert w.getvalue() == want
    assert lz.Reader(io.BytesIO(w.getvalue())).Read() == data


def test_flush_header_is_7_bytes():
    """writer_test.go:243-246: an empty stream's header is exactly 7 bytes."""
    import lz4_amd as lz
    w = io.BytesIO()
    zw = lz.Writer(w)
    zw.Close()
    assert len(w.getvalue()) == 7 + 4 + 4  # header + end mark + content checksum
    assert w.getvalue()[7:] == b"\x00\x00\x00\x00\x05\x5d\xcc\x02"


CR_FILES = ["e.txt", "gettysburg.txt", "Mark.Twain-Tom.Sawyer.txt", "pg1661.txt", "pi.txt",
            "random.data", "repeat.txt", "issue102.data"]


@pytest.mark.parametrize("name", CR_FILES)
@pytest.mark.parametrize("option", ["BlockChecksumOption(true)", "SizeOption(123)"])
def test_compressing_reader(name, option, oracle):
    """TestCompressingReader (compressing_reader_test.go:13-78): read the frame out of a
    CompressingReader at Level1, decode it with Reader, compare; with SizeOption(123) the
    Reader reports Size() 123.  Also: the bytes are the oracle's Writer bytes."""
    import lz4_amd as lz
    raw = load(name)
    zcomp = lz.CompressingReader(io.BytesIO(raw))
    opt = lz.BlockChecksumOption(True) if option.startswith("Block") else lz.SizeOption(123)
    zcomp.Apply(opt, lz.CompressionLevelOption(lz.Level1))
    zout = bytearray()
    while True:                       # ioutil.ReadAll: growing reads
        b = zcomp.Read(4096 + len(zout))
        if not b:
            break
        zout += b
    flags = 2 | (1 if option.startswith("Block") else 4)
    assert bytes(zout) == oracle.frame_compress(raw, 1 << 22, flags, lz.Level1,
                                                0 if option.startswith("Block") else 123)
    zr = lz.Reader(io.BytesIO(bytes(zout)))
    assert zr.Read() == raw
    with pytest.raises(lz.ErrOptionClosedOrError):
        zcomp.Apply(lz.ChecksumOption(False))      # compressing_reader.go:60-62
    with pytest.raises(lz.ErrOptionNotApplicable):
        lz.CompressingReader(io.BytesIO(b"")).Apply(lz.ConcurrencyOption(2))   # options.go:136-147
    zcomp.Reset(io.BytesIO(b"abc"))                 # keeps the options
    again = zcomp.Read()
    assert lz.Reader(io.BytesIO(again)).Read() == b"abc"
    assert again[4] == zout[4] and again[5] == zout[5]   # same FLG / BD


def test_on_block_done_handlers(oracle):
    """OnBlockDoneOption (options.go:197-220): Writer and CompressingReader report
    len(block.Data) of every block (writer.go:127), ReadFrom also the bytes read for it
    (writer.go:234); Reader.WriteTo the decoded length of every block (reader.go:255)."""
    import lz4_amd as lz
    raw = load("pg1661.txt")           # 594,933 B = 9 x 64 KiB + 5,109
    bs = 1 << 16
    blocks = [raw[i:i + bs] for i in range(0, len(raw), bs)]
    want = []
    for b in blocks:                   # FrameDataBlock.Compress: dst = len(src); 0 => stored raw
        n = oracle.CompressBlock(b, len(b))[0]
        want.append(n if n > 0 else len(b))
    seen = []
    w = io.BytesIO()
    zw = lz.Writer(w)
    zw.Apply(lz.BlockSizeOption(bs), lz.OnBlockDoneOption(seen.append))
    zw.Write(raw)
    zw.Close()
    assert seen == want
    seen2 = []
    zw = lz.Writer(io.BytesIO())
    zw.Apply(lz.BlockSizeOption(bs), lz.OnBlockDoneOption(seen2.append))
    zw.ReadFrom(io.BytesIO(raw))
    zw.Close()
    assert seen2 == [x for pair in zip(want, map(len, blocks)) for x in pair]
    seen3 = []
    zc = lz.CompressingReader(io.BytesIO(raw))
    zc.Apply(lz.BlockSizeOption(bs), lz.OnBlockDoneOption(seen3.append))
    assert zc.Read() == w.getvalue() and seen3 == want
    got = []
    zr = lz.Reader(io.BytesIO(w.getvalue()))
    zr.Apply(lz.OnBlockDoneOption(got.append))
    sink = io.BytesIO()
    assert zr.WriteTo(sink) == len(raw) and sink.getvalue() == raw
    assert got == [len(b) for b in blocks]
    got = []
    zr = lz.Reader(io.BytesIO(w.getvalue()))
    zr.Apply(lz.OnBlockDoneOption(got.append))
    out = bytearray()
    while True:
        b = zr.Read(100000)            # straddles block ends: portions are reported
        if not b:
            break
        out += b
    assert bytes(out) == raw and sum(got) == len(raw) and max(got) <= bs
    # a stream that is a block multiple read through ReadFrom ends in an empty block (T9)
    seen4 = []
    zw = lz.Writer(io.BytesIO())
    zw.Apply(lz.BlockSizeOption(bs), lz.OnBlockDoneOption(seen4.append))
    zw.ReadFrom(io.BytesIO(raw[:2 * bs]))
    zw.Close()
    assert seen4 == [want[0], bs, want[1], bs, 0, 0]
    # legacy frames: 8 MiB blocks, len(block.Data) again
    seen5 = []
    lw = io.BytesIO()
    zw = lz.Writer(lw)
    zw.Apply(lz.LegacyOption(True), lz.OnBlockDoneOption(seen5.append))
    zw.Write(raw)
    zw.Close()
    assert seen5 == [len(lw.getvalue()) - 8]


@pytest.mark.parametrize("level", [0, 512])
def test_frame_compress_equals_reference_golden_random(level):
    """testdata/random.data.lz4 (a reference fixture) is all stored blocks: the HIP Writer path
    must reproduce it byte for byte -- header, raw size words, end mark, content checksum
    (lz4stream/block.go:238-276, frame.go:59-74)."""
    from lz4_amd import _lib
    lib = _lib.load()
    src = np.frombuffer(load("random.data"), dtype=np.uint8)
    cap = lib.lz4hip_frame_bound(src.size, 1 << 16, _lib.F_CONTENT_CHECKSUM)
    dst = np.zeros(cap, dtype=np.uint8)
    k = lib.lz4hip_frame_compress(src.ctypes.data, src.size, 1 << 16, _lib.F_CONTENT_CHECKSUM, level, 0,
                                  dst.ctypes.data, cap)
    assert k > 0, lib.lz4hip_last_error()
    assert dst[:k].tobytes() == load("random.data.lz4")
