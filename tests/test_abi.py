"""CPU-only checks of the drop-in boundary: liblz4hip.so loads, exports every
symbol include/lz4hip.h declares, and its host-only helpers (bound, frame
header/trailer) agree with the oracle and the reference's known answers.
No compute entry point is called here (no GPU)."""
import os
import re

import numpy as np
import pytest

import kats
from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "lz4hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lz4hip_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from lz4_amd import _lib
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "missing export: " + n
    assert set(names) == set(_lib.SYMBOLS), set(names) ^ set(_lib.SYMBOLS)


def test_compress_bound_and_constants(oracle):
    import lz4_amd as lz
    for n in (0, 1, 254, 255, 256, 65536, 1 << 22, 594933):
        assert lz.CompressBlockBound(n) == oracle.CompressBlockBound(n) == n + n // 255 + 16
    # lz4.go:16-24, options.go:154-165
    assert (lz.Block64Kb, lz.Block256Kb, lz.Block1Mb, lz.Block4Mb) == (1 << 16, 1 << 18, 1 << 20, 1 << 22)
    assert lz.Fast == 0 and lz.Level1 == 512 and lz.Level9 == 131072


@pytest.mark.parametrize("bs,want", kats.HEADERS)
def test_frame_header_known_answers(bs, want):
    from lz4_amd import _lib
    lib = _lib.load()
    buf = np.zeros(16, dtype=np.uint8)
    k = lib.lz4hip_frame_header(buf.ctypes.data, bs, _lib.F_CONTENT_CHECKSUM, 0)
    assert k == 7 and buf[:4].tobytes() == b"\x04\x22\x4d\x18" and buf[4:7].tobytes() == want


def test_frame_header_and_trailer_match_oracle(oracle):
    from lz4_amd import _lib
    lib = _lib.load()
    for bs in (1 << 16, 1 << 18, 1 << 20, 1 << 22):
        for flags in range(8):
            for size in (0, 123, 1 << 40):
                buf = np.zeros(16, dtype=np.uint8)
                k = lib.lz4hip_frame_header(buf.ctypes.data, bs, flags, size)
                assert buf[:k].tobytes() == oracle.frame_header(bs, flags, size)
    t = np.zeros(8, dtype=np.uint8)
    assert lib.lz4hip_frame_trailer(t.ctypes.data, 0, 0x11223344) == 4 and not t[:4].any()
    assert lib.lz4hip_frame_trailer(t.ctypes.data, _lib.F_CONTENT_CHECKSUM, 0x11223344) == 8
    assert t.tobytes() == b"\x00\x00\x00\x00\x44\x33\x22\x11"


def test_argument_validation_without_gpu():
    """Shape checks happen on the host before any launch."""
    from lz4_amd import _lib
    lib = _lib.load()
    assert lib.lz4hip_compress_fast_blocks(16, 100, 12345, 0, 16, 65536, 16, None, 0, None) == _lib.ERR_ARG
    assert b"block size" in lib.lz4hip_last_error()
    assert lib.lz4hip_compress_fast_blocks(17, 100, 1 << 16, 0, 16, 65536, 16, None, 0, None) == _lib.ERR_ARG
    assert lib.lz4hip_compress_fast_blocks(16, 100, 1 << 16, 0, 16, 1024, 16, None, 0, None) == _lib.ERR_ARG
    assert lib.lz4hip_uncompress_blocks(None, 0, None, None, 4, None, 0, 0, None, 0, None, None) == _lib.ERR_ARG
    # heads + 64 u16 links per entry (one 128-byte line), + 256 bytes for the block counter of the schedule
    assert lib.lz4hip_hc_workspace_bytes(1) == 65536 * 4 + 65536 * 32 * 2 + 65536 + 256  # heads, 32 links per position, position map, counter


def test_xxh32_host_update_chain_without_gpu(oracle):
    """lz4hip_xxh32_host_update is XXHZero.Write / Sum32 on the calling thread (no GPU): pieces
    chained through the 6-word state equal the oracle's one-shot checksum and its streaming one,
    and a handed-over totalLen at the 2^32 boundary takes Sum32's short-input branch as the
    reference does (xxh32zero.go:113-118, SURVEY trap T6)."""
    import ctypes as C
    from lz4_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(3)
    data = rng.integers(0, 256, 100_003, dtype=np.uint8)
    for cuts in ([0, 100_003], [0, 16, 100_003], [0, 4096, 65536, 99_984, 100_003], [0, 0, 48, 100_003]):
        st = np.zeros(6, dtype=np.uint32)
        dg = C.c_uint32()
        for i in range(len(cuts) - 1):
            a, b = cuts[i], cuts[i + 1]
            last = i == len(cuts) - 2
            piece = data[a:b]
            rc = lib.lz4hip_xxh32_host_update(piece.ctypes.data if piece.size else None, piece.size,
                                              st.ctypes.data, int(i == 0), int(last),
                                              C.addressof(dg) if last else None)
            assert rc == 0, lib.lz4hip_last_error()
        assert dg.value == oracle.ChecksumZero(data), cuts
    # a non-final piece must be a stripe multiple
    st = np.zeros(6, dtype=np.uint32)
    assert lib.lz4hip_xxh32_host_update(data.ctypes.data, 17, st.ctypes.data, 1, 0, None) == _lib.ERR_ARG
    # T6: state handed over with totalLen = 2^32 - 32, then 32 + 5 bytes: uint32(totalLen) = 5 < 16
    h = oracle.XXHZero()
    h.Write(data[:64])
    h.set_total_len((1 << 32) - 32)
    st = np.zeros(6, dtype=np.uint32)
    st[:4] = np.frombuffer(h._s[0:16].tobytes(), dtype=np.uint32)
    st[4], st[5] = ((1 << 32) - 32) & 0xFFFFFFFF, 0
    tail = data[64:64 + 37]
    h.Write(tail)
    dg = C.c_uint32()
    assert lib.lz4hip_xxh32_host_update(tail.ctypes.data, 37, st.ctypes.data, 0, 1, C.addressof(dg)) == 0
    assert dg.value == h.Sum32()
    assert (int(st[4]) | int(st[5]) << 32) == (1 << 32) + 5


def test_bad_arguments_are_not_the_codec_verdict():
    """-1 from the single-block calls means ErrInvalidSourceShortBuffer, which the frame
    step turns into "store raw" (lz4stream/block.go:238-245): a call that fails for another
    reason must return a different code (checked before any HIP call: runs without a GPU)."""
    import ctypes as C
    from lz4_amd import _lib
    lib = _lib.load()
    dst = np.zeros(64, dtype=np.uint8)
    assert _lib.ERR_BADARG not in (-1, 0) and _lib.ERR_HIP not in (-1, 0)
    assert lib.lz4hip_compress_block(None, 10, dst.ctypes.data, 64) == _lib.ERR_BADARG
    assert lib.lz4hip_compress_block_hc(None, 10, dst.ctypes.data, 64, 16) == _lib.ERR_BADARG
    assert lib.lz4hip_compress_block(dst.ctypes.data, (1 << 30) + 1, dst.ctypes.data, 64) == _lib.ERR_BADARG
    assert lib.lz4hip_uncompress_block(None, 10, dst.ctypes.data, 64, None, 0) == _lib.ERR_BADARG
    out = C.c_uint32(77)
    assert lib.lz4hip_xxh32_checked(None, 10, C.addressof(out)) == _lib.ERR_BADARG and out.value == 77
    assert lib.lz4hip_stream_create(12345, 0, 0, 4) is None and b"block size" in lib.lz4hip_last_error()
    assert lib.lz4hip_stream_create(1 << 16, 0, 0, 0) is None
    assert lib.lz4hip_stream_compress_blocks(None, None, 0, None, 0, None) == _lib.ERR_ARG


def test_no_cpu_fallback_in_product_path():
    """The package must not import, link or execute the oracle."""
    pkg = os.path.join(ROOT, "lz4_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in text.lower() or f == "corpus.py", (dirpath, f)


def test_options_mirror_reference_errors():
    import io

    import lz4_amd as lz
    from lz4_amd import frame
    w = lz.Writer(io.BytesIO())
    with pytest.raises(lz.ErrOptionInvalidBlockSize):
        w.Apply(lz.BlockSizeOption(12345))
    with pytest.raises(frame.ErrOptionInvalidCompressionLevel):
        w.Apply(lz.CompressionLevelOption(3))
    with pytest.raises(frame.ErrOptionNotApplicable):
        lz.Reader(io.BytesIO()).Apply(lz.BlockSizeOption(lz.Block64Kb))
    w.Apply(lz.BlockSizeOption(lz.Block64Kb), lz.ChecksumOption(False), lz.BlockChecksumOption(True),
            lz.CompressionLevelOption(lz.Level9), lz.ConcurrencyOption(-1), lz.SizeOption(5))
    assert (w.block_size, w.content_checksum, w.block_checksum, w.level, w.content_size) == \
        (1 << 16, False, True, 131072, 5)


def test_frame_parse_header_host_only():
    """Frame.ParseHeaders + FrameDescriptor.initR (frame.go:80-111,169-200): host code."""
    import ctypes as C
    import struct
    from lz4_amd import _lib
    lib = _lib.load()

    def parse(b):
        bs, fl, cs, hl = C.c_uint32(), C.c_uint32(), C.c_uint64(), C.c_size_t()
        buf = (C.c_uint8 * max(len(b), 1)).from_buffer_copy(bytes(b) or b"\0")
        rc = lib.lz4hip_frame_parse_header(C.addressof(buf), len(b), C.addressof(bs), C.addressof(fl),
                                           C.addressof(cs), C.addressof(hl))
        return rc, bs.value, fl.value, cs.value, hl.value

    magic = struct.pack("<I", 0x184D2204)
    for bs, hdr in kats.HEADERS:                      # frame_test.go:19-22
        assert parse(magic + hdr) == (0, bs, _lib.F_CONTENT_CHECKSUM, 0, 7)
        bad = bytearray(magic + hdr)
        bad[6] ^= 0x10
        assert parse(bad)[0] == _lib.ERR_HEADER_CHECKSUM
        assert parse((magic + hdr)[:6])[0] == _lib.ERR_SHORT
    # built by the library: every flag combination parses back
    for flags in range(8):
        h = (C.c_uint8 * 16)()
        k = lib.lz4hip_frame_header(C.addressof(h), 1 << 18, flags, 0x1122334455)
        rc, bs, fl, cs, hl = parse(bytes(h[:k]))
        assert (rc, bs, fl, hl) == (0, 1 << 18, flags, k)
        assert cs == (0x1122334455 if flags & _lib.F_CONTENT_SIZE else 0)
    # legacy magic, skippable frame in front, bad magic, linked flag, bad block size index
    assert parse(struct.pack("<I", 0x184C2102))[:3] == (0, 8 << 20, _lib.F_LEGACY)
    skip = struct.pack("<II", 0x184D2A53, 5) + b"hello"
    assert parse(skip + magic + kats.HEADERS[0][1]) == (0, 1 << 16, _lib.F_CONTENT_CHECKSUM, 0, 13 + 7)
    assert parse(b"PK\x03\x04abcdef")[0] == _lib.ERR_FRAME
    linked = bytes([0x44, 0x40])
    hc = (oracle_xxh32(linked) >> 8) & 0xFF
    assert parse(magic + linked + bytes([hc]))[2] == _lib.F_CONTENT_CHECKSUM | _lib.F_LINKED
    small = bytes([0x64, 0x30])
    assert parse(magic + small + bytes([(oracle_xxh32(small) >> 8) & 0xFF]))[0] == _lib.ERR_BLOCK_SIZE


def oracle_xxh32(b):
    from oracle import pyoracle as o
    return o.ChecksumZero(b)


def test_fixture_list_is_complete():
    """tests/golden/make_fixtures.py names every committed fixture (and only those)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_fixtures", os.path.join(ROOT, "tests", "golden", "make_fixtures.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    m.check()
