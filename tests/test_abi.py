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
    assert lib.lz4hip_uncompress_blocks(None, None, None, 4, None, 0, 0, None, None) == _lib.ERR_ARG
    assert lib.lz4hip_hc_workspace_bytes(1) == 2 * 65536 * 4


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
