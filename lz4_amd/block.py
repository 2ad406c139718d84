"""Host-side mirror of pierrec/lz4's block API (lz4.go:27-131) over the C ABI.

Same names, argument meaning and error behaviour as the Go functions:
``(n, nil)`` -> returns n; ``(0, nil)`` -> returns 0;
``(0, ErrInvalidSourceShortBuffer)`` -> raises ErrInvalidSourceShortBuffer.
``dst`` is a writable buffer (bytearray / numpy uint8 array) like Go's ``dst []byte``.
All compute runs in liblz4hip.so on the GPU; there is no CPU path.
"""
import numpy as np

from . import _lib


class ErrInvalidSourceShortBuffer(Exception):
    """lz4.ErrInvalidSourceShortBuffer (lz4.go:134-136)."""

    def __str__(self):
        return "lz4: invalid source or destination buffer too short"


# CompressionLevel constants: options.go:154-165 (the value IS the HC chain depth).
Fast = 0
Level1, Level2, Level3, Level4, Level5 = 1 << 9, 1 << 10, 1 << 11, 1 << 12, 1 << 13
Level6, Level7, Level8, Level9 = 1 << 14, 1 << 15, 1 << 16, 1 << 17

Block64Kb, Block256Kb, Block1Mb, Block4Mb = 1 << 16, 1 << 18, 1 << 20, 1 << 22


def _ro(b):
    if isinstance(b, np.ndarray):
        return np.ascontiguousarray(b, dtype=np.uint8)
    return np.frombuffer(bytes(b), dtype=np.uint8)


def _rw(b):
    a = b if isinstance(b, np.ndarray) else np.frombuffer(b, dtype=np.uint8)
    if not a.flags.writeable or not a.flags.c_contiguous:
        raise TypeError("dst must be a writable contiguous buffer")
    return a


def _ptr(a):
    return a.ctypes.data if a.size else None


def _ret(r):
    if r == -1:
        raise ErrInvalidSourceShortBuffer()
    if r < 0:
        raise _lib.Lz4HipError(int(r), _lib.last_error())
    return int(r)


def CompressBlockBound(n):
    """lz4.CompressBlockBound (lz4.go:27-29)."""
    return _lib.load().lz4hip_compress_bound(n)


def CompressBlock(src, dst):
    """lz4.CompressBlock / Compressor.CompressBlock (lz4.go:70-94)."""
    _lib.require_gpu()
    s, d = _ro(src), _rw(dst)
    return _ret(_lib.load().lz4hip_compress_block(_ptr(s), s.size, _ptr(d), d.size))


def CompressBlockHC(src, dst, depth):
    """lz4.CompressBlockHC / CompressorHC.CompressBlock (lz4.go:121-131)."""
    _lib.require_gpu()
    s, d = _ro(src), _rw(dst)
    return _ret(_lib.load().lz4hip_compress_block_hc(_ptr(s), s.size, _ptr(d), d.size, depth))


def UncompressBlock(src, dst):
    """lz4.UncompressBlock (lz4.go:37-39)."""
    return UncompressBlockWithDict(src, dst, b"")


def UncompressBlockWithDict(src, dst, dict):
    """lz4.UncompressBlockWithDict (lz4.go:47-49)."""
    _lib.require_gpu()
    s, d, dc = _ro(src), _rw(dst), _ro(dict)
    return _ret(_lib.load().lz4hip_uncompress_block(_ptr(s), s.size, _ptr(d), d.size,
                                                    _ptr(dc), dc.size))


def ChecksumZero(data):
    """xxh32.ChecksumZero (internal/xxh32/xxh32zero.go:142)."""
    import ctypes
    _lib.require_gpu()
    a = _ro(data)
    out = ctypes.c_uint32()
    rc = _lib.load().lz4hip_xxh32_checked(_ptr(a), a.size, ctypes.addressof(out))
    if rc < 0:  # never a silent 0: a failed call is an error, not a checksum
        raise _lib.Lz4HipError(int(rc), _lib.last_error())
    return out.value


class Compressor:
    """lz4.Compressor (lz4.go:57-72). Stateless here: the table lives in LDS."""

    def CompressBlock(self, src, dst):
        return CompressBlock(src, dst)


class CompressorHC:
    """lz4.CompressorHC (lz4.go:103-123)."""

    def __init__(self, Level=0):
        self.Level = Level

    def CompressBlock(self, src, dst):
        return CompressBlockHC(src, dst, self.Level)
