"""ctypes binding of the CPU oracle (oracle/lz4_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, by ``__graft_entry__.smoke()`` and
by ``bench.py``'s ``cpu_baseline`` leg, never by the ``lz4_amd`` package.

Function names follow the reference's Go API (``lz4.go:27-131``) so the parity
tests read like the reference's own tests.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liblz4oracle.so")

F_BLOCK_CHECKSUM = 1
F_CONTENT_CHECKSUM = 2
F_CONTENT_SIZE = 4
F_READFROM = 8
F_LEGACY = 16


class ErrInvalidSourceShortBuffer(Exception):
    """lz4errors.ErrInvalidSourceShortBuffer (internal/lz4errors/errors.go)."""


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liblz4oracle.so"])


def _load():
    if not os.path.exists(_SO):
        build()
    lib = C.CDLL(_SO)
    u8p, sz, i64, u32 = C.c_void_p, C.c_size_t, C.c_int64, C.c_uint32
    lib.lz4o_compress_bound.restype = sz
    lib.lz4o_compress_bound.argtypes = [sz]
    lib.lz4o_block_hash.restype = u32
    lib.lz4o_block_hash.argtypes = [C.c_uint64]
    lib.lz4o_block_hash_hc.restype = u32
    lib.lz4o_block_hash_hc.argtypes = [u32]
    lib.lz4o_compress_fast.restype = i64
    lib.lz4o_compress_fast.argtypes = [u8p, sz, u8p, sz]
    lib.lz4o_compress_hc.restype = i64
    lib.lz4o_compress_hc.argtypes = [u8p, sz, u8p, sz, u32]
    lib.lz4o_decode_block.restype = i64
    lib.lz4o_decode_block.argtypes = [u8p, sz, u8p, sz, u8p, sz]
    lib.lz4o_uncompress_block.restype = i64
    lib.lz4o_uncompress_block.argtypes = [u8p, sz, u8p, sz, u8p, sz]
    lib.lz4o_xxh32.restype = u32
    lib.lz4o_xxh32.argtypes = [u8p, sz]
    lib.lz4o_xxh32_reset.argtypes = [u8p]
    lib.lz4o_xxh32_write.argtypes = [u8p, u8p, sz]
    lib.lz4o_xxh32_sum32.restype = u32
    lib.lz4o_xxh32_sum32.argtypes = [u8p]
    lib.lz4o_frame_header.restype = sz
    lib.lz4o_frame_header.argtypes = [u8p, u32, u32, C.c_uint64]
    lib.lz4o_frame_bound.restype = sz
    lib.lz4o_frame_bound.argtypes = [sz, u32, u32]
    lib.lz4o_frame_compress.restype = i64
    lib.lz4o_frame_compress.argtypes = [u8p, sz, u32, u32, u32, C.c_uint64, u8p, sz]
    lib.lz4o_frame_uncompress.restype = i64
    lib.lz4o_frame_uncompress.argtypes = [u8p, sz, u8p, sz]
    lib.lz4o_compress_blocks_mt.restype = C.c_int
    lib.lz4o_compress_blocks_mt.argtypes = [u8p, sz, u32, u32, u8p, sz, u8p, u8p, C.c_int]
    lib.lz4o_uncompress_blocks_mt.restype = C.c_int
    lib.lz4o_uncompress_blocks_mt.argtypes = [u8p, sz, u8p, sz, u32, u8p, u8p, C.c_int]
    return lib


lib = _load()


def _arr(b):
    """bytes-like -> contiguous uint8 ndarray (no copy for ndarrays)."""
    if isinstance(b, np.ndarray):
        return np.ascontiguousarray(b, dtype=np.uint8)
    return np.frombuffer(bytes(b), dtype=np.uint8)


def _p(a):
    return a.ctypes.data if a.size else None


def CompressBlockBound(n):
    return lib.lz4o_compress_bound(n)


def _ret(r):
    if r < 0:
        raise ErrInvalidSourceShortBuffer()
    return int(r)


def CompressBlock(src, dst_len=None):
    """lz4.CompressBlock (lz4.go:92-94). Returns (n, bytes[:n])."""
    s = _arr(src)
    if dst_len is None:
        dst_len = CompressBlockBound(s.size)
    d = np.zeros(max(dst_len, 1), dtype=np.uint8)
    n = _ret(lib.lz4o_compress_fast(_p(s), s.size, d.ctypes.data, dst_len))
    return n, d[:n].tobytes()


def CompressBlockHC(src, depth, dst_len=None):
    """lz4.CompressBlockHC (lz4.go:129-131). Returns (n, bytes[:n])."""
    s = _arr(src)
    if dst_len is None:
        dst_len = CompressBlockBound(s.size)
    d = np.zeros(max(dst_len, 1), dtype=np.uint8)
    n = _ret(lib.lz4o_compress_hc(_p(s), s.size, d.ctypes.data, dst_len, depth))
    return n, d[:n].tobytes()


def decodeBlock(src, dst_len, dict=b"", guard=8):
    """decodeBlock (decode_other.go:10). Returns (ret, dst ndarray incl. guard bytes)."""
    s = _arr(src)
    dc = _arr(dict)
    d = (np.arange(dst_len + guard) & 0xFF).astype(np.uint8)
    r = lib.lz4o_decode_block(d.ctypes.data, dst_len, _p(s), s.size, _p(dc), dc.size)
    return int(r), d


def UncompressBlock(src, dst_len, dict=b""):
    """lz4.UncompressBlock / UncompressBlockWithDict (lz4.go:37-49). Returns bytes."""
    s = _arr(src)
    dc = _arr(dict)
    d = np.zeros(max(dst_len, 1), dtype=np.uint8)
    n = _ret(lib.lz4o_uncompress_block(_p(s), s.size, d.ctypes.data, dst_len, _p(dc), dc.size))
    return d[:n].tobytes()


def ChecksumZero(data):
    """xxh32.ChecksumZero (internal/xxh32/xxh32zero.go:142)."""
    a = _arr(data)
    return lib.lz4o_xxh32(_p(a), a.size)


class XXHZero:
    """xxh32.XXHZero streaming hash (internal/xxh32/xxh32zero.go:22-139)."""

    def __init__(self):
        self._s = np.zeros(64, dtype=np.uint8)
        lib.lz4o_xxh32_reset(self._s.ctypes.data)

    def Reset(self):
        lib.lz4o_xxh32_reset(self._s.ctypes.data)

    def Write(self, data):
        a = _arr(data)
        lib.lz4o_xxh32_write(self._s.ctypes.data, _p(a), a.size)

    def Sum32(self):
        return lib.lz4o_xxh32_sum32(self._s.ctypes.data)

    def set_total_len(self, n):
        """Test hook: overwrite totalLen (offset 16 in the state struct)."""
        self._s[16:24] = np.frombuffer(np.uint64(n).tobytes(), dtype=np.uint8)


def frame_header(block_size, flags=F_CONTENT_CHECKSUM, content_size=0):
    d = np.zeros(16, dtype=np.uint8)
    k = lib.lz4o_frame_header(d.ctypes.data, block_size, flags, content_size)
    return d[:k].tobytes()


def frame_compress(src, block_size=1 << 22, flags=F_CONTENT_CHECKSUM, level=0, content_size=0):
    """Writer.Write(src); Writer.Close() -- or io.Copy with F_READFROM."""
    s = _arr(src)
    cap = lib.lz4o_frame_bound(s.size, block_size, flags)
    d = np.zeros(cap, dtype=np.uint8)
    k = lib.lz4o_frame_compress(_p(s), s.size, block_size, flags, level, content_size,
                                d.ctypes.data, cap)
    if k < 0:
        raise ValueError("lz4o_frame_compress failed: %d" % k)
    return d[:k].tobytes()


def frame_uncompress(frame, dst_cap):
    """io.ReadAll(lz4.NewReader(frame)). Returns bytes or raises ValueError(code)."""
    s = _arr(frame)
    d = np.zeros(max(dst_cap, 1), dtype=np.uint8)
    k = lib.lz4o_frame_uncompress(_p(s), s.size, d.ctypes.data, dst_cap)
    if k < 0:
        raise ValueError(int(k))
    return d[:k].tobytes()


def compress_blocks(src, block_size, level=0, threads=1, with_xxh=True):
    """Frame-block semantics per block (lz4stream/block.go:224-252).
    Returns (payload ndarray [n_blocks, block_size], size_words, block_xxh)."""
    s = _arr(src)
    nb = (s.size + block_size - 1) // block_size
    out = np.zeros((max(nb, 1), block_size), dtype=np.uint8)
    sw = np.zeros(max(nb, 1), dtype=np.uint32)
    xx = np.zeros(max(nb, 1), dtype=np.uint32)
    rc = lib.lz4o_compress_blocks_mt(_p(s), s.size, block_size, level, out.ctypes.data,
                                     block_size, sw.ctypes.data,
                                     xx.ctypes.data if with_xxh else None, threads)
    assert rc == 0
    return out[:nb], sw[:nb], xx[:nb]


def uncompress_blocks(payload, size_words, block_size, threads=1):
    """payload: ndarray [n_blocks, stride]. Returns (out [n_blocks, block_size], out_len)."""
    p = np.ascontiguousarray(payload, dtype=np.uint8)
    sw = np.ascontiguousarray(size_words, dtype=np.uint32)
    nb = sw.size
    out = np.zeros((max(nb, 1), block_size), dtype=np.uint8)
    ol = np.zeros(max(nb, 1), dtype=np.int32)
    rc = lib.lz4o_uncompress_blocks_mt(p.ctypes.data, p.shape[1], sw.ctypes.data, nb, block_size,
                                       out.ctypes.data, ol.ctypes.data, threads)
    assert rc == 0
    return out[:nb], ol[:nb]
