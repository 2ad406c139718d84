"""Host-side mirror of lz4.Writer / lz4.Reader / lz4.CompressingReader (writer.go,
reader.go, compressing_reader.go) and their options (options.go) over the C ABI's
whole-frame entry points.

The Go Writer cuts the stream into blocks and compresses them one at a time (or
ConcurrencyOption(n) at a time); here the stream is buffered and handed to the
GPU as ONE batch of independent blocks at Close() -- the output bytes are the
ones the Go Writer produces (ordering contract of lz4stream/block.go:31-57).
"""
import ctypes as C

import numpy as np

from . import _lib
from .block import Fast


class ErrInvalidFrame(Exception):
    """lz4.ErrInvalidFrame"""


class ErrInvalidHeaderChecksum(Exception):
    """lz4.ErrInvalidHeaderChecksum"""


class ErrInvalidBlockChecksum(Exception):
    """lz4.ErrInvalidBlockChecksum"""


class ErrInvalidFrameChecksum(Exception):
    """lz4.ErrInvalidFrameChecksum"""


class ErrOptionInvalidBlockSize(Exception):
    """lz4.ErrOptionInvalidBlockSize"""


class ErrOptionInvalidCompressionLevel(Exception):
    """lz4.ErrOptionInvalidCompressionLevel"""


class ErrOptionNotApplicable(Exception):
    """lz4.ErrOptionNotApplicable"""


class ErrOptionClosedOrError(Exception):
    """lz4.ErrOptionClosedOrError"""


_FRAME_ERRORS = {
    _lib.ERR_FRAME: ErrInvalidFrame,
    _lib.ERR_HEADER_CHECKSUM: ErrInvalidHeaderChecksum,
    _lib.ERR_BLOCK_SIZE: ErrOptionInvalidBlockSize,
    _lib.ERR_BLOCK_CHECKSUM: ErrInvalidBlockChecksum,
    _lib.ERR_FRAME_CHECKSUM: ErrInvalidFrameChecksum,
}

_VALID_BLOCK_SIZES = (1 << 16, 1 << 18, 1 << 20, 1 << 22)
_VALID_LEVELS = (0,) + tuple(1 << k for k in range(9, 18))


# ---- options: options.go:14-242.  Each returns a function applied by Apply(). ----
def BlockSizeOption(size):
    """options.go:73-91"""
    def opt(a):
        if not isinstance(a, (Writer, CompressingReader)):
            raise ErrOptionNotApplicable()
        if size not in _VALID_BLOCK_SIZES:
            raise ErrOptionInvalidBlockSize()
        a.block_size = size
    return opt


def BlockChecksumOption(flag):
    """options.go:93-107"""
    def opt(a):
        if not isinstance(a, (Writer, CompressingReader)):
            raise ErrOptionNotApplicable()
        a.block_checksum = bool(flag)
    return opt


def ChecksumOption(flag):
    """options.go:109-123 (content checksum; default on)"""
    def opt(a):
        if not isinstance(a, (Writer, CompressingReader)):
            raise ErrOptionNotApplicable()
        a.content_checksum = bool(flag)
    return opt


def SizeOption(size):
    """options.go:125-141: content size in the header when > 0"""
    def opt(a):
        if not isinstance(a, (Writer, CompressingReader)):
            raise ErrOptionNotApplicable()
        a.content_size = int(size)
    return opt


def ConcurrencyOption(n):
    """options.go:129-149.  Accepted for API compatibility: the GPU path always
    processes all buffered blocks as one batch, and the Go output does not
    depend on n either."""
    def opt(a):
        if not isinstance(a, (Writer, Reader)):
            raise ErrOptionNotApplicable()
        a.concurrency = n
    return opt


def OnBlockDoneOption(handler):
    """options.go:197-220: handler(size) per processed block.  Writer and
    CompressingReader report each block's stored payload length (writer.go:127,
    compressing_reader.go:126), Writer.ReadFrom also the bytes read for it
    (writer.go:234); Reader reports decoded bytes (reader.go:154,255).  The blocks are
    processed as one batch here, so the calls come in block order once the batch is done."""
    if handler is None:
        handler = _on_block_done

    def opt(a):
        if not isinstance(a, (Writer, Reader, CompressingReader)):
            raise ErrOptionNotApplicable()
        a.handler = handler
    return opt


def _on_block_done(size):
    """options.go:195"""


def CompressionLevelOption(level):
    """options.go:167-195"""
    def opt(a):
        if not isinstance(a, (Writer, CompressingReader)):
            raise ErrOptionNotApplicable()
        if level not in _VALID_LEVELS:
            raise ErrOptionInvalidCompressionLevel()
        a.level = level
    return opt


def LegacyOption(legacy):
    """options.go:222-242: legacy frame (magic 0x184C2102, 8 MiB blocks), Writer only"""
    def opt(a):
        if not isinstance(a, Writer):
            raise ErrOptionNotApplicable()
        a.legacy = bool(legacy)
    return opt


def _ro(b):
    if isinstance(b, np.ndarray):
        return np.ascontiguousarray(b, dtype=np.uint8)
    return np.frombuffer(bytes(b), dtype=np.uint8)


def compress_frame(data, block_size=1 << 22, block_checksum=False, content_checksum=True,
                   content_size=0, level=Fast, readfrom=False, legacy=False, block_lens=None):
    """One frame for `data`: Writer.Write(data); Writer.Close()  (or io.Copy when
    readfrom=True: writer.go:199-242).  A list passed as block_lens receives the stored
    payload length of every block written."""
    _lib.require_gpu()
    lib = _lib.load()
    s = _ro(data)
    flags = ((_lib.F_BLOCK_CHECKSUM if block_checksum else 0)
             | (_lib.F_CONTENT_CHECKSUM if content_checksum else 0)
             | (_lib.F_CONTENT_SIZE if content_size > 0 else 0)
             | (_lib.F_READFROM if readfrom else 0)
             | (_lib.F_LEGACY if legacy else 0))
    cap = lib.lz4hip_frame_bound(s.size, block_size, flags)
    d = np.empty(cap, dtype=np.uint8)
    if block_lens is None:
        k = lib.lz4hip_frame_compress(s.ctypes.data if s.size else None, s.size, block_size, flags,
                                      level, content_size, d.ctypes.data, cap)
    else:
        max_blocks = s.size // (8 << 20 if legacy else block_size) + 2
        lens = np.empty(max_blocks, dtype=np.uint32)
        nb = C.c_size_t(0)
        k = lib.lz4hip_frame_compress_lens(s.ctypes.data if s.size else None, s.size, block_size,
                                           flags, level, content_size, d.ctypes.data, cap,
                                           lens.ctypes.data, max_blocks, C.addressof(nb))
        if k >= 0:
            block_lens.extend(int(x) for x in lens[:nb.value])
    if k < 0:
        raise _FRAME_ERRORS.get(k, _lib.Lz4HipError)(*(() if k in _FRAME_ERRORS else (k, _lib.last_error())))
    return d[:k].tobytes()


def uncompress_frame(frame, size_hint=None, block_lens=None):
    """io.ReadAll(lz4.NewReader(frame)).  A list passed as block_lens receives each
    block's decoded length in stream order."""
    _lib.require_gpu()
    lib = _lib.load()
    s = _ro(frame)
    # Upper bound of the decoded size: every block decodes to at most the frame's
    # block size; count size words cheaply from the descriptor when no hint is given.
    cap = size_hint if size_hint is not None else _decoded_bound(s)
    d = np.empty(max(cap, 1), dtype=np.uint8)
    if block_lens is None:
        k = lib.lz4hip_frame_uncompress(s.ctypes.data if s.size else None, s.size, d.ctypes.data, cap)
    else:
        max_blocks = s.size // 4 + 1    # every block costs at least its size word
        lens = np.empty(max_blocks, dtype=np.uint32)
        nb = C.c_size_t(0)
        k = lib.lz4hip_frame_uncompress_lens(s.ctypes.data if s.size else None, s.size, d.ctypes.data,
                                             cap, lens.ctypes.data, max_blocks, C.addressof(nb))
        if k >= 0:
            block_lens.extend(int(x) for x in lens[:nb.value])
    if k < 0:
        if k in _FRAME_ERRORS:
            raise _FRAME_ERRORS[k](_lib.last_error())
        if k == _lib.ERR_BLOCK:
            from .block import ErrInvalidSourceShortBuffer
            raise ErrInvalidSourceShortBuffer()
        raise _lib.Lz4HipError(int(k), _lib.last_error())
    return d[:k].tobytes()


def _decoded_bound(s):
    """Walk the size words (lz4stream/block.go:279-317) to bound the output."""
    n = s.size
    if n < 7:
        return 0
    if int.from_bytes(s[:4].tobytes(), "little") == 0x184C2102:   # legacy: 8 MiB blocks
        si, nb = 4, 0
        while si + 4 <= n:
            w = int.from_bytes(s[si:si + 4].tobytes(), "little")
            si += 4
            if w != 0x184C2102:
                si += w & 0x7FFFFFFF
                nb += 1
        return nb << 23
    flg = int(s[4])
    bd = int(s[5])
    si = 6 + (8 if flg >> 3 & 1 else 0) + 1
    idx = bd >> 4 & 7
    bs = 1 << (8 + 2 * idx) if idx >= 4 else 0
    bsum = flg >> 4 & 1
    total = 0
    while si + 4 <= n:
        w = int.from_bytes(s[si:si + 4].tobytes(), "little")
        si += 4
        if w == 0:
            break
        size = w & 0x7FFFFFFF
        si += size + (4 if bsum else 0)
        # a stored block keeps its size, a compressed one grows at most 255-fold
        total += min(bs, size if w >> 31 else size * 255)
    return total


class Writer:
    """lz4.Writer (writer.go:20-242): buffers Write()s, emits one frame at Close()."""

    def __init__(self, w):
        self._w = w
        self.block_size = 1 << 22       # DefaultBlockSizeOption: options.go:29
        self.block_checksum = False
        self.content_checksum = True    # DefaultChecksumOption: options.go:30
        self.content_size = 0
        self.level = Fast
        self.concurrency = 1
        self.legacy = False
        self.handler = _on_block_done
        self._buf = bytearray()
        self._readfrom = False
        self._closed = False

    def Apply(self, *options):
        for o in options:
            o(self)

    def Write(self, buf):
        self._buf += bytes(buf)
        return len(buf)

    def ReadFrom(self, r):
        """io.Copy(zw, r) path: writer.go:199-242 (trailing empty block quirk)."""
        data = r.read()
        self._buf += data
        self._readfrom = True
        return len(data)

    def Close(self):
        if self._closed:
            return
        self._closed = True
        lens = []
        self._w.write(compress_frame(self._buf, self.block_size, self.block_checksum,
                                     self.content_checksum, self.content_size, self.level,
                                     self._readfrom, self.legacy, lens))
        bs = (8 << 20) if self.legacy else self.block_size
        for i, size in enumerate(lens):
            self.handler(size)                                           # writer.go:127
            if self._readfrom:                                           # writer.go:234
                self.handler(max(0, min(bs, len(self._buf) - i * bs)))

    def Reset(self, w):
        self._w = w
        self._buf = bytearray()
        self._readfrom = False
        self._closed = False


class Reader:
    """lz4.Reader (reader.go:19-275) over one frame held in memory."""

    def __init__(self, r):
        self._r = r
        self._out = None
        self._pos = 0
        self.concurrency = 1
        self.handler = _on_block_done
        self._ends = []     # end offset of every block in the decoded stream

    def Apply(self, *options):
        for o in options:
            o(self)

    def _fill(self):
        if self._out is None:
            lens = []
            self._out = uncompress_frame(self._r.read(), block_lens=lens)
            self._ends, at = [], 0
            for k in lens:
                at += k
                self._ends.append(at)

    def _report(self, lo, hi):
        """handler(bn) for every block portion in [lo, hi) of the stream: reader.go:154,255."""
        if self.handler is _on_block_done:
            return
        import bisect
        i = bisect.bisect_right(self._ends, lo)
        while lo < hi and i < len(self._ends):
            cut = min(hi, self._ends[i])
            self.handler(cut - lo)
            lo = cut
            i += 1

    def Read(self, n=-1):
        self._fill()
        if n < 0:
            n = len(self._out) - self._pos
        b = self._out[self._pos:self._pos + n]
        self._report(self._pos, self._pos + len(b))
        self._pos += len(b)
        return b

    read = Read

    def WriteTo(self, w):
        self._fill()
        self._report(self._pos, len(self._out))
        w.write(self._out[self._pos:])
        k = len(self._out) - self._pos
        self._pos = len(self._out)
        return k

    def Reset(self, r):
        self._r = r
        self._out = None
        self._pos = 0
        self._ends = []


class CompressingReader:
    """lz4.CompressingReader (compressing_reader.go:21-174): Read() yields the LZ4 frame
    of everything the source holds -- the opposite of Reader.  The source is drained and
    compressed as one batch on the first Read; the frame bytes are those of
    Writer.Write(all) + Close() (init / Compress().Write / CloseW: :76-81,116-157)."""

    def __init__(self, src):
        self.block_size = 1 << 22       # DefaultBlockSizeOption
        self.block_checksum = False
        self.content_checksum = True    # DefaultChecksumOption
        self.content_size = 0
        self.level = Fast
        self.handler = _on_block_done
        self.Reset(src)

    def Source(self):
        """compressing_reader.go:46-49"""
        return self._src

    def Close(self):
        """compressing_reader.go:51-56: closes the source"""
        close = getattr(self._src, "close", None)
        if close is not None:
            close()

    close = Close

    def Apply(self, *options):
        """compressing_reader.go:58-72: only before the first Read"""
        if self._out is not None:
            raise ErrOptionClosedOrError()
        for o in options:
            o(self)

    def Read(self, n=-1):
        """compressing_reader.go:83-165; b"" is io.EOF"""
        if self._out is None:
            lens = []
            self._out = compress_frame(self._src.read(), self.block_size, self.block_checksum,
                                       self.content_checksum, self.content_size, self.level,
                                       block_lens=lens)
            for size in lens:
                self.handler(size)      # :126,142
        if n < 0:
            n = len(self._out) - self._pos
        b = self._out[self._pos:self._pos + n]
        self._pos += len(b)
        return b

    read = Read

    def Reset(self, src):
        """compressing_reader.go:167-174: keeps the options"""
        self._src = src
        self._out = None
        self._pos = 0
