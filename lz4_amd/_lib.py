"""ctypes loader of liblz4hip.so (the C ABI declared in include/lz4hip.h).

There is no CPU fallback: if the library is missing, or no GPU is visible when a
compute entry point is called, the call raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# LZ4HIP_LIB overrides the path (kernel experiments); default is the in-tree build.
LIB_PATH = os.environ.get("LZ4HIP_LIB") or os.path.join(_HERE, "liblz4hip.so")

# status codes of include/lz4hip.h
OK, ERR_ARG, ERR_HIP, ERR_SHORT, ERR_FRAME = 0, -1, -2, -3, -4
ERR_HEADER_CHECKSUM, ERR_BLOCK_SIZE, ERR_BLOCK_CHECKSUM, ERR_FRAME_CHECKSUM, ERR_BLOCK = -5, -6, -7, -8, -9
ERR_BADARG = -10  # single-block host calls: bad argument (-1 is the codec's verdict there)

DST_FRAME, DST_BOUND = 0, 1
F_BLOCK_CHECKSUM, F_CONTENT_CHECKSUM, F_CONTENT_SIZE, F_READFROM, F_LEGACY, F_LINKED = 1, 2, 4, 8, 16, 32

# every symbol include/lz4hip.h declares: (restype, argtypes)
_vp, _sz, _u32, _u64, _i64, _int = C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint64, C.c_int64, C.c_int
SYMBOLS = {
    "lz4hip_last_error": (C.c_char_p, []),
    "lz4hip_device_count": (_int, []),
    "lz4hip_version": (C.c_char_p, []),
    "lz4hip_compress_bound": (_sz, [_sz]),
    "lz4hip_fast_workspace_bytes": (_sz, [_sz]),
    "lz4hip_compress_fast_blocks": (_int, [_vp, _sz, _u32, _int, _vp, _sz, _vp, _vp, _sz, _vp]),
    "lz4hip_compress_fast_blocks_noalloc": (_int, [_vp, _sz, _u32, _int, _vp, _sz, _vp, _vp]),
    "lz4hip_hc_workspace_bytes": (_sz, [_sz]),
    "lz4hip_compress_hc_blocks": (_int, [_vp, _sz, _u32, _u32, _int, _vp, _sz, _vp, _vp, _sz, _vp]),
    "lz4hip_uncompress_blocks": (_int, [_vp, _sz, _vp, _vp, _sz, _vp, _sz, _u32, _vp, _sz, _vp, _vp]),
    "lz4hip_init": (_int, [_int]),
    "lz4hip_shutdown": (_int, []),
    "lz4hip_xxh32_checked": (_int, [_vp, _sz, _vp]),
    "lz4hip_stream_create": (_vp, [_u32, _u32, _u32, _sz]),
    "lz4hip_stream_destroy": (None, [_vp]),
    "lz4hip_stream_reset": (None, [_vp]),
    "lz4hip_stream_checksum": (_u32, [_vp]),
    "lz4hip_stream_compress_blocks": (_i64, [_vp, _vp, _sz, _vp, _sz, _vp]),
    "lz4hip_stream_uncompress_blocks": (_i64, [_vp, _vp, _sz, _vp, _sz, _vp, _vp, _vp, _vp]),
    "lz4hip_xxh32_blocks": (_int, [_vp, _sz, _sz, _vp, _vp, _sz, _vp, _vp]),
    "lz4hip_xxh32_stream": (_int, [_vp, _sz, _vp, _vp]),
    "lz4hip_xxh32_stream_update": (_int, [_vp, _sz, _vp, _int, _int, _vp, _vp]),
    "lz4hip_xxh32_stream_update_host": (_int, [_vp, _sz, _vp, _int, _int, _vp, _vp]),
    "lz4hip_xxh32_host_update": (_int, [_vp, _sz, _vp, _int, _int, _vp]),
    "lz4hip_frame_layout": (_int, [_vp, _sz, _u32, _sz, _vp, _u64, _vp, _vp, _vp]),
    "lz4hip_frame_scatter": (_int, [_vp, _sz, _u32, _vp, _sz, _vp, _vp, _vp, _sz, _vp, _u64, _vp]),
    "lz4hip_frame_index": (_int, [_vp, _sz, _u64, _u32, _int, _sz, _vp, _vp, _vp, _vp, _vp]),
    "lz4hip_pack_blocks": (_int, [_vp, _sz, _vp, _vp, _sz, _vp, _vp]),
    "lz4hip_frame_header": (_sz, [_vp, _u32, _u32, _u64]),
    "lz4hip_frame_trailer": (_sz, [_vp, _u32, _u32]),
    "lz4hip_frame_parse_header": (_int, [_vp, _sz, _vp, _vp, _vp, _vp]),
    "lz4hip_frame_bound": (_sz, [_sz, _u32, _u32]),
    "lz4hip_compress_block": (_i64, [_vp, _sz, _vp, _sz]),
    "lz4hip_compress_block_hc": (_i64, [_vp, _sz, _vp, _sz, _u32]),
    "lz4hip_uncompress_block": (_i64, [_vp, _sz, _vp, _sz, _vp, _sz]),
    "lz4hip_xxh32": (_u32, [_vp, _sz]),
    "lz4hip_frame_compress": (_i64, [_vp, _sz, _u32, _u32, _u32, _u64, _vp, _sz]),
    "lz4hip_frame_compress_lens": (_i64, [_vp, _sz, _u32, _u32, _u32, _u64, _vp, _sz, _vp, _sz, _vp]),
    "lz4hip_frame_uncompress": (_i64, [_vp, _sz, _vp, _sz]),
    "lz4hip_frame_uncompress_lens": (_i64, [_vp, _sz, _vp, _sz, _vp, _sz, _vp]),
}

_lib = None


class Lz4HipError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("lz4hip error %d: %s" % (code, text))
        self.code = code


def load():
    """Load liblz4hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "liblz4hip.so is not built: run `make -C lz4_amd/csrc` "
                "(or __graft_entry__.build()); there is no CPU fallback")
        # One HIP runtime per process: when torch is present its bundled
        # libamdhip64 must be the one loaded first, or torch later finds no GPU.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def last_error():
    return load().lz4hip_last_error().decode()


def check(rc):
    if rc < 0:
        raise Lz4HipError(rc, last_error())
    return rc


def require_gpu():
    if load().lz4hip_device_count() < 1:
        raise RuntimeError("lz4_amd: no HIP device visible; the codec has no CPU path")
