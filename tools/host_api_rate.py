"""PCIe-inclusive rates of the host-pointer entry points (what the cgo shim binds):
lz4hip_frame_compress / lz4hip_frame_uncompress called through ctypes on preallocated,
pre-touched pageable host buffers (no Python copies in the timed region)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lz4_amd import _lib, corpus
lib = _lib.load()
lib.lz4hip_init(0)  # freed device memory stays in the pool between calls, as in a long-lived Writer
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
n = int(gib * (1 << 30))
data = corpus.generate("mixed", 0, n)
for bs in (1 << 22, 1 << 18):
    for flags in (0, _lib.F_CONTENT_CHECKSUM, _lib.F_CONTENT_CHECKSUM | _lib.F_BLOCK_CHECKSUM):
        cap = lib.lz4hip_frame_bound(n, bs, flags)
        f = np.zeros(cap, dtype=np.uint8); back = np.zeros(n + 16, dtype=np.uint8)
        lib.lz4hip_frame_compress(data.ctypes.data, n, bs, flags, 0, 0, f.ctypes.data, cap)  # warm up: pool and staging buffers sized
        lib.lz4hip_frame_uncompress(f.ctypes.data, lib.lz4hip_frame_compress(data.ctypes.data, n, bs, flags, 0, 0, f.ctypes.data, cap), back.ctypes.data, back.size)
        t = time.perf_counter(); k = lib.lz4hip_frame_compress(data.ctypes.data, n, bs, flags, 0, 0, f.ctypes.data, cap); tc = time.perf_counter() - t
        t = time.perf_counter(); m = lib.lz4hip_frame_uncompress(f.ctypes.data, k, back.ctypes.data, back.size); td = time.perf_counter() - t
        ok = m == n and bool((back[:n] == data).all())
        print("block %7d flags %d  frame_compress %6.2f GB/s (%.3f s)  frame_uncompress %6.2f GB/s (%.3f s)  ratio %.3f ok %s"
              % (bs, flags, n / tc / 1e9, tc, n / td / 1e9, td, k / n, ok), flush=True)
