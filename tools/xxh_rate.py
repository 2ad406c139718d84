"""Rate of the serial content checksum (k_xxh32_stream) and of the few-long-spans block
checksum (k_xxh32_spans_wave) with the data resident in HBM; checks against python-xxhash
when it is importable.   python tools/xxh_rate.py [MiB]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from lz4_amd import device as dv

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = mib << 20
g = torch.Generator(device="cuda").manual_seed(7)
src = torch.randint(0, 256, (n + 37,), dtype=torch.uint8, device="cuda", generator=g)
for size in (n + 37, n, 1 << 20, 65536 + 5):
    dv.xxh32_stream(src, size)
    torch.cuda.synchronize()
    t = time.perf_counter()
    out = dv.xxh32_stream(src, size)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    got = int(out.item()) & 0xFFFFFFFF
    try:
        import xxhash
        ok = got == xxhash.xxh32(src[:size].cpu().numpy().tobytes(), seed=0).intdigest()
    except ImportError:
        ok = None
    print("stream %11d B: %8.3f ms  %6.2f GB/s  check=%s" % (size, dt * 1e3, size / dt / 1e9, ok))
for bs, nb in ((4 << 20, 16), (1 << 20, 64), (1 << 16, 1024)):
    dv.xxh32_blocks(src, bs, nb, nb * bs)
    torch.cuda.synchronize()
    t = time.perf_counter()
    dv.xxh32_blocks(src, bs, nb, nb * bs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print("spans  %4d x %7d B: %8.3f ms  %6.2f GB/s" % (nb, bs, dt * 1e3, nb * bs / dt / 1e9))
