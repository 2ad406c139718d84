"""Single-block host entry points called from T threads at once (the cgo + ConcurrencyOption(n) shape)."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lz4_amd as lz
from lz4_amd import corpus
bs = 4 << 20
blocks = [corpus.generate("text", i * bs, bs) for i in range(32)]
dsts = [np.zeros(lz.CompressBlockBound(bs), dtype=np.uint8) for _ in blocks]
lz.CompressBlock(blocks[0], dsts[0])
for T in (1, 4, 16, 32):
    def work(t):
        for j in range(t, len(blocks), T):
            lz.CompressBlock(blocks[j], dsts[j])
    ts = [threading.Thread(target=work, args=(t,)) for t in range(T)]
    t0 = time.perf_counter()
    for t in ts: t.start()
    for t in ts: t.join()
    dt = time.perf_counter() - t0
    print("threads %2d: %d x 4 MiB CompressBlock in %.3f s = %.2f GB/s" % (T, len(blocks), dt, len(blocks) * bs / dt / 1e9), flush=True)
