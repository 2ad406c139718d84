"""Fast compress with few blocks: LDS-table kernel vs global-table kernel, by block count."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lz4_amd import corpus, device as dev
dv = torch.device('cuda', 0)
kind = sys.argv[1] if len(sys.argv) > 1 else 'text'
src_all = corpus.generate_device(kind, 4 << 30, dv)
for bs in (1 << 22, 1 << 18):
    for nb in (64, 128, 256, 384, 512, 768, 1024, 2048):
        n = nb * bs
        if n > src_all.numel():
            continue
        src = src_all[:n]
        row = []
        for mode in ('lds', 'gtab'):
            os.environ['LZ4HIP_FAST_GLOBAL_MIN'] = '1000000' if mode == 'lds' else '1'
            dev._ws_cache.clear()
            dev.compress_blocks(src, bs, 0); torch.cuda.synchronize()
            t = time.perf_counter(); dev.compress_blocks(src, bs, 0); torch.cuda.synchronize(); dt = time.perf_counter() - t
            row.append(n / dt / 1e9)
        print('%s bs %8d nb %5d  lds %6.2f GB/s  gtab %6.2f GB/s' % (kind, bs, nb, row[0], row[1]), flush=True)
