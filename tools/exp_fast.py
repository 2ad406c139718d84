import sys, time, torch, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lz4_amd import corpus, device as dev, _lib
lib = _lib.load()
n = int(float(os.environ.get('GIB', '2')) * (1 << 30))
BS = 1 << 18
dv = torch.device('cuda', 0)
kinds = sys.argv[1].split(',') if len(sys.argv) > 1 else ['mixed', 'text', 'random']
for kind in kinds:
    src = corpus.generate_device(kind, n, dv)
    nb = n // BS
    out = torch.empty((nb, BS), dtype=torch.uint8, device=dv); res = torch.zeros(nb, dtype=torch.int32, device=dv)
    ref_out, ref_res = dev.compress_blocks(src, BS, 0)
    ref_res = ref_res.clone()
    for waves_per_cu in (4, 6, 8, 10, 12, 16, 20):
        slots = 256 * waves_per_cu
        ws = torch.empty(slots * 131072, dtype=torch.uint8, device=dv) if slots else None
        ts = []
        for it in range(3):
            torch.cuda.synchronize(); t = time.perf_counter()
            dev.compress_blocks(src, BS, 0, out=out, result=res, workspace=ws)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        ok = torch.equal(res, ref_res)
        print(kind, 'waves/CU', waves_per_cu, 'GB/s %.2f' % (n / min(ts) / 1e9), 'same', ok, flush=True)
