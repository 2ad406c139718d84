import sys, time, torch, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lz4_amd import corpus, device as dev
from oracle import pyoracle as o
dv = torch.device('cuda', 0)
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
kind = sys.argv[3] if len(sys.argv) > 3 else 'binary'
BS = 1 << 20
n = mib << 20
src = corpus.generate_device(kind, n, dv)
nb = n // BS
torch.cuda.synchronize(); t = time.perf_counter()
comp, res = dev.compress_blocks(src, BS, depth)
torch.cuda.synchronize(); dt = time.perf_counter() - t
r = res.cpu().numpy()
print(kind, 'HC depth', depth, 'GPU %.3f GB/s' % (n / dt / 1e9), 'ratio %.3f' % (np.where(r > 0, r, BS).sum() / n), flush=True)
m = min(n, 64 << 20)
h = src[:m].cpu().numpy()
t = time.perf_counter(); pl, sw, _ = o.compress_blocks(h, BS, depth, 16, False); dt = time.perf_counter() - t
print('CPU oracle 16 threads %.3f GB/s' % (m / dt / 1e9))
ok = all((r[b] == (0 if sw[b] >> 31 else int(sw[b]))) for b in range(len(sw)))
ok = ok and all(comp[b, :r[b]].cpu().numpy().tobytes() == pl[b, :r[b]].tobytes() for b in range(0, len(sw), 7) if r[b] > 0)
print('byte-identical sample', ok)
