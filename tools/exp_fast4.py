"""A/B of the Fast kernel generations: LZ4HIP_FAST=3|4 python tools/exp_fast4.py kind gib bs_log"""
import sys, time, torch, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lz4_amd import corpus, device as dev, _lib
from oracle import pyoracle as orc
kind = sys.argv[1] if len(sys.argv) > 1 else 'mixed'
gib = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
BS = 1 << (int(sys.argv[3]) if len(sys.argv) > 3 else 18)
lib = _lib.load()
dv = torch.device('cuda', 0)
n = int(gib * (1 << 30))
src = corpus.generate_device(kind, n, dv)
nb = n // BS
out = torch.empty((nb, BS), dtype=torch.uint8, device=dv); res = torch.zeros(nb, dtype=torch.int32, device=dv)
wsb = lib.lz4hip_fast_workspace_bytes(nb)
ws = torch.empty(wsb, dtype=torch.uint8, device=dv)
ts = []
for it in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    dev.compress_blocks(src, BS, 0, out=out, result=res, workspace=ws)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    print(kind, 'bs', BS, 'ws MiB', wsb >> 20, 'GB/s %.2f' % (n / ts[-1] / 1e9), flush=True)
import os
if os.environ.get('NOPARITY'): sys.exit(0)
rng = np.random.default_rng(3)
ok = True
res_h = res.cpu().numpy()
for b in sorted(set(rng.integers(0, nb, 12).tolist()) | {0, nb - 1}):
    blk = src[b * BS:(b + 1) * BS].cpu().numpy()
    try:
        k, pl = orc.CompressBlock(blk, BS)
    except Exception:
        k, pl = 0, b''
    r = int(res_h[b])
    good = max(r, 0) == k and out[b, :k].cpu().numpy().tobytes() == pl
    if not good:
        print('MISMATCH block', b, 'got', r, 'want', k)
    ok &= good
print('parity', ok, 'ratio %.4f' % (np.where(res_h > 0, res_h, BS).sum() / n))
