import sys, time, torch
sys.path.insert(0, '.')
from lz4_amd import corpus, device as dev
kind = sys.argv[1] if len(sys.argv) > 1 else 'text'
wpc = int(sys.argv[2]) if len(sys.argv) > 2 else 16
gib = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
dv = torch.device('cuda', 0)
n = int(gib * (1 << 30)); BS = 1 << 18
src = corpus.generate_device(kind, n, dv)
nb = n // BS
out = torch.empty((nb, BS), dtype=torch.uint8, device=dv); res = torch.zeros(nb, dtype=torch.int32, device=dv)
ws = torch.empty(256 * wpc * 131072, dtype=torch.uint8, device=dv) if wpc else None
for it in range(2):
    torch.cuda.synchronize(); t = time.perf_counter()
    dev.compress_blocks(src, BS, 0, out=out, result=res, workspace=ws)
    torch.cuda.synchronize(); print(kind, wpc, 'GB/s %.2f' % (n / (time.perf_counter() - t) / 1e9), flush=True)
