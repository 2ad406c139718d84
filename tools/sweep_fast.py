import sys, time, torch
sys.path.insert(0, '.')
from lz4_amd import corpus, device as dev
kind = sys.argv[1] if len(sys.argv) > 1 else 'text'
gib = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
BS = int(sys.argv[3]) if len(sys.argv) > 3 else (1 << 18)
dv = torch.device('cuda', 0)
n = int(gib * (1 << 30))
src = corpus.generate_device(kind, n, dv)
nb = n // BS
out = torch.empty((nb, BS), dtype=torch.uint8, device=dv); res = torch.zeros(nb, dtype=torch.int32, device=dv)
for wpc in [int(a) for a in (sys.argv[4].split(',') if len(sys.argv) > 4 else '4,5,6,8,10,12,16,20,24'.split(','))]:
    ws = torch.empty(256 * wpc * 131072 + 256, dtype=torch.uint8, device=dv)
    ts = []
    for it in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        dev.compress_blocks(src, BS, 0, out=out, result=res, workspace=ws)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    print(kind, BS, 'waves/CU', wpc, 'GB/s %.2f' % (n / min(ts) / 1e9), flush=True)
    del ws
