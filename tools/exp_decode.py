import sys, time, torch, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lz4_amd import corpus, device as dev
dv = torch.device('cuda', 0)
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
only = sys.argv[2] if len(sys.argv) > 2 else None
for kind, BS in (('text', 1 << 16), ('mixed', 1 << 18), ('binary', 1 << 18), ('random', 1 << 18)):
    if only and kind != only:
        continue
    n = int(gib * (1 << 30))
    src = corpus.generate_device(kind, n, dv)
    nb = n // BS
    comp, res = dev.compress_blocks(src, BS, 0)
    off, sw = dev.frame_layout(res, n, BS, nb)
    packed = torch.empty(int(off[nb].item()) + 16, dtype=torch.uint8, device=dv)
    dev.frame_scatter(src, BS, comp, res, off, nb, packed)
    del comp
    so = (off[:nb] + 4).contiguous()
    out = torch.empty((nb, BS), dtype=torch.uint8, device=dv); ol = torch.empty(nb, dtype=torch.int32, device=dv)
    ts = []
    for it in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        dev.uncompress_blocks(packed, so, sw, BS, out=out, out_len=ol)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    print(kind, BS, 'decode GB/s %.1f' % (n / min(ts) / 1e9), 'exact', torch.equal(out.view(-1), src), bool((ol == BS).all()), flush=True)
    del src, out, packed
