"""One-off wide fuzz of the batched Fast compressor (global-table kernel from 513 blocks up, team
kernel below) against the oracle:
    python tools/fuzz_fast.py [SEED]
Blocks of adversarial synthetic data (runs, short periods, tiny alphabets, sparse noise on
text, sorted bytes) at 64 KiB and 256 KiB block sizes, frame and bound dst modes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lz4_amd import corpus, device as dev, _lib
from oracle import pyoracle as orc

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
text = corpus.generate("text", 0, 9 << 20)

def piece(n):
    k = int(rng.integers(0, 8))
    if k == 0:
        return np.full(n, int(rng.integers(0, 256)), dtype=np.uint8)
    if k == 1:
        p = rng.integers(0, 256, int(rng.integers(1, 300)), dtype=np.uint8)
        return np.tile(p, n // p.size + 1)[:n]
    if k == 2:
        return rng.integers(0, int(rng.integers(2, 6)), n, dtype=np.uint8)
    if k == 3:
        a = text[int(rng.integers(0, text.size - n)):][:n].copy()
        m = rng.random(n) < rng.choice([0.001, 0.01, 0.1])
        a[m] = rng.integers(0, 256, int(m.sum()), dtype=np.uint8)
        return a
    if k == 4:
        return np.sort(rng.integers(0, 256, n, dtype=np.uint8))
    if k == 5:
        return rng.integers(0, 256, n, dtype=np.uint8)
    if k == 6:  # long runs of different lengths
        out = np.empty(n, dtype=np.uint8); i = 0
        while i < n:
            l = int(rng.integers(1, 5000)); out[i:i + l] = int(rng.integers(0, 4)); i += l
        return out
    a = text[int(rng.integers(0, text.size - n)):][:n].copy()  # text with a repeated far segment
    j = int(rng.integers(0, max(n - 3000, 1))); a[-2000:] = a[j:j + 2000] if n > 5000 else a[-2000:]
    return a

for bs, nb in ((1 << 16, 700), (1 << 18, 600), (1 << 16, 400), (1 << 18, 200), (1 << 20, 48), (1 << 22, 12)):
    parts = []
    for b in range(nb):
        cuts = sorted(rng.integers(1, bs, int(rng.integers(0, 3))).tolist()) + [bs]
        prev = 0
        for c in cuts:
            if c > prev:
                parts.append(piece(c - prev)); prev = c
    data = np.concatenate(parts)[:nb * bs - int(rng.integers(0, bs // 2))]
    src = torch.zeros(data.size + 16, dtype=torch.uint8, device="cuda")
    src[:data.size] = torch.from_numpy(data).cuda()
    src = src[:data.size]
    for mode in (_lib.DST_FRAME, _lib.DST_BOUND):
        comp, res = dev.compress_blocks(src, bs, 0, dst_mode=mode)
        res_h, comp_h = res.cpu().numpy(), comp.cpu().numpy()
        n_blocks = res_h.shape[0]
        for b in range(n_blocks):
            blk = data[b * bs:(b + 1) * bs]
            dl = blk.size if mode == _lib.DST_FRAME else orc.CompressBlockBound(blk.size)
            try:
                k, pl = orc.CompressBlock(blk, dl)
            except Exception:
                k, pl = -1, b""
            r = int(res_h[b])
            assert r == k, (bs, mode, b, r, k)
            if k > 0:
                assert comp_h[b, :k].tobytes() == pl, (bs, mode, b)
        print("bs", bs, "mode", mode, "ok:", n_blocks, "blocks", flush=True)
print("fuzz ok")
