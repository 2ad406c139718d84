"""One-off wide fuzz of the batched HC compressor against the oracle:
    python tools/fuzz_hc.py [SEED]
Adversarial synthetic blocks (runs, short periods, tiny alphabets, noisy text, sorted bytes, long
runs of a few values, a repeated far segment, dense chains with jumped members) -- the long chains,
the gate behind a held match, the run insertion, the miss front and the dense-chain scan -- at 64 KiB / 256 KiB / 1 MiB blocks and depths 1 .. 131072, frame dst mode
(dst = len(src): overflow and "not compressible" outcomes included)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lz4_amd import corpus, device as dev
from oracle import pyoracle as orc

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
text = corpus.generate("text", 0, 1 << 22)
binary = corpus.generate("binary", 0, 1 << 22)


def piece(n, deep):
    k = int(rng.integers(0, 11))
    if k >= 9:  # dense chains with members that the parse jumps (the scan and its position map, hc.hip)
        out, size, fill = [], 0, int(rng.integers(0, 3)) * 0x55
        while size < n:
            j = int(rng.integers(0, 4))
            if j == 0:
                a = np.full(int(rng.integers(4, 700)), fill, np.uint8)
            elif j == 1:
                a = rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8)
            elif j == 2:
                a = rng.integers(0, 256, int(rng.choice([300, 2000, 9000, 30000])), dtype=np.uint8)
                for q in rng.integers(0, a.size - 8, a.size // int(rng.choice([40, 150, 600])) + 1):
                    a[q:q + int(rng.integers(4, 9))] = fill
            else:
                m = int(rng.integers(4, 200))
                rec = np.full((m, 16), fill, np.uint8)
                rec[:, 0] = np.arange(m) & 0xFF
                a = rec.reshape(-1)
            out.append(a); size += a.size
        return np.concatenate(out)[:n]
    if k == 2 and deep:   # a tiny alphabet makes 64 Ki-candidate chains at every position:
        k = 3             # hours on the CPU side at full depth
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
    if k == 6:
        out = np.empty(n, dtype=np.uint8); i = 0
        while i < n:
            l = int(rng.integers(1, 5000)); out[i:i + l] = int(rng.integers(0, 4)); i += l
        return out
    if k == 7:
        return binary[int(rng.integers(0, binary.size - n)):][:n].copy()
    a = text[int(rng.integers(0, text.size - n)):][:n].copy()
    j = int(rng.integers(0, max(n - 3000, 1)))
    if n > 5000:
        a[-2000:] = a[j:j + 2000]
    return a


threads = min(len(os.sched_getaffinity(0)), 16)
for bs, nb, depth in ((1 << 16, 300, 131072), (1 << 16, 300, 16), (1 << 18, 96, 512), (1 << 18, 64, 1),
                      (1 << 20, 24, 131072)):
    parts = []
    for b in range(nb):
        cuts = sorted(rng.integers(1, bs, int(rng.integers(0, 3))).tolist()) + [bs]
        prev = 0
        for c in cuts:
            if c > prev:
                parts.append(piece(c - prev, depth > 512)); prev = c
    data = np.concatenate(parts)[:nb * bs - int(rng.integers(0, bs // 2))]
    src = torch.from_numpy(data).cuda()
    comp, res = dev.compress_blocks(src, bs, depth)
    res_h, comp_h = res.cpu().numpy(), comp.cpu().numpy()
    t = time.time()
    pl, sw, _ = orc.compress_blocks(data, bs, depth, threads, False)
    bad = 0
    for b in range(res_h.shape[0]):
        raw = int(sw[b]) >> 31
        k = 0 if raw else int(sw[b])
        r = int(res_h[b])
        assert (r if r > 0 else 0) == k, (bs, depth, b, r, k)
        if k > 0:
            assert comp_h[b, :k].tobytes() == pl[b, :k].tobytes(), (bs, depth, b)
    print("bs", bs, "depth", depth, "ok:", res_h.shape[0], "blocks, oracle %.1f s" % (time.time() - t), flush=True)
print("fuzz ok")
