"""One-off wide fuzz of the batched decoder against the oracle (not part of the suite):
    python tools/fuzz_decode.py [N] [SEED]
Valid blocks (Fast-compressed corpus pieces and hand-built dense-token streams) with random
mutations / truncations, three dst sizes, guard bytes behind every dst."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from lz4_amd import corpus, device as dev
from oracle import pyoracle as orc
from test_gpu_batch import _random_block

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
kinds = ("text", "binary", "mixed")
blobs = []
for i in range(N):
    if i % 2:
        n = int(rng.integers(20, 9000))
        raw = corpus.generate(kinds[i % 3], int(rng.integers(0, 1 << 24)), n)
        _, c = orc.CompressBlock(raw, orc.CompressBlockBound(n))
    else:
        shape = [(0, 0, 1), (0, 3, 8), (2, 6, 40), (20, 30, 300), (3, 300, 70000), (300, 4, 5000)][int(rng.integers(0, 6))]
        c, _ = _random_block(rng, int(rng.integers(1, 400)), *shape)
    c = bytearray(c)
    for _ in range(int(rng.integers(0, 4))):
        c[int(rng.integers(0, len(c)))] = int(rng.integers(0, 256))
    if rng.integers(0, 8) == 0:
        c = c[:int(rng.integers(1, len(c) + 1))]
    blobs.append(bytes(c))
blob = b"".join(blobs)
offs = np.cumsum([0] + [len(b) for b in blobs[:-1]]).astype(np.int64)
lens = np.array([len(b) for b in blobs], dtype=np.int32)
a = np.frombuffer(blob, dtype=np.uint8)
d_blob = torch.zeros(a.size + 16, dtype=torch.uint8, device="cuda")
d_blob[:a.size] = torch.from_numpy(a.copy()).cuda()
stride = 16384
bad_total = 0
for dst_len in (16000, 9000, 1500, 100):
    out = torch.full((N, stride), 0xAB, dtype=torch.uint8, device="cuda")
    o, ol = dev.uncompress_blocks(d_blob[:a.size], torch.from_numpy(offs).cuda(), torch.from_numpy(lens).cuda(),
                                  dst_len, out=out, dst_stride=stride)
    got, outs = ol.cpu().numpy(), o.cpu().numpy()
    assert (outs[:, dst_len:] == 0xAB).all(), "wrote at or beyond dst_len"
    for i, c in enumerate(blobs):
        want, wd = orc.decodeBlock(c, dst_len)
        assert (got[i] < 0) == (want < 0), (dst_len, i, got[i], want)
        if want >= 0:
            assert got[i] == want and outs[i, :want].tobytes() == wd[:want].tobytes(), (dst_len, i)
        else:
            bad_total += 1
    print("dst_len", dst_len, "ok", flush=True)
print("fuzz ok:", N, "blocks x 4 dst sizes,", bad_total, "rejected as the oracle rejects them")
