"""cfg5's per-GPU share on one MI355X, stage by stage: python tools/cfg5_share.py [GIB]
16 GiB - 4 MiB + 12,345 B of the mixed corpus, 4 MiB blocks, block + content checksums,
frame assembled on the device (the stages of lz4_amd/device.py: frame_compress), then the
frame decoded again (blocks + block checksums + content checksum)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lz4_amd import _lib, corpus, device as dev

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 16.0
BS = 1 << 22
n = int(gib * (1 << 30)) - BS + 12345
dv = torch.device("cuda", 0)
src = torch.zeros(n + 16, dtype=torch.uint8, device=dv)
ch = 1 << 24
bases = corpus.device_bases(torch, dv)
for a in range(0, n, ch):
    c = min(ch, n - a)
    src[a:a + c] = corpus.generate("mixed", a, c, xp=torch, device=dv, bases=bases)
src = src[:n]
nb = dev.n_blocks(n, BS)

def timed(name, fn, nbytes=n):
    torch.cuda.synchronize(); t = time.perf_counter()
    r = fn()
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("%-34s %8.1f ms  %8.2f GB/s" % (name, dt * 1e3, nbytes / dt / 1e9), flush=True)
    return r

for rep in range(2):
    print("-- pass", rep, "(%d blocks of 4 MiB, %.2f GiB)" % (nb, n / 2**30))
    comp, res = timed("Fast compress (k_fast_gtab)", lambda: dev.compress_blocks(src, BS, 0))
    xxh = timed("block checksums (k_xxh32_blocks)", lambda: dev.xxh32_blocks(src, BS, nb, n=n))
    cs = timed("content checksum (k_xxh32_stream)", lambda: dev.xxh32_stream(src, n))
    def assemble():
        off, sw = dev.frame_layout(res, n, BS, nb, xxh, 7)
        end = int(off[nb].item())
        frame = torch.empty(end + 8, dtype=torch.uint8, device=dv)
        dev.frame_scatter(src, BS, comp, res, off, nb, frame, 0, xxh, n=n)
        return off, sw, frame, end
    off, sw, frame, end = timed("layout scan + scatter", assemble)
    print("frame bytes %d (ratio %.4f)" % (end + 8, (end + 8) / n))
    src_off = (off[:nb] + 4).contiguous()
    out, ol = timed("decode blocks (k_decode)", lambda: dev.uncompress_blocks(frame, src_off, sw, BS))
    ok = bool((ol[:-1] == BS).all().item()) and int(ol[-1].item()) == n - (nb - 1) * BS
    ok = ok and torch.equal(out.view(-1)[:n], src)
    print("round trip exact:", ok)
    del comp, frame, out
    torch.cuda.empty_cache()
