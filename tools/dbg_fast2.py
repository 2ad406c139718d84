import sys, torch, numpy as np
sys.path.insert(0, '.')
from lz4_amd import corpus, device as dev
from oracle import pyoracle as o
dv = torch.device('cuda', 0)
BS = 1 << 18
n = 8 << 20
data = corpus.generate('text', 0, n)
src = torch.from_numpy(data).to(dv)
pl, sw, _ = o.compress_blocks(data, BS, 0, 8, False)
nb = n // BS
for trial in range(3):
    ws = torch.zeros(64 * 131072, dtype=torch.uint8, device=dv)
    out, res = dev.compress_blocks(src, BS, 0, workspace=ws)
    r = res.cpu().numpy(); outh = out.cpu().numpy()
    bad = [b for b in range(nb) if r[b] != (0 if sw[b] >> 31 else int(sw[b])) or outh[b, :max(r[b],0)].tobytes() != pl[b, :max(r[b],0)].tobytes()]
    print('trial', trial, 'bad blocks', len(bad), bad[:8])
    if bad:
        b = bad[0]
        want = pl[b, :int(sw[b] & 0x7fffffff)]; got = outh[b, :max(r[b], 0)]
        m = min(len(want), len(got)); diff = np.nonzero(want[:m] != got[:m])[0]
        print(' block', b, 'res', r[b], 'want', int(sw[b]), 'first diff at', diff[:3], 'of', m)
        print(' want', want[diff[0]-8:diff[0]+8].tolist()); print(' got ', got[diff[0]-8:diff[0]+8].tolist())
        try:
            dec = o.UncompressBlock(got.tobytes(), BS)
            print(' got decodes ok:', dec == data[b*BS:(b+1)*BS].tobytes())
        except Exception as e:
            print(' got decode error', e)
        # walk sequences of `want` to locate the one containing the diff
        w = want.tobytes(); si = 0; di_ = 0
        while si < len(w):
            t0 = si; tok = w[si]; si += 1; ll = tok >> 4
            if ll == 15:
                while True:
                    bb = w[si]; si += 1; ll += bb
                    if bb != 255: break
            si += ll; di_ += ll
            if si >= len(w): break
            off = w[si] | (w[si+1] << 8); si += 2; ml = (tok & 15) + 4
            if ml == 19:
                while True:
                    bb = w[si]; si += 1; ml += bb
                    if bb != 255: break
            if t0 <= diff[0] < si:
                print(' seq at comp', t0, 'out pos', di_, 'lit', ll, 'off', off, 'ml', ml, 'diff rel', diff[0]-t0)
                break
            di_ += ml
        break
