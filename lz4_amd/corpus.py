"""Seeded synthetic corpora of the benchmark configs.

Every byte is a pure function of (corpus kind, seed, global byte index), built
from small base tiles (the reference's own test files, committed under
tests/golden/testdata) plus a seeded substitution, so that
  * the GPU can materialise 8 GiB in HBM without shipping a file, and
  * the CPU oracle can regenerate any sampled block bit-for-bit.
The same code runs on numpy (xp=numpy) and on torch CUDA tensors (xp=torch).

kinds:
  "text"   cfg2: pg1661 + Tom Sawyer + gettysburg tiled, 1-in-64 printable substitution
  "binary" cfg4: bzImage_lz4_isolated + pg_control.tar tiled, 1-in-256 byte substitution
  "random" PRNG bytes (random.data-style, incompressible => raw blocks)
  "mixed"  cfg3: per 256 KiB region r: r%4 in {0,1} text, 2 binary, 3 random
"""
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
TESTDATA = os.path.join(os.path.dirname(_HERE), "tests", "golden", "testdata")

SEED_TEXT, SEED_MIXED, SEED_BINARY = 0x5EED0002, 0x5EED0003, 0x5EED0004
MIX_REGION = 1 << 18

_TILES = {
    "text": ["pg1661.txt", "Mark.Twain-Tom.Sawyer.txt", "gettysburg.txt"],
    "binary": ["bzImage_lz4_isolated", "pg_control.tar"],
}
_base_cache = {}


def base_tile(kind):
    """numpy uint8 base tile of a kind (host)."""
    if kind not in _base_cache:
        parts = []
        for name in _TILES[kind]:
            with open(os.path.join(TESTDATA, name), "rb") as f:
                parts.append(np.frombuffer(f.read(), dtype=np.uint8))
        _base_cache[kind] = np.concatenate(parts)
    return _base_cache[kind]


def _mix(xp, p, seed):
    """32-bit avalanche of a 64-bit index; int64 arithmetic that never overflows."""
    m = 0xFFFFFFFF
    x = ((p & m) ^ ((p >> 32) * 0x9E37) ^ seed) & m
    x = ((x ^ (x >> 16)) * 0x45D9F3B) & m
    x = ((x ^ (x >> 16)) * 0x45D9F3B) & m
    return x ^ (x >> 16)


def _gen_kind(xp, kind, seed, p, bases):
    h = _mix(xp, p, seed)
    if kind == "random":
        return h & 0xFF
    base = bases[kind]
    v = base[p % base.shape[0]]
    if xp is not np:
        v = v.to(p.dtype)
    else:
        v = v.astype(np.int64)
    if kind == "text":
        return xp.where((h & 63) == 0, 32 + ((h >> 8) % 95), v)
    return xp.where((h & 255) == 0, (h >> 8) & 0xFF, v)


def generate(kind, start, count, seed=None, xp=np, device=None, bases=None):
    """Bytes [start, start+count) of the corpus as a uint8 array/tensor."""
    if seed is None:
        seed = {"text": SEED_TEXT, "binary": SEED_BINARY, "mixed": SEED_MIXED,
                "random": SEED_MIXED}[kind]
    if xp is np:
        p = np.arange(start, start + count, dtype=np.int64)
        if bases is None:
            bases = {k: base_tile(k) for k in _TILES}
    else:
        p = xp.arange(start, start + count, dtype=xp.int64, device=device)
        if bases is None:
            bases = device_bases(xp, device)
    if kind != "mixed":
        out = _gen_kind(xp, kind, seed, p, bases)
    else:
        r = (p // MIX_REGION) % 4
        t = _gen_kind(xp, "text", seed, p, bases)
        b = _gen_kind(xp, "binary", seed, p, bases)
        z = _gen_kind(xp, "random", seed, p, bases)
        out = xp.where(r < 2, t, xp.where(r == 2, b, z))
    return out.astype(np.uint8) if xp is np else out.to(xp.uint8)


def device_bases(torch, device):
    return {k: torch.from_numpy(base_tile(k).copy()).to(device) for k in _TILES}


def generate_device(kind, n, device, seed=None, chunk=1 << 24):
    """Materialise n corpus bytes in HBM (torch CUDA uint8 tensor), chunk by chunk."""
    import torch
    out = torch.empty(n, dtype=torch.uint8, device=device)
    bases = device_bases(torch, device)
    for a in range(0, n, chunk):
        c = min(chunk, n - a)
        out[a:a + c] = generate(kind, a, c, seed, xp=torch, device=device, bases=bases)
    return out
