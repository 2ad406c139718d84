"""GPU parity of the single-block API (lz4.go:27-131) against the CPU oracle.

Mirrors internal/lz4block/block_test.go and writer_test.go's block cases; every
call goes through the C ABI (lz4_amd.block -> liblz4hip.so).  Bit-exact bar.
"""
import numpy as np
import pytest

from conftest import load

pytestmark = pytest.mark.gpu

RAW = ["e.txt", "gettysburg.txt", "Mark.Twain-Tom.Sawyer.txt", "pg1661.txt", "pi.txt",
       "random.data", "repeat.txt", "pg_control.tar", "upperbound.data", "issue51.data",
       "bzImage_lz4_isolated"]


def _gpu_compress(lz, src, dst_len, hc=None):
    dst = np.zeros(max(dst_len, 1), dtype=np.uint8)[:dst_len]
    try:
        n = lz.CompressBlock(src, dst) if hc is None else lz.CompressBlockHC(src, dst, hc)
    except lz.ErrInvalidSourceShortBuffer:
        return "ERR"
    return dst[:n].tobytes()


def _cpu_compress(o, src, dst_len, hc=None):
    try:
        return (o.CompressBlock(src, dst_len) if hc is None else o.CompressBlockHC(src, hc, dst_len))[1]
    except o.ErrInvalidSourceShortBuffer:
        return "ERR"


@pytest.mark.parametrize("name", RAW)
def test_fast_whole_file_byte_identical(name, oracle):
    """TestCompressUncompressBlock (block_test.go:34-115) + byte identity."""
    import lz4_amd as lz
    src = load(name)
    bound = lz.CompressBlockBound(len(src))
    assert bound == oracle.CompressBlockBound(len(src))
    for dst_len in (bound, len(src), len(src) // 2):
        got = _gpu_compress(lz, src, dst_len)
        want = _cpu_compress(oracle, src, dst_len)
        assert got == want, (name, dst_len)
    z = _gpu_compress(lz, src, bound)
    out = np.zeros(len(src), dtype=np.uint8)
    assert lz.UncompressBlock(z, out) == len(src)
    assert out.tobytes() == src


@pytest.mark.parametrize("name", ["gettysburg.txt", "repeat.txt", "pg_control.tar",
                                  "upperbound.data", "issue51.data", "random.data", "e.txt"])
@pytest.mark.parametrize("depth", [0, 1, 10, 16, 512, 131072])
def test_hc_byte_identical(name, depth, oracle):
    import lz4_amd as lz
    src = load(name)
    for dst_len in (lz.CompressBlockBound(len(src)), len(src), int(len(src) * 0.4)):
        assert _gpu_compress(lz, src, dst_len, depth) == _cpu_compress(oracle, src, dst_len, depth)


def test_upperbound_short_dst_errors():
    """TestCompressCornerCase_CopyDstUpperBound (block_test.go:117-151)."""
    import lz4_amd as lz
    src = load("upperbound.data")
    dst = np.zeros(int(len(src) * 0.40), dtype=np.uint8)
    with pytest.raises(lz.ErrInvalidSourceShortBuffer):
        lz.CompressBlock(src, dst)
    with pytest.raises(lz.ErrInvalidSourceShortBuffer):
        lz.CompressBlockHC(src, dst, 16)


def test_issue23(oracle):
    """TestIssue23 (block_test.go:153-167)."""
    import lz4_amd as lz
    dst = np.zeros(lz.CompressBlockBound(1 << 16), dtype=np.uint8)
    for j in range(1, 16):
        buf = np.zeros(1 << 16, dtype=np.uint8)
        buf[::j] = 1
        n = lz.CompressBlock(buf, dst)
        assert n <= 300
        assert dst[:n].tobytes() == oracle.CompressBlock(buf)[1]


def test_issue71(oracle):
    """TestIssue71 (writer_test.go:201-231)."""
    import lz4_amd as lz
    for s in (b"abc", b"abcdefghijklmnopq"):
        bound = lz.CompressBlockBound(len(s))
        assert lz.CompressBlock(s, np.zeros(bound - 1, dtype=np.uint8)) == 0
        d = np.zeros(bound, dtype=np.uint8)
        n = lz.CompressBlock(s, d)
        assert n > 0 and d[:n].tobytes() == oracle.CompressBlock(s)[1]


def test_issue167():
    """writer_test.go:192-199: must error, not overrun an 18-byte dst."""
    import lz4_amd as lz
    src = b"\xe300000000000000\t\x00\x00"
    dst = np.zeros(18, dtype=np.uint8)
    with pytest.raises(lz.ErrInvalidSourceShortBuffer):
        lz.UncompressBlock(src, dst)


def test_small_inputs_all_dst_sizes(oracle):
    """Every (len, dst_len) corner of T5 on tiny inputs, Fast and HC."""
    import lz4_amd as lz
    rng = np.random.default_rng(7)
    for n in (0, 1, 5, 13, 14, 15, 16, 17, 40, 200):
        for alpha in (1, 3, 256):
            src = rng.integers(0, alpha, n, dtype=np.uint8).tobytes()
            for dst_len in sorted({0, 1, 2, n // 2, n, n + 1, lz.CompressBlockBound(n) - 1,
                                   lz.CompressBlockBound(n)}):
                for hc in (None, 7):
                    assert _gpu_compress(lz, src, dst_len, hc) == _cpu_compress(oracle, src, dst_len, hc), \
                        (n, alpha, dst_len, hc)


def test_host_entry_points_from_many_threads(oracle):
    """The package-level Go functions are goroutine-safe (block.go:96,310) and cgo runs
    every call on its caller's OS thread: the host-pointer entry points must be
    re-entrant.  8 threads mix Fast, HC, decode and xxh32 calls; every result must be
    the oracle's."""
    import threading
    import lz4_amd as lz
    from lz4_amd import corpus
    rng = np.random.default_rng(11)
    jobs = []
    for i in range(24):
        kind = ("text", "binary", "mixed")[i % 3]
        n = int(rng.integers(1, 200000))
        data = corpus.generate(kind, int(rng.integers(0, 1 << 24)), n).tobytes()
        jobs.append(data)
    want = []
    for d in jobs:
        k, pl = oracle.CompressBlock(d, oracle.CompressBlockBound(len(d)))
        k2, pl2 = oracle.CompressBlockHC(d[:20000], 16, oracle.CompressBlockBound(min(len(d), 20000)))
        want.append((pl, pl2, oracle.ChecksumZero(d)))
    errors = []

    def work(tid):
        try:
            for j in range(tid, len(jobs), 8):
                d = jobs[j]
                dst = bytearray(lz.CompressBlockBound(len(d)))
                k = lz.CompressBlock(d, dst)
                assert bytes(dst[:k]) == want[j][0], ("fast", j)
                back = bytearray(len(d))
                assert lz.UncompressBlock(bytes(dst[:k]), back) == len(d) and bytes(back) == d, ("decode", j)
                dst2 = bytearray(lz.CompressBlockBound(min(len(d), 20000)))
                k2 = lz.CompressBlockHC(d[:20000], dst2, 16)
                assert bytes(dst2[:k2]) == want[j][1], ("hc", j)
                assert lz.ChecksumZero(d) == want[j][2], ("xxh32", j)
        except Exception as e:  # noqa: BLE001
            errors.append((tid, repr(e)))

    ts = [threading.Thread(target=work, args=(t,)) for t in range(8)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors


def test_quirk_paths_differential(oracle):
    """Differential fuzz aimed at the reference's corner rules rather than at bulk data:
    matches that run into the last mfLimit (12) / 5 bytes of a block (block.go:126,184-193,
    363-372: Fast stops at sn = len-14 in 8-byte strides, HC's prefix loop may overshoot
    sn-si by up to 7), every dst_len from "one byte short" to the bound (T5: (0,nil) vs error),
    periods 1..9, Fast and three HC depths."""
    import lz4_amd as lz
    rng = np.random.default_rng(2024)
    cases = []
    for n in (17, 18, 19, 25, 26, 27, 28, 29, 33, 40, 41, 64, 65, 77, 130, 271, 300):
        for period in (1, 2, 3, 4, 5, 7, 9):
            unit = rng.integers(97, 123, period, dtype=np.uint8).tobytes()
            head = rng.integers(0, 256, int(rng.integers(0, 9)), dtype=np.uint8).tobytes()
            body = (head + unit * (n // period + 2))[:n]
            if rng.integers(0, 3) == 0:          # break the run a few bytes before the end
                k = n - 1 - int(rng.integers(0, 14))
                body = body[:k] + bytes([body[k] ^ 0x55]) + body[k + 1:]
            cases.append(body)
    for src in cases[::2]:
        n = len(src)
        bound = lz.CompressBlockBound(n)
        for dst_len in sorted({n // 3, n // 2, n - 1, n, n + 1, bound - 1, bound}):
            for hc in (None, 1, 4, 65536):
                assert _gpu_compress(lz, src, dst_len, hc) == _cpu_compress(oracle, src, dst_len, hc), \
                    (n, dst_len, hc, src[:16])


def test_hc_single_block_beyond_32mib_match(oracle):
    """One call, 40 MiB: a run longer than 2^25 bytes (the match length alone fills most of a 32-bit
    word: the winner is picked in two steps, hc.hip), noise that the parse steps over in growing
    jumps, and dense zero words behind the 8 MiB mark, where the position map's marks repeat and the
    dense-chain scan must not trust them (block.go:319-481)."""
    import lz4_amd as lz
    rng = np.random.default_rng(11)
    parts = [rng.integers(0, 256, 3000, dtype=np.uint8), np.zeros((34 << 20) + 123, np.uint8)]
    noise = rng.integers(0, 256, 2 << 20, dtype=np.uint8)
    for q in rng.integers(0, noise.size - 8, 20000):
        noise[q:q + 4] = 0
    parts.append(noise)
    tail = np.zeros(3 << 20, np.uint8)
    tail[rng.integers(0, tail.size, tail.size // 61)] = rng.integers(1, 256, tail.size // 61, dtype=np.uint8)
    parts.append(tail)
    parts.append(noise[:1 << 20])
    src = np.concatenate(parts).tobytes()
    bound = lz.CompressBlockBound(len(src))
    assert _gpu_compress(lz, src, bound, 131072) == _cpu_compress(oracle, src, bound, 131072)
