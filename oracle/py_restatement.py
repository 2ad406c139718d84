"""Second, independent restatement of the reference's block compressors in pure
Python.  TEST INFRASTRUCTURE ONLY (differential partner of oracle/lz4_oracle.c).

Written from the algorithm description in SURVEY.md §3.1/§3.4/§8 (traps T1-T5),
with its own control structure (generators of sequences + a separate emitter)
so that a slip in one restatement is unlikely to be repeated in the other.
Reference anchors: internal/lz4block/block.go:105-295 (Fast), :319-481 (HC).

Pure-Python speed (~1 MB/s): use on inputs up to a few hundred KB.
"""

WIN = 1 << 16
MASK64 = (1 << 64) - 1


class ShortBuffer(Exception):
    pass


def bound(n):
    return n + n // 255 + 16


def hash6(x):
    """block.go:35-38"""
    return (((x << 16) & MASK64) * 227718039650203 & MASK64) >> 48


def hash4(x):
    """block.go:298-301"""
    return ((x * 2654435761) & 0xFFFFFFFF) >> 16


def _u64(src, i):
    return int.from_bytes(src[i:i + 8], "little")


def _u32(src, i):
    return int.from_bytes(src[i:i + 4], "little")


def fast_sequences(src):
    """Yield (anchor, lit_len, offset, match_len) for each match of the greedy
    Fast parse; the caller adds the last literals."""
    n = len(src)
    sn = n - 14
    table = {}  # hash -> low 16 bits of the last position put (absent == 0: T1)

    def get(h, si):
        i = table.get(h, 0) + (si & ~0xFFFF)
        return i - WIN if i >= si else i

    si = anchor = 0
    while si < sn:
        x = _u64(src, si)
        hs = (hash6(x), hash6(x >> 8), hash6(x >> 16))
        r0, r1 = get(hs[0], si), get(hs[1], si + 1)  # both gets precede both puts (T3)
        table[hs[0]] = si & 0xFFFF
        table[hs[1]] = (si + 1) & 0xFFFF

        def ok(ref, pos, word):
            off = pos - ref
            return 0 < off < WIN and _u32(src, ref) == word

        found = None
        if ok(r0, si, x & 0xFFFFFFFF):
            found = (si, r0)
        else:
            r2 = get(hs[2], si + 2)  # sees the two puts above
            if ok(r1, si + 1, (x >> 8) & 0xFFFFFFFF):
                found = (si + 1, r1)
            else:
                table[hs[2]] = (si + 2) & 0xFFFF
                if ok(r2, si + 2, (x >> 16) & 0xFFFFFFFF):
                    found = (si + 2, r2)
        if found is None:
            si = si + 4 + ((si + 2 - anchor) >> 7)
            continue
        pos, ref = found
        off = pos - ref
        # backward extension (T4)
        back = 0
        while pos - back > anchor and ref - back - 1 >= 0 and src[pos - back - 1] == src[ref - back - 1]:
            back += 1
        start = pos - back
        end = pos + 4
        while end + 8 <= sn:
            d = _u64(src, end) ^ _u64(src, end - off)
            if d == 0:
                end += 8
            else:
                end += ((d & -d).bit_length() - 1) >> 3
                break
        yield anchor, start - anchor, off, end - start
        anchor = si = end
        if si >= sn:
            break
        table[hash6(_u64(src, si - 2))] = (si - 2) & 0xFFFF


def hc_sequences(src, depth):
    n = len(src)
    sn = n - 14
    if depth == 0:
        depth = WIN
    head = [0] * WIN
    chain = [0] * WIN
    si = anchor = 0
    while si < sn:
        h = hash4(_u32(src, si))
        best, best_off = 0, 0
        nxt, tries = head[h], depth
        while tries > 0 and nxt > 0 and si - nxt < WIN:
            if src[nxt + best] == src[si + best]:
                ml = 0
                while ml < sn - si:
                    d = _u64(src, nxt + ml) ^ _u64(src, si + ml)
                    if d == 0:
                        ml += 8
                    else:
                        ml += ((d & -d).bit_length() - 1) >> 3
                        break
                if ml >= 4 and ml > best:
                    best, best_off = ml, si - nxt
            nxt = chain[nxt & 0xFFFF]
            tries -= 1
        chain[si & 0xFFFF] = head[h]
        head[h] = si
        if best == 0:
            si += 1 + ((si - anchor) >> 7)
            continue
        lo = max(si + 1, si + best - WIN)
        roll = _u32(src, si)  # rolled from si even when lo skips ahead (block.go:399-400)
        for p in range(lo, si + best):
            roll = (roll >> 8) | (src[p + 3] << 24)
            hh = hash4(roll)
            chain[p & 0xFFFF] = head[hh]
            head[hh] = p
        yield anchor, si - anchor, best_off, best
        si += best
        anchor = si


def _emit(src, seqs, dst_len, hc):
    """Serialise sequences + last literals with the reference's dst-capacity
    behaviour (T5).  Returns compressed bytes, b"" for (0, nil), or raises."""
    n = len(src)
    incompressible = dst_len < bound(n)
    out = bytearray()
    anchor = 0
    any_match = False

    def put(b):
        if len(out) >= dst_len:
            raise ShortBuffer()
        out.append(b)

    def put_len_tail(v):  # v = length - 15
        while v >= 255:
            put(255)
            v -= 255
        put(v)

    for a, lit, off, ml in seqs:
        any_match = True
        m = ml - 4
        put((min(lit, 15) << 4) | min(m, 15))
        if lit >= 15:
            put_len_tail(lit - 15)
        if len(out) + lit > dst_len:
            raise ShortBuffer()
        out += src[a:a + lit]
        if len(out) + 2 > dst_len:
            raise ShortBuffer()
        out += bytes((off & 0xFF, off >> 8))
        if m >= 15:
            put_len_tail(m - 15)
        anchor = a + lit + ml
    short_input = n - 14 <= 0
    if incompressible and not any_match and not (hc and short_input):
        return b""
    lit = n - anchor
    put(min(lit, 15) << 4)
    if lit >= 15:
        put_len_tail(lit - 15)
    if incompressible and len(out) >= anchor:
        return b""
    if len(out) + lit > dst_len:
        raise ShortBuffer()
    out += src[anchor:]
    return bytes(out)


def compress_fast(src, dst_len=None):
    src = bytes(src)
    if dst_len is None:
        dst_len = bound(len(src))
    return _emit(src, fast_sequences(src), dst_len, hc=False)


def compress_hc(src, depth, dst_len=None):
    src = bytes(src)
    if dst_len is None:
        dst_len = bound(len(src))
    return _emit(src, hc_sequences(src, depth), dst_len, hc=True)


def decode(src, dst_len, dict=b""):
    """Safe LZ4 block decode, byte-serial matches (decode_other.go:10-137).
    Returns bytes or raises ShortBuffer."""
    src = bytes(src)
    n = len(src)
    if n == 0:
        raise ShortBuffer()
    out = bytearray()
    si = 0
    while si < n:
        tok = src[si]
        si += 1
        lit = tok >> 4
        if lit == 15:
            while True:
                if si >= n:
                    raise ShortBuffer()
                b = src[si]
                si += 1
                lit += b
                if b != 255:
                    break
        if si + lit > n or len(out) + lit > dst_len:
            raise ShortBuffer()
        out += src[si:si + lit]
        si += lit
        ml = tok & 15
        if si == n and ml == 0:
            break
        if si + 2 > n:
            raise ShortBuffer()
        off = src[si] | (src[si + 1] << 8)
        si += 2
        if off == 0:
            raise ShortBuffer()
        ml += 4
        if ml == 19:
            while True:
                if si >= n:
                    raise ShortBuffer()
                b = src[si]
                si += 1
                ml += b
                if b != 255:
                    break
        if len(out) + ml > dst_len:
            raise ShortBuffer()
        if off > len(out) + len(dict):
            raise ShortBuffer()
        for _ in range(ml):
            p = len(out) - off
            out.append(out[p] if p >= 0 else dict[len(dict) + p])
    return bytes(out)
