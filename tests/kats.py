"""Known-answer vectors transcribed as DATA from the reference's own tests.
Shared by the CPU (oracle) and GPU (HIP) tests.

Sources (paths relative to the pierrec/lz4 checkout):
  DECODE       internal/lz4block/decode_test.go:73-140  (TestBlockDecode)
  INVALID      internal/lz4block/decode_test.go:159-207 (TestDecodeBlockInvalid)
  WITH_DICT    internal/lz4block/decode_test.go:262-284 (TestDecodeWithDict)
  XXH32        internal/xxh32/xxh32zero_test.go:17-32
  HEADERS      internal/lz4stream/frame_test.go:19-22
  CONTENT_XXH  SURVEY.md §8c: content checksums of testdata/*.lz4 (frames made by the C lz4 tool)
"""


def _append_len(p, size):
    while size > 0xFF:
        p.append(0xFF)
        size -= 0xFF
    p.append(size)


def emit_seq(lit, offset, match_len):
    """The emitSeq helper of decode_test.go:20-64."""
    lit = bytes(lit)
    lit_len = len(lit)
    if lit_len < 15:
        b = lit_len << 4
        lit_len = -1
    else:
        b = 0xF0
        lit_len -= 15
    out = bytearray()
    if match_len < 4 or offset == 0:
        out.append(b)
        if lit_len >= 0:
            _append_len(out, lit_len)
        return bytes(out + lit)
    match_len -= 4
    if match_len < 15:
        b |= match_len
        match_len = -1
    else:
        b |= 0x0F
        match_len -= 15
    out.append(b)
    if lit_len >= 0:
        _append_len(out, lit_len)
    out += lit
    out += bytes((offset & 0xFF, offset >> 8))
    if match_len >= 0:
        _append_len(out, match_len)
    return bytes(out)


DECODE = [
    ("empty_input", b"\x00", b""),
    ("literal_only_short", emit_seq(b"hello", 0, 0), b"hello"),
    ("literal_only_long", emit_seq(b"A" * (15 + 255 + 255 + 1), 0, 0), b"A" * (15 + 255 + 255 + 1)),
    ("literal_only_long_1", emit_seq(b"A" * 15, 0, 0), b"A" * 15),
    ("literal_only_long_2", emit_seq(b"A" * 16, 0, 0), b"A" * 16),
    ("repeat_match_len", emit_seq(b"a", 1, 4), b"aaaaa"),
    ("repeat_match_len_2_seq", emit_seq(b"a", 1, 4) + emit_seq(b"B", 1, 4), b"aaaaaBBBBB"),
    ("long_match", emit_seq(b"A", 1, 16), b"A" * 17),
    ("memmove_last", emit_seq(b"ABCD" * 20, 80, 60), (b"ABCD" * 36)[:140]),
    ("repeat_match_log_len_2_seq",
     emit_seq(b"a", 1, 15) + emit_seq(b"B", 1, 15) + emit_seq(b"end", 0, 0),
     b"a" * 16 + b"B" * 16 + b"end"),
    ("fuzz-3e4ce8cc0da392ca5a353b6ffef6d08f400ac5f9", b"0000\x01\x00", b"0000000"),
]

INVALID = [
    ("empty_input", b"", 100),
    ("repeated_zero_few", b"\x00" * 6, 100),
    ("repeated_zero_many", b"\x00" * 100, 100),
    ("final_lit_too_short", b"\x20a", 100),
    ("no_space_for_offset", b"\x01", 10),
    ("write_beyond_len_dst", b"\x1b0\x01\x00000000000000", len(b"\x1b0\x01\x00000000000000")),
    ("bounds-crasher", b"\x000000", 10),
    ("zero_offset", b"\xe1abcdefghijklmn\x00\x00\xe0abcdefghijklmn", 40),
]

WITH_DICT = [  # (src, dict, want or None for error)
    (b"\x11b\x0a\x00\x401234", b"barbazquux", b"barbaz1234"),
    (b"\x35foo\x09\x00\x401234", b"0barbaz", b"foobarbazfoo1234"),
    (b"\x08\x04\x00\x50abcde", b"---1234", b"123412341234abcde"),
    (b"\x1a1\x05\x00\x50abcde", b"---2345", b"123451234512345abcde"),
    (b"\x35abc\x09\x00\xf0\x0f0123456789abcdefghijklmnopqrst", b"012defghi",
     b"abcdefghiabc0123456789abcdefghijklmnopqrst"),
    (b"\x35foo\xff\xff\x401234", b"XYZ", None),
]

XXH32 = [
    (0x02cc5d05, b""),
    (0x550d7456, b"a"),
    (0x4999fc53, b"ab"),
    (0x32d153ff, b"abc"),
    (0xa3643705, b"abcd"),
    (0x9738f19b, b"abcde"),
    (0x8b7cd587, b"abcdef"),
    (0x9dd093b3, b"abcdefg"),
    (0x0bb3c6bb, b"abcdefgh"),
    (0xd03c13fd, b"abcdefghi"),
    (0x8b988cfe, b"abcdefghij"),
    (0x9d2d8b62, b"abcdefghijklmnop"),
    (0x42ae804d, b"abcdefghijklmnopqrstuvwxyz0123456789"),
    (0x62b4ed00, b"Lorem ipsum dolor sit amet, consectetur adipiscing elit, sed do eiusmod tempor "
                 b"incididunt ut labore et dolore magna aliqua. Ut enim ad minim veniam, quis nostrud "
                 b"exercitation ullamco laboris nisi ut aliquip ex ea commodo consequat. Duis aute "
                 b"irure dolor in reprehenderit in voluptate velit esse cillum dolore eu fugiat nulla "
                 b"pariatur. Excepteur sint occaecat cupidatat non proident, sunt in culpa qui officia "
                 b"deserunt mollit anim id est laborum."),
]

HEADERS = [  # (block size, FLG BD HC) with content checksum on
    (1 << 16, b"\x64\x40\xa7"),
    (1 << 18, b"\x64\x50\x08"),
    (1 << 20, b"\x64\x60\x85"),
    (1 << 22, b"\x64\x70\xb9"),
]

GOLDEN_FRAMES = {  # frame file -> (raw file, content checksum)
    "pg1661.txt.lz4": ("pg1661.txt", 0x3970effb),
    "e.txt.lz4": ("e.txt", 0x07ffb51d),
    "gettysburg.txt.lz4": ("gettysburg.txt", 0x8df0920e),
    "repeat.txt.lz4": ("repeat.txt", 0x70938b91),
    "random.data.lz4": ("random.data", 0xa88a2f73),
    "pi.txt.lz4": ("pi.txt", 0xa3ac4742),
    "Mark.Twain-Tom.Sawyer.txt.lz4": ("Mark.Twain-Tom.Sawyer.txt", 0xf1574086),
    "pg_control.tar.lz4": ("pg_control.tar", 0xdbd91b0a),
}

# Legacy-format golden (magic 0x184C2102, frame.go:20), made by the Linux kernel build.
LEGACY_FRAME = "bzImage_lz4_isolated.lz4"

# Candidate compressed sizes from SURVEY.md §6 (an independent throwaway
# transliteration made during the survey; NOT a Go run).
CANDIDATE_SIZES = {  # file -> (Fast, HC depth 512, HC depth 131072)
    "gettysburg.txt": (1323, 1180, 1180),
    "repeat.txt": (52, 49, 49),
    "random.data": (16450, 16450, 16450),
    "pg_control.tar": (376, 335, 335),
    "upperbound.data": (607, 596, 596),
    "issue51.data": (4824, 4473, 4473),
    "pg1661.txt": (329190, None, None),
    "e.txt": (100397, None, None),
    "pi.txt": (98139, None, None),
    "Mark.Twain-Tom.Sawyer.txt": (226723, None, None),
}


def legacy_frames(compress_block):
    """Synthetic legacy frames (lz4stream/block.go:279-317 legacy rules), built from
    `compress_block(bytes) -> bytes`.  Returns [(name, frame, expected or None)]."""
    import struct
    magic = struct.pack("<I", 0x184C2102)
    a = (b"legacy block one " * 500)[:7001]
    b = bytes(range(256)) * 40
    ca, cb = compress_block(a), compress_block(b)
    blk = lambda c: struct.pack("<I", len(c)) + c
    raw = lambda d: struct.pack("<I", len(d) | 0x80000000) + d
    return [
        ("two_blocks", magic + blk(ca) + blk(cb), a + b),
        ("kernel_trailer", magic + blk(ca) + blk(cb) + struct.pack("<I", len(a) + len(b)), a + b),
        ("trailer_then_garbage", magic + blk(ca) + struct.pack("<I", len(a)) + b"\xff" * 9, a),
        ("concatenated", magic + blk(ca) + magic + blk(cb), a + b),
        ("raw_block", magic + raw(b"abc" * 11) + blk(ca), b"abc" * 11 + a),
        ("empty", magic, b""),
        ("truncated_payload", magic + blk(ca) + blk(cb)[:-5], None),
        ("oversize_word", magic + blk(ca) + struct.pack("<I", (8 << 20) + 1) + b"x" * 64, None),
        ("bad_block", magic + struct.pack("<I", 6) + b"\xf0\x00\x00\x00\x00\x00", None),
    ]


# Goldens made by the C lz4 tool from one 4,266,361-byte text (reader_test.go:40-46): the
# same content as 4 MiB independent blocks and as 64 KiB LINKED blocks.  The raw file is
# not shipped by the reference; both frames carry its XXH32 as their content checksum.
LONG_FRAMES = {
    "Mark.Twain-Tom.Sawyer_long.txt.lz4": (4266361, 0x788577F8),
    "Mark.Twain-Tom.Sawyer_linked.txt.lz4": (4266361, 0x788577F8),
}


def _seq(lit, offset=0, mlen=0):
    """One LZ4 sequence (decode_test.go emitSeq): literals then, when mlen >= 4, a match."""
    out = bytearray()
    ll = len(lit)
    ml = mlen - 4 if mlen else 0
    out.append((min(ll, 15) << 4) | (min(ml, 15) if mlen else 0))
    if ll >= 15:
        r = ll - 15
        out += b"\xff" * (r // 255) + bytes([r % 255])
    out += lit
    if mlen:
        out += bytes([offset & 0xFF, offset >> 8])
        if ml >= 15:
            r = ml - 15
            out += b"\xff" * (r // 255) + bytes([r % 255])
    return bytes(out)


def linked_frames(xxh32):
    """Hand-built LINKED-block frames (FLG bit 5 clear; reader.go:176-189).
    `xxh32(bytes) -> int`.  Returns [(name, frame, expected bytes or None for an error)]."""
    import struct

    def frame(blocks, bsum=False, csum=False, expect=b"", bad_cs=False):
        flg = 0x40 | (0x10 if bsum else 0) | (0x04 if csum else 0)
        hdr = bytes([flg, 0x40])
        out = struct.pack("<I", 0x184D2204) + hdr + bytes([(xxh32(hdr) >> 8) & 0xFF])
        for payload, raw, decoded in blocks:
            out += struct.pack("<I", len(payload) | (0x80000000 if raw else 0)) + payload
            if bsum:
                x = xxh32(decoded)
                if x:
                    out += struct.pack("<I", x)
        out += struct.pack("<I", 0)
        if csum:
            out += struct.pack("<I", xxh32(expect) ^ (1 if bad_cs else 0))
        return out

    a = bytes(range(97, 123)) * 3                       # 78 bytes of history
    b1 = _seq(a)
    # block 2: 5 literals, then 20 bytes from 30 bytes back (all inside block 1)
    b2 = _seq(b"HELLO", 30 + 5, 20) + _seq(b"tail!")
    d2 = b"HELLO" + a[-30:-10] + b"tail!"
    # block 3: match that starts in the history and runs over its own output (overlap)
    b3 = _seq(b"", 3, 40) + _seq(b"end..")
    h3 = a + d2
    d3 = bytearray()
    for k in range(40):
        d3.append((h3 + bytes(d3))[len(h3) + k - 3])
    d3 = bytes(d3) + b"end.."
    raw4 = b"RAWBLOCK" * 4
    b5 = _seq(b"", len(raw4), 12) + _seq(b"12345")
    d5 = raw4[:12] + b"12345"
    blocks = [(b1, False, a), (b2, False, d2), (b3, False, d3)]
    full = a + d2 + d3
    out = [
        ("three_blocks", frame(blocks, expect=full), full),
        ("all_checksums", frame(blocks, bsum=True, csum=True, expect=full), full),
        ("raw_block_as_history",
         frame(blocks + [(raw4, True, raw4), (b5, False, d5)], bsum=True, csum=True,
               expect=full + raw4 + d5), full + raw4 + d5),
        ("bad_content_checksum", frame(blocks, csum=True, expect=full, bad_cs=True), None),
        ("offset_before_history", frame([(b1, False, a), (_seq(b"x", 80, 4) + _seq(b"yyyyy"), False, b"")]), None),
        ("first_block_needs_history", frame([(_seq(b"ab", 3, 4) + _seq(b"yyyyy"), False, b"")]), None),
    ]
    # far history: 70,000 bytes of filler, then a match 65,535 back
    filler1 = bytes((i * 7 + 3) & 0xFF for i in range(60000))
    filler2 = bytes((i * 13 + 5) & 0xFF for i in range(10000))
    far = _seq(b"Z", 65535, 8) + _seq(b"done.")
    hist = filler1 + filler2 + b"Z"
    dfar = b"Z" + hist[len(hist) - 65535:len(hist) - 65535 + 8] + b"done."
    out.append(("offset_65535", frame([(_seq(filler1), False, filler1), (_seq(filler2), False, filler2),
                                       (far, False, dfar)], csum=True, expect=filler1 + filler2 + dfar),
                filler1 + filler2 + dfar))
    return out


def legacy_write_cases(load):
    """Inputs for LegacyOption(true) (writer_test.go:256-299 + edge cases):
    [(name, data, readfrom, level, block_checksum)]."""
    import numpy as np
    bz = load("bzImage_lz4_isolated")
    rnd = np.random.default_rng(7).integers(0, 256, 2 * (8 << 20), dtype=np.uint8).tobytes()
    return [
        ("bzImage_write", bz, False, 0, False),
        ("bzImage_iocopy", bz, True, 0, False),
        ("empty_write", b"", False, 0, False),
        ("empty_iocopy", b"", True, 0, False),
        ("short_incompressible", rnd[:100], False, 0, False),       # grows: 102 bytes, not raw
        ("two_raw_blocks_iocopy", rnd, True, 0, False),             # raw blocks + 01 00 00 00 00
        ("block_and_tail", (bz * 9)[:(8 << 20) + 12345], False, 0, False),
        ("exact_block_iocopy", (bz * 9)[:8 << 20], True, 0, True),
        ("hc_level", bz[:300000], False, 1 << 9, False),
    ]


def ragged_frames(compress_block, xxh32):
    """Independent-block frames whose middle blocks are short (a Writer that was Flush()ed
    mid-stream: writer.go:148-169).  Returns [(name, frame, expected)]."""
    import struct
    pieces = [bytes((i * 31 + k) & 0xFF for i in range(n)) if k % 2 else (b"flush me " * 9000)[:n]
              for k, n in enumerate([100, 65536, 7, 3000, 65536, 1])]
    out = []
    for bsum, csum in ((False, False), (True, True)):
        flg = 0x60 | (0x10 if bsum else 0) | (0x04 if csum else 0)
        hdr = bytes([flg, 0x40])
        f = struct.pack("<I", 0x184D2204) + hdr + bytes([(xxh32(hdr) >> 8) & 0xFF])
        for d in pieces:
            c = compress_block(d)
            if c and len(c) < len(d):
                f += struct.pack("<I", len(c)) + c
            else:
                f += struct.pack("<I", len(d) | 0x80000000) + d
            if bsum and xxh32(d):
                f += struct.pack("<I", xxh32(d))
        f += struct.pack("<I", 0)
        if csum:
            f += struct.pack("<I", xxh32(b"".join(pieces)))
        out.append(("ragged_bsum%d_csum%d" % (bsum, csum), f, b"".join(pieces)))
    return out
