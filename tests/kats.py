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
