"""How the fixtures under tests/golden were made (run in the build container,
where the reference checkout is mounted at /root/reference).

Fixtures are DATA the reference's own tests hold -- test inputs and golden
frames -- never reference source:
  testdata/*                    copied from /root/reference/testdata
                                (raw inputs + *.lz4 frames made by the C lz4 tool,
                                testdata/README.txt)
  fuzz_uncompress_corpus.json   the 108 decoder fuzz seeds of
                                /root/reference/fuzz/uncompress/corpus, hex-encoded
The reference is Go and cannot run here (no Go toolchain), so there are no
reference-generated compressed outputs among the fixtures.
"""
import json
import os
import shutil

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
FILES = """Mark.Twain-Tom.Sawyer.txt Mark.Twain-Tom.Sawyer.txt.lz4
Mark.Twain-Tom.Sawyer_long.txt.lz4 Mark.Twain-Tom.Sawyer_linked.txt.lz4 bzImage_lz4_isolated
bzImage_lz4_isolated.lz4 e.txt e.txt.lz4 empty.txt gettysburg.txt gettysburg.txt.lz4 issue102.data
issue51.data malformed.block.lz4 pg1661.txt pg1661.txt.lz4 pg_control.tar pg_control.tar.lz4 pi.txt
pi.txt.lz4 random.data random.data.lz4 repeat.txt repeat.txt.lz4 upperbound.data""".split()
# (Mark.Twain-Tom.Sawyer_long.txt itself, issue43.data and vmlinux_LZ4_19377[.lz4] are listed in the
# reference's .MISSING_LARGE_BLOBS and absent from the checkout: the _long / _linked frames are
# checked through their content checksum and against each other, tests/kats.py LONG_FRAMES.)

def check():
    """Every committed fixture is listed above, and nothing listed is missing."""
    have = sorted(os.listdir(os.path.join(HERE, "testdata")))
    assert have == sorted(FILES), sorted(set(have) ^ set(FILES))


if __name__ == "__main__":
    os.makedirs(os.path.join(HERE, "testdata"), exist_ok=True)
    for f in FILES:
        shutil.copyfile(os.path.join(REF, "testdata", f), os.path.join(HERE, "testdata", f))
    d = os.path.join(REF, "fuzz", "uncompress", "corpus")
    seeds = {f: open(os.path.join(d, f), "rb").read().hex() for f in sorted(os.listdir(d))}
    with open(os.path.join(HERE, "fuzz_uncompress_corpus.json"), "w") as fh:
        json.dump(seeds, fh, indent=0, sort_keys=True)
