"""The N>1 frame path with the PRODUCT compute ops, as part of the GPU suite: 2 and 3 ranks run
as fresh child processes on the one GPU of the box (tests/shard_gpu_worker.py; gloo carries the
control plane and, staged through the host, the spans that RCCL carries on a node).  Writer:
rank 0's frame equals the oracle's single-process frame byte for byte (span lengths, span
offsets, the gather into the frame, the XXHZero state relay with ranks > 0 staging their shard
into pinned memory).  Reader: the ranks' pieces in rank order are the source; the first failing
block in stream order decides the error on every rank (lz4stream/block.go:187-194).

An 8-GPU node has not been available to any round: the RCCL transfers themselves are still
unmeasured; this covers everything around them."""
import json
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
BS = 1 << 16


def _run(world, job, tmp_path):
    jf = tmp_path / "job.json"
    jf.write_text(json.dumps(job))
    port = 23000 + (os.getpid() * 7 + world * 131 + len(job["writes"]) + 17 * len(job["reads"])) % 4000
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if world == 3:      # a staging cap below the shard sizes: part of a shard staged, the rest streamed
        env["LZ4_SHARD_STAGE_GIB"] = "0.0005"
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "shard_gpu_worker.py"), str(r), str(world),
                               str(port), str(jf), str(tmp_path)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out.decode(errors="replace"))
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d:\n%s" % (r, outs[r][-3000:])
    return [json.loads((tmp_path / ("rank%d.json" % r)).read_text()) for r in range(world)]


@pytest.mark.parametrize("world", [2, 3])
def test_writer_and_reader_ranks_on_one_gpu(world, oracle, tmp_path):
    from lz4_amd import _lib, corpus
    writes = [
        # both checksums, ragged tail; HC level; ReadFrom's empty block behind a block multiple;
        # fewer blocks than ranks
        {"n": 37 * BS + 999, "bs": BS, "flags": 3, "kind": "mixed", "start": 5},
        {"n": 9 * BS + 17, "bs": BS, "flags": 3, "kind": "text", "start": 77, "level": 512},
        {"n": 8 * BS, "bs": BS, "flags": 3 | 8, "kind": "mixed", "start": 1 << 20},
        {"n": 5 * (1 << 22) + 3, "bs": 1 << 22, "flags": 3, "kind": "mixed", "start": 3 << 20},
        {"n": 1000, "bs": BS, "flags": 2, "kind": "text", "start": 0},
    ]
    datas = [corpus.generate(w["kind"], w["start"], w["n"]) for w in writes]
    want = [oracle.frame_compress(d, w["bs"], w["flags"], w.get("level", 0)) for d, w in zip(datas, writes)]

    # reader inputs: the good frames and damaged copies of the first one
    good = want[0]
    offs, si = [], 7
    while struct.unpack_from("<I", good, si)[0]:
        offs.append(si)
        si += 8 + (struct.unpack_from("<I", good, si)[0] & 0x7FFFFFFF)
    # a late COMPRESSED block (the mixed corpus also holds stored ones, whose damage is a checksum matter)
    late = next(b for b in range(25, len(offs)) if not struct.unpack_from("<I", good, offs[b])[0] >> 31)
    bad_sum = bytearray(good)
    bad_sum[offs[late] + 4:offs[late] + 8] = b"\xff\xff\xff\xff"  # corrupt payload in a late block
    bad_sum[offs[3] - 1] ^= 1                                      # checksum word of block 2: comes first
    bad_payload = bytearray(good)
    bad_payload[offs[late] + 4:offs[late] + 8] = b"\xff\xff\xff\xff"
    truncated = bytes(good[:offs[35] + 100])
    bad_content = bytearray(good)
    bad_content[-1] ^= 1
    frames = [good, want[3], bytes(bad_sum), bytes(bad_payload), truncated, bytes(bad_content), want[2]]
    reads = []
    for i, f in enumerate(frames):
        path = tmp_path / ("in%d.lz4" % i)
        path.write_bytes(f)
        reads.append({"frame": str(path), "gather": i == 6})
    rep = _run(world, {"writes": writes, "reads": reads}, tmp_path)

    for i, w in enumerate(want):
        got = (tmp_path / ("write%d.lz4" % i)).read_bytes()
        assert got == w, "write %d: %d vs %d bytes" % (i, len(got), len(w))

    def pieces(i):
        return [np.fromfile(tmp_path / ("read%d.rank%d.bin" % (i, r)), dtype=np.uint8).tobytes()
                for r in range(world)]

    for i, src in ((0, datas[0]), (1, datas[3])):
        assert all(rep[r]["reads"][i][0] == "ok" for r in range(world)), [rep[r]["reads"][i] for r in range(world)]
        p = pieces(i)
        assert b"".join(p) == src.tobytes()
        assert [rep[r]["reads"][i][1] for r in range(world)] == [sum(len(x) for x in p[:r]) for r in range(world)]
        assert all(rep[r]["reads"][i][2] == src.size for r in range(world))
    assert [rep[r]["reads"][2] for r in range(world)] == [["err", _lib.ERR_BLOCK_CHECKSUM, 2]] * world
    assert [rep[r]["reads"][3] for r in range(world)] == [["err", _lib.ERR_BLOCK, late]] * world
    assert [rep[r]["reads"][4] for r in range(world)] == [["err", _lib.ERR_FRAME, 35]] * world
    assert [rep[r]["reads"][5][:2] for r in range(world)] == [["err", _lib.ERR_FRAME_CHECKSUM]] * world
    assert rep[0]["reads"][6][0] == "ok" and pieces(6)[0] == datas[2].tobytes()      # gather=True
