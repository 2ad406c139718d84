"""One rank of the sharded frame path on ONE GPU (run as a fresh child process by
tests/test_gpu_shard_ranks.py, never imported by pytest): the product compute ops
(lz4_amd.shard.DeviceOps -> liblz4hip.so) on cuda:0, control plane and span transfers over gloo
(device tensors staged through the host by shard._Wire).  What RCCL would carry on a node
travels through the host here; everything else is the code bench.py --gpus N runs.

    python shard_gpu_worker.py RANK WORLD PORT JOB.json OUTDIR
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    job = json.load(open(sys.argv[4]))
    outdir = sys.argv[5]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from lz4_amd import _lib, corpus, shard
    _lib.require_gpu()
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ops = shard.DeviceOps()
    report = {"writes": [], "reads": []}
    for i, w in enumerate(job["writes"]):
        n, bs, flags, level = w["n"], w["bs"], w["flags"], w.get("level", 0)
        nb = (n + bs - 1) // bs
        lo, hi = shard.block_range(nb, rank, world)
        a, b = min(n, lo * bs), min(n, hi * bs)
        local = torch.zeros(b - a + 16, dtype=torch.uint8, device="cuda")[:b - a]
        if b > a:
            local.copy_(torch.from_numpy(corpus.generate(w["kind"], w["start"] + a, b - a)))
        frame = shard.frame_compress_sharded(local, n, bs, flags, level, ops=ops)
        if rank == 0:
            frame.cpu().numpy().tofile(os.path.join(outdir, "write%d.lz4" % i))
        report["writes"].append(int(frame.numel()) if rank == 0 else None)
    for i, r in enumerate(job["reads"]):
        f, n = None, 0
        if rank == 0:
            raw = np.fromfile(r["frame"], dtype=np.uint8)
            n = raw.size
            f = torch.zeros(n + 16, dtype=torch.uint8, device="cuda")[:n]
            f.copy_(torch.from_numpy(raw))
        try:
            flat, start, total = shard.frame_uncompress_sharded(f, n, ops=ops, gather=r.get("gather", False))
            flat.cpu().numpy().tofile(os.path.join(outdir, "read%d.rank%d.bin" % (i, rank)))
            report["reads"].append(["ok", int(start), int(total)])
        except shard.ShardError as e:
            report["reads"].append(["err", e.code, e.block])
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()
    with open(os.path.join(outdir, "rank%d.json" % rank), "w") as fh:
        json.dump(report, fh)


if __name__ == "__main__":
    main()
