#!/usr/bin/env python3
"""bench.py -- Fast compress + decompress throughput of the HIP LZ4 block codec.

Metric (BASELINE.json): GB/s of UNCOMPRESSED bytes, whole job, Fast compress +
decompress, inputs resident in HBM when the timed region starts.

One step = one pass of the hot path over one synthetic batch per GPU:
  (a) cfg3  CompressBlock Fast over `--size-gib` GiB of the "mixed" corpus cut in
      256 KiB independent blocks, frame semantics (dst_len = len(src)), xxh32
      block checksums on;
  (b) cfg2  UncompressBlock over `--size-gib` GiB of the "text" corpus that was
      pre-compressed (64 KiB independent blocks, packed as frame entries).
value = steps * (N_a + N_b) * n_gpus / time.  Blocks shard by contiguous range,
one process per GPU, no data-path collective ("scaling": "weak").

After the timed region the same run exercises the cfg5 FRAME path through
lz4_amd.shard (Writer: compress + span gather to rank 0 + assembly; Reader: index,
span scatter, decode, verification), with and without the content checksum, on
`--frame-gib` GiB per GPU of the mixed corpus in 4 MiB blocks, and reports it under
"frame_path" (extra keys; the headline metric does not include it).  A watchdog prints the
headline without that section if it does not finish within `--frame-timeout` seconds.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size-gib G]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from lz4_amd import _lib, corpus, device as dev  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
RANDOM_RW_PEAK_GREQ = 49.9  # profiles/r1_random_access_microbench.txt: random reads, 8 GiB footprint
# (read+write pairs: 40.3, writes alone: 22.1 -- reads are the upper bound of any mix)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size-gib", type=float, default=8.0, help="GiB per corpus per GPU")
    ap.add_argument("--frame-timeout", type=int, default=240,
                    help="seconds after which the headline is printed without the frame-path section")
    ap.add_argument("--frame-gib", type=float, default=16.0,
                    help="GiB per GPU for the cfg5 frame path after the timed region (cfg5's share is 16 "
                         "GiB = 4096 blocks of 4 MiB; halved while free HBM does not allow it; 0: skip)")
    ap.add_argument("--hc-gib", type=float, default=4.0,
                    help="GiB per corpus of the cfg4 (HC Level9) extra section on rank 0 (0: skip)")
    ap.add_argument("--cpu-sample-mib", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    _lib.require_gpu()
    # LZ4_BENCH_ONE_DEVICE=1 rehearses the N>1 path on a single-GPU box: every rank
    # uses cuda:0 and the (tiny) control collectives run over gloo.
    one_dev = os.environ.get("LZ4_BENCH_ONE_DEVICE") == "1"
    dev_index = 0 if one_dev else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    # one process per GPU over RCCL; a single rank still gets a group of one, so that the
    # frame path below is the same code at every N
    if one_dev:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        import datetime
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device,
                                timeout=datetime.timedelta(minutes=5))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    n = int(args.size_gib * (1 << 30)) // (1 << 18) * (1 << 18)
    BS_C, BS_D = 1 << 18, 1 << 16
    start = rank * n  # each rank owns its own contiguous range of the corpus

    # ---- inputs, resident in HBM ----
    bases = corpus.device_bases(torch, device)

    def gen(kind):
        out = torch.empty(n, dtype=torch.uint8, device=device)
        ch = 1 << 24
        for a in range(0, n, ch):
            c = min(ch, n - a)
            out[a:a + c] = corpus.generate(kind, start + a, c, xp=torch, device=device, bases=bases)
        return out

    mixed = gen("mixed")
    text = gen("text")
    nb_c, nb_d = n // BS_C, n // BS_D

    # cfg2 input: text pre-compressed into 64 KiB independent blocks, packed like a frame body
    comp_t, res_t = dev.compress_blocks(text, BS_D, 0)
    off_t, sw_t = dev.frame_layout(res_t, n, BS_D, nb_d)
    packed_len = int(off_t[nb_d].item())
    packed = torch.empty(packed_len + 16, dtype=torch.uint8, device=device)
    dev.frame_scatter(text, BS_D, comp_t, res_t, off_t, nb_d, packed)
    src_off = (off_t[:nb_d] + 4).contiguous()
    c_text = packed_len - 4 * nb_d
    del comp_t
    torch.cuda.empty_cache()

    # work buffers
    comp = torch.empty((nb_c, BS_C), dtype=torch.uint8, device=device)
    res = torch.zeros(nb_c, dtype=torch.int32, device=device)
    xxh = torch.empty(nb_c, dtype=torch.int32, device=device)
    out = torch.empty((nb_d, BS_D), dtype=torch.uint8, device=device)
    out_len = torch.empty(nb_d, dtype=torch.int32, device=device)

    ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
    marks = []

    def step(record):
        e = [ev() for _ in range(4)] if record else None
        if record:
            e[0].record()
        dev.compress_blocks(mixed, BS_C, 0, out=comp, result=res)
        if record:
            e[1].record()
        dev.xxh32_blocks(mixed, BS_C, nb_c, out=xxh)
        if record:
            e[2].record()
        dev.uncompress_blocks(packed, src_off, sw_t, BS_D, out=out, out_len=out_len)
        if record:
            e[3].record()
            marks.append(e)

    for _ in range(args.warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if one_dev else device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    t_comp = sum(e[0].elapsed_time(e[1]) for e in marks) / len(marks) / 1e3
    t_xxh = sum(e[1].elapsed_time(e[2]) for e in marks) / len(marks) / 1e3
    t_dec = sum(e[2].elapsed_time(e[3]) for e in marks) / len(marks) / 1e3

    # ---- parity (outside the timed region) ----
    ok_dec = bool((out_len == BS_D).all().item()) and torch.equal(out.view(-1), text)
    res_h = res.cpu().numpy()
    c_mixed = int(np.where(res_h > 0, res_h, BS_C).sum())
    parity = {"decode_equals_corpus": ok_dec}

    cpu = None
    if rank == 0:
        from oracle import pyoracle as orc  # checker + cpu_baseline leg only
        rng = np.random.default_rng(1)
        pick = sorted(set(rng.integers(0, nb_c, 24).tolist()) | {0, 1, 2, 3, nb_c - 1})
        ok = True
        for b in pick:
            blk = mixed[b * BS_C:(b + 1) * BS_C].cpu().numpy()
            pl, sw, xx = orc.compress_blocks(blk, BS_C, 0, 1, True)
            r = int(res_h[b])
            want_r = 0 if sw[0] >> 31 else int(sw[0])
            got = comp[b, :max(r, 0)].cpu().numpy().tobytes()
            ok &= (r if r > 0 else 0) == want_r and got == pl[0, :want_r].tobytes()
            ok &= (int(xxh[b].item()) & 0xFFFFFFFF) == int(xx[0])
            # corpus regenerated on the host must equal the device bytes
            if b in (0, nb_c - 1):
                ok &= bool((corpus.generate("mixed", start + b * BS_C, BS_C) == blk).all())
        parity["compress_sample_byte_identical"] = bool(ok)
        parity["sampled_blocks"] = len(pick)

        if not args.no_cpu_baseline and world == 1:  # the CPU leg runs at N=1 only
            threads = min(len(os.sched_getaffinity(0)), 16)
            m = min(args.cpu_sample_mib << 20, n) // BS_C * BS_C
            h_mixed = mixed[:m].cpu().numpy()
            h_text = text[:m].cpu().numpy()
            pl_t, sw_tc, _ = orc.compress_blocks(h_text, BS_D, 0, threads, False)
            reps, tc, td = 0, 0.0, 0.0
            while tc + td < 10.0 and reps < 50:
                a = time.perf_counter()
                orc.compress_blocks(h_mixed, BS_C, 0, threads, True)
                b_ = time.perf_counter()
                orc.uncompress_blocks(pl_t, sw_tc, BS_D, threads)
                c_ = time.perf_counter()
                tc += b_ - a
                td += c_ - b_
                reps += 1
            cpu = {"value": round(2 * m * reps / (tc + td) / 1e9, 3), "unit": "GB/s", "cores": threads,
                   "kind": "port",
                   "sample": "%d MiB mixed Fast-compressed (256 KiB blocks, xxh32) + %d MiB text "
                             "decoded (64 KiB blocks), %d reps, C oracle with %d pthreads "
                             "(no Go toolchain on the box)" % (m >> 20, m >> 20, reps, threads),
                   "compress_GBps": round(m * reps / tc / 1e9, 3),
                   "decompress_GBps": round(m * reps / td / 1e9, 3)}

    line = None
    if rank == 0:
        total = args.steps * 2 * n * world
        alg_bytes = n + c_mixed + 8 * nb_c  # N read + C written + 8 B/block
        # HBM-side traffic of the dominant kernel from the committed PMC passes of this
        # same workload (profiles/traffic.json: per-input-byte figures, tools/pmc_bench.sh);
        # a launch's traffic is those figures times the launch's input bytes.
        traffic, ra = None, None
        try:
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                tj = json.load(f)["k_fast_gtab"]
            traffic = round(tj["traffic_bytes_per_byte"] * n)
            req = (tj["read_requests_per_byte"] + tj["write_requests_per_byte"]) * n
            ra = {"achieved_Greq_s": round(req / t_comp / 1e9, 2), "peak_Greq_s": RANDOM_RW_PEAK_GREQ,
                  "frac": round(req / t_comp / 1e9 / RANDOM_RW_PEAK_GREQ, 3),
                  "note": "fabric requests of the random 2-byte table/candidate accesses vs the measured "
                          "random-read rate of the chip (tools/micro/randacc.hip, 8 GiB footprint)"}
        except (OSError, KeyError, ValueError):
            pass
        line = {
            "metric": "GB/s uncompressed (node) for Fast compress + decompress",
            "value": round(total / dt / 1e9, 3),
            "unit": "GB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": "cfg3 Fast compress (%.2f GiB/GPU mixed text/binary/random, 256KB blocks, "
                            "xxh32 block checksums) + cfg2 decode (%.2f GiB/GPU text-like, 64KB blocks)"
                            % (n / 2**30, n / 2**30),
                "bytes_per_gpu_per_step": 2 * n,
                "blocks_compress": nb_c, "blocks_decode": nb_d,
                "ratio_mixed": round(c_mixed / n, 4), "ratio_text": round(c_text / n, 4),
                "sharding": "contiguous block ranges, one process per GPU, no collective"},
            "compress_GBps": round(n / (t_comp + t_xxh) / 1e9, 3),
            "decompress_GBps": round(n / t_dec / 1e9, 3),
            "kernel_ms": {"fast_compress": round(t_comp * 1e3, 3), "xxh32_blocks": round(t_xxh * 1e3, 3),
                          "decode": round(t_dec * 1e3, 3)},
            "roofline": {"bound": "hbm", "kernel": "k_fast_gtab",
                         "achieved": round(alg_bytes / t_comp / 1e9, 3), "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": round(alg_bytes / t_comp / 1e9 / HBM_PEAK_GBPS, 5),
                         "traffic": traffic,
                         "traffic_source": "profiles/traffic.json (rocprofv3 --pmc passes of this workload "
                                           "at 8 GiB, bytes per input byte) x this launch's input bytes; "
                                           "not collected in this run",
                         "random_access": ra,
                         "decode_achieved": round((n + c_text) / t_dec / 1e9, 3),
                         "decode_frac": round((n + c_text) / t_dec / 1e9 / HBM_PEAK_GBPS, 5)},
            "cpu_baseline": cpu,
            "parity": parity,
        }

    # The line is complete but for the extra frame-path section.  That section is the only
    # part of this program with rank-to-rank transfers; a watchdog prints the headline
    # without it and ends the process if it (or the closing barrier) does not come back, so a
    # stuck transfer cannot take the measured line with it.
    import threading
    emit_lock = threading.Lock()
    emitted = [False]

    def emit(fp):
        with emit_lock:
            if emitted[0] or line is None:
                return
            emitted[0] = True
            line["frame_path"] = fp
            print(json.dumps(line), flush=True)

    def bail():
        emit({"error": "the frame path did not finish within %d s" % args.frame_timeout})
        sys.stdout.flush()
        os._exit(3)   # a stuck transfer must not reach the driver as a successful run

    guard = threading.Timer(args.frame_timeout, bail)
    guard.daemon = True
    guard.start()

    # ---- cfg5 frame path through lz4_amd.shard (outside the headline's timed region) ----
    frame_path = None
    if args.frame_gib > 0:
        try:
            from lz4_amd import shard
            BS_F = 1 << 22
            del comp, out, packed, text
            torch.cuda.empty_cache()
            share = int(args.frame_gib * (1 << 30)) // BS_F * BS_F
            # source + slots + entries + frame + decoded copy: ~5x the share must fit
            while share > BS_F and torch.cuda.mem_get_info(device)[0] < 5 * share + (8 << 30):
                share = share // 2 // BS_F * BS_F
            if share <= n:
                src_f = mixed[:share]               # this rank's block range of the stream
            else:
                del mixed
                torch.cuda.empty_cache()
                src_f = torch.empty(share, dtype=torch.uint8, device=device)
                for a in range(0, share, 1 << 24):
                    c = min(1 << 24, share - a)
                    src_f[a:a + c] = corpus.generate("mixed", rank * share + a, c, xp=torch, device=device,
                                                     bases=bases)
            n_total = share * world
            if rank > 0:
                try:
                    shard.reserve_staging(share)    # pinned memory of the checksum relay, outside the timing
                except RuntimeError:
                    pass                            # (the relay then streams the shard in its turn)
            fp = {"write_GBps": {}, "read_GBps": {}, "phases_s": {}}
            ok_rt = True
            for label, flags in (("no_content_checksum", 1), ("content_checksum", 3)):
                ws, rs = {}, {}
                shard.frame_compress_sharded(src_f[:BS_F], BS_F * world, BS_F, flags, 0)   # warm-up
                barrier()
                t0 = time.perf_counter()
                frame = shard.frame_compress_sharded(src_f, n_total, BS_F, flags, 0, stats=ws)
                barrier()
                t1 = time.perf_counter()
                flat, _, tot = shard.frame_uncompress_sharded(frame, frame.numel() if rank == 0 else 0, stats=rs)
                barrier()
                t2 = time.perf_counter()
                ok_rt &= tot == n_total and torch.equal(flat, src_f)
                fp["write_GBps"][label] = round(n_total / (t1 - t0) / 1e9, 3)
                fp["read_GBps"][label] = round(n_total / (t2 - t1) / 1e9, 3)
                fp["phases_s"][label] = {"write": {k: round(v, 4) for k, v in ws.items() if k.endswith("_s")},
                                         "read": {k: round(v, 4) for k, v in rs.items() if k.endswith("_s")}}
                if rank == 0 and flags == 1:
                    gb = ws.get("gather_bytes", 0)
                    peak = 153.0 * min(world - 1, 7)
                    fp["gather"] = {"bytes": gb, "seconds": round(ws.get("gather_s", 0.0), 4),
                                    "GBps": round(gb / max(ws.get("gather_s", 0.0), 1e-9) / 1e9, 2) if gb else 0.0,
                                    "peak_GBps": peak,
                                    "note": "spans received straight into the frame, one xGMI link per sender "
                                            "(7 x 153 GB/s inbound at the root)"}
                    fp["frame_bytes"] = int(frame.numel())
                del frame, flat
            flag = torch.tensor([1 if ok_rt else 0], dtype=torch.int64, device="cpu" if one_dev else device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            fp["round_trip_equals_source"] = bool(flag.item())
            fp["workload"] = ("cfg5 share: %.2f GiB/GPU (%d blocks/GPU) of the mixed corpus, 4MB blocks, xxh32 "
                              "block checksums, Writer + Reader through lz4_amd.shard; content checksum chain "
                              "on the host" % (share / 2**30, share // BS_F))
            frame_path = fp
        except Exception as e:  # the headline line must survive a failure of this extra section
            frame_path = {"error": "%s: %s" % (type(e).__name__, e)}
    # ---- cfg4 (HC Level9, 4 GiB of 1 MiB blocks per corpus) on rank 0: an extra key, not the headline ----
    if rank == 0 and args.hc_gib > 0 and isinstance(frame_path, dict):
        try:
            torch.cuda.empty_cache()
            n_hc = int(args.hc_gib * (1 << 30))
            hc = {"workload": "cfg4: %.2f GiB per corpus, 1MB blocks, HC depth 131072 (Level9), one wave per block"
                              % (n_hc / 2**30)}
            for kind in ("binary", "text"):
                src_h = corpus.generate_device(kind, n_hc, device)
                nb_h = n_hc >> 20
                from lz4_amd import _lib as _l
                ws_h = torch.empty(_l.load().lz4hip_hc_workspace_bytes(nb_h), dtype=torch.uint8, device=device)
                out_h = torch.empty((nb_h, 1 << 20), dtype=torch.uint8, device=device)
                res_h = torch.zeros(nb_h, dtype=torch.int32, device=device)
                dev.compress_blocks(src_h[:1 << 24], 1 << 20, 131072, out=out_h, result=res_h, workspace=ws_h)  # warm-up
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                outs = dev.compress_blocks(src_h, 1 << 20, 131072, out=out_h, result=res_h, workspace=ws_h)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1)
                res = outs[1]
                csz = int(torch.where(res > 0, res, torch.full_like(res, 1 << 20)).sum().item())
                hc[kind + "_GBps"] = round(n_hc / ms / 1e6, 3)
                hc[kind + "_ms"] = round(ms, 1)
                hc[kind + "_ratio"] = round(csz / n_hc, 4)
                del src_h, outs, res, ws_h, out_h, res_h
                torch.cuda.empty_cache()
            frame_path["hc_cfg4"] = hc
        except Exception as e:
            frame_path["hc_cfg4"] = {"error": "%s: %s" % (type(e).__name__, e)}
    emit(frame_path)

    dist.barrier()
    guard.cancel()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
