#!/usr/bin/env python3
"""Turn the PMC passes of tools/pmc_bench.sh into profiles/traffic.json (per-input-byte figures).
    python tools/traffic_json.py gpurun_out/pmc_bench_TAG SIZE_GIB > profiles/traffic.json
The timed Fast launch is the LAST k_fast_gtab dispatch (the first one builds the decode input)."""
import collections, csv, glob, json, sys
d, gib = sys.argv[1], float(sys.argv[2])
n = int(gib * 2**30)
vals = collections.defaultdict(dict)   # kernel -> counter -> value of the last dispatch
for p in ("ea", "fetch", "write"):
    per = collections.defaultdict(dict)
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (d, p), recursive=True):
        for r in csv.DictReader(open(f)):
            if "lz4hip" in r["Kernel_Name"]:
                k = r["Kernel_Name"].split("(")[0].replace("lz4hip::", "")
                e = per[k].setdefault(int(r["Dispatch_Id"]), collections.Counter())
                e[r["Counter_Name"]] += float(r["Counter_Value"])
    for k, disp in per.items():
        vals[k].update(disp[max(disp)])
def per_byte(c):
    rd = c["TCC_EA0_RDREQ_sum"] * 64
    wr = c["TCC_EA0_WRREQ_64B_sum"] * 64 + (c["TCC_EA0_WRREQ_sum"] - c["TCC_EA0_WRREQ_64B_sum"]) * 32
    return {"read_requests_per_byte": round(c["TCC_EA0_RDREQ_sum"] / n, 4),
            "write_requests_per_byte": round(c["TCC_EA0_WRREQ_sum"] / n, 4),
            "read_bytes_per_byte": round(rd / n, 3), "write_bytes_per_byte": round(wr / n, 3),
            "FETCH_SIZE_bytes_per_byte": round(c["FETCH_SIZE"] * 1024 / n, 3),
            "WRITE_SIZE_bytes_per_byte": round(c["WRITE_SIZE"] * 1024 / n, 3),
            "traffic_bytes_per_byte": round((rd + wr) / n, 3)}
out = {"method": "rocprofv3 --pmc, one run per counter group (TCC_EA0_RDREQ/_32B/WRREQ/_64B; FETCH_SIZE; WRITE_SIZE): "
                 "tools/pmc_bench.sh = python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --size-gib %g; figures per "
                 "input byte of the timed launch; bytes = RDREQ*64 + WRREQ_64B*64 + (WRREQ-WRREQ_64B)*32, which equals "
                 "FETCH_SIZE + WRITE_SIZE as reported (the x2 FETCH_SIZE correction of MI355X_MICROARCH.md is for wide "
                 "streaming reads; these are random 2-byte gathers)" % gib,
       "input_bytes_per_launch_measured": n}
for k in ("k_fast_gtab", "k_decode", "k_decode3", "k_xxh32_blocks"):
    if k in vals:
        out[k] = per_byte(vals[k])
print(json.dumps(out, indent=1))
