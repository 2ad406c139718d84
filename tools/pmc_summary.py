#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counters (counter_collection.csv) over the dispatches of one kernel.
    python tools/pmc_summary.py DIR KERNEL_SUBSTRING [INPUT_BYTES]"""
import csv, glob, sys, collections
d, key = sys.argv[1], sys.argv[2]
nbytes = float(sys.argv[3]) if len(sys.argv) > 3 else 0
tot, disp = collections.Counter(), set()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if key in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
            disp.add(r["Dispatch_Id"])
print("# %s: %d dispatches of *%s*" % (d, len(disp), key))
for k in sorted(tot):
    print("%-28s %16.0f" % (k, tot[k]) + ("  per input byte %.3f" % (tot[k] / nbytes) if nbytes else ""))
