#!/usr/bin/env python3
"""Per-dispatch rows of a rocprofv3 --pmc run (counter_collection.csv) for the lz4hip kernels:
one line per dispatch with every counter of the pass, so that per-byte figures can be re-derived
from what is committed under profiles/.
    python tools/prof/pmc_rows.py DIR [INPUT_BYTES]"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
nbytes = float(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = collections.OrderedDict()
for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "lz4hip" not in r["Kernel_Name"]:
            continue
        k = (int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0].replace("lz4hip::", ""))
        rows.setdefault(k, collections.Counter())[r["Counter_Name"]] += float(r["Counter_Value"])
for (disp, name), c in sorted(rows.items()):
    parts = ["%s=%.0f" % (k, v) + (" (%.4f/B)" % (v / nbytes) if nbytes else "") for k, v in sorted(c.items())]
    print("dispatch %-4d %-28s %s" % (disp, name, "  ".join(parts)))
