#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats result (rocpd sqlite .db) as text:
per-kernel calls, total / average / min / max duration.  Usage:
    python tools/rocprof_summary.py gpurun_out/prof/x_results.db > profiles/NAME.txt"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("# rocprofv3 --kernel-trace --stats summary of %s" % path)
    print("%-72s %7s %14s %14s %14s %14s %6s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct"))
    for name, calls, tot, avg, mn, mx in rows:
        print("%-72s %7d %14d %14.0f %14d %14d %6.2f" % (name[:72], calls, tot, avg, mn, mx, 100.0 * tot / total))


if __name__ == "__main__":
    main(sys.argv[1])
