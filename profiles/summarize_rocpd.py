#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls, total / average / min / max duration.
usage: summarize_rocpd.py <results.db> [> summary.txt]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                  "from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows) or 1
print("%-64s %8s %14s %12s %12s %12s %6s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "%"))
for name, calls, total, avg, mn, mx in rows:
    short = re.sub(r"\(.*", "", name)
    short = re.sub(r"^void ", "", short)
    print("%-64s %8d %14d %12.0f %12d %12d %6.2f" % (short[:64], calls, total, avg, mn, mx, 100.0 * total / tot))
