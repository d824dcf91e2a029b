#!/usr/bin/env python3
"""Per-kernel summary (calls, total/avg/min/max duration) of a rocprofv3 rocpd .db, as CSV on stdout.
Usage: rocpd_summary.py results.db > profiles/rNN_kernel_stats.csv"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                  "from kernels group by name order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows) or 1
print("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage")
for n, c, s, a, mn, mx in rows:
    print('"%s",%d,%d,%.1f,%d,%d,%.2f' % (n, c, s, a, mn, mx, 100.0 * s / tot))
