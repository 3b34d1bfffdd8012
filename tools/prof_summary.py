#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace results DB (rocpd sqlite) into the per-kernel table committed under profiles/."""
import sqlite3
import sys

db, steps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1
con = sqlite3.connect(db)
rows = con.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels "
                   "group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print("# rocprofv3 --kernel-trace --stats summary; %d kernel dispatches, %.3f ms total GPU kernel time over %d steps "
      "(%.3f ms/step, %d launches/step)" % (sum(r[1] for r in rows), tot / 1e6, steps, tot / 1e6 / steps, sum(r[1] for r in rows) // steps))
print("%-78s %7s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
for r in rows:
    print("%-78s %7d %12.1f %10.2f %10.2f %10.2f %6.2f" % (r[0][:78], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / tot))
