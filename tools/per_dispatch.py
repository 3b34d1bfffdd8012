#!/usr/bin/env python
"""Per-dispatch durations of one kernel (substring match) over the LAST step of a rocprofv3 --kernel-trace DB: the sequence of
launches of a step in order, so that two builds / knob settings can be compared layer by layer."""
import sqlite3, sys
db, pat, per_step = sys.argv[1], sys.argv[2], int(sys.argv[3])
con = sqlite3.connect(db)
rows = con.execute("select start, end-start from kernels where name like ? order by start", ("%" + pat + "%",)).fetchall()
last = rows[-per_step:]
print(" ".join("%.1f" % (d / 1e3) for _, d in last))
