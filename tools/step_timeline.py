#!/usr/bin/env python
"""Ordered timeline of the LAST step in a rocprofv3 --kernel-trace DB: one line per dispatch with its duration, the gap to the
previous dispatch's end and the grid, so that launch-bound stretches and individual GEMM / conv launches can be read off.
    python tools/step_timeline.py <results.db> <launches per step>"""
import re
import sqlite3
import sys

db, per_step = sys.argv[1], int(sys.argv[2])
con = sqlite3.connect(db)
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
g = [c for c in ("grid_size_x", "grid_size_y", "grid_size_z", "workgroup_size_x", "lds_block_size", "vgpr_count") if c in cols]
rows = con.execute("select name, start, end%s from kernels order by start" % "".join(", " + c for c in g)).fetchall()
last = rows[-per_step:]
prev_end = None
tot = gaps = 0.0
for r in last:
    name = re.sub(r"\(anonymous namespace\)::|void |\(.*$", "", r[0])[:60]
    gap = 0.0 if prev_end is None else (r[1] - prev_end) / 1e3
    d = (r[2] - r[1]) / 1e3
    tot += d; gaps += max(gap, 0.0)
    print("%-60s %8.2f us  gap %6.2f  %s" % (name, d, gap, " ".join("%s=%s" % (c.replace("_size", ""), v) for c, v in zip(g, r[3:]))))
    prev_end = r[2]
print("# %d dispatches, kernel time %.1f us, gaps %.1f us, span %.1f us" % (len(last), tot, gaps, (last[-1][2] - last[0][1]) / 1e3))
