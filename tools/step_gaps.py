#!/usr/bin/env python
"""Gaps between consecutive kernel dispatches in a rocprofv3 --kernel-trace DB (developer tool): every gap above a threshold with
the kernels on both sides, to spot stalls inside a replayed HIP graph.   python tools/step_gaps.py <results.db> [min_us]"""
import re
import sqlite3
import sys

db, thr = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
con = sqlite3.connect(db)
rows = con.execute("select name, start, end from kernels order by start").fetchall()
short = lambda n: re.sub(r"\(anonymous namespace\)::|void |\(.*$", "", n)[:40]
for i in range(1, len(rows)):
    gap = (rows[i][1] - rows[i - 1][2]) / 1e3
    if thr <= gap < 2000:
        print("%6d  gap %8.1f us  %-40s -> %s" % (i, gap, short(rows[i - 1][0]), short(rows[i][0])))
