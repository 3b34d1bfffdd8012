"""Average PMC counters per kernel name from a rocprofv3 --pmc results DB."""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
cols = [c[1] for c in con.execute("pragma table_info(counters_collection)")]
rows = con.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall() if "kernel_name" in cols else []
if not rows:
    print(cols); sys.exit()
by = {}
for k, c, v, n in rows:
    by.setdefault(k, {})[c] = (v, n)
for k, d in by.items():
    if "conv" not in k and "wgrad" not in k and (len(sys.argv) < 3 or sys.argv[2] not in k):
        continue
    print(k[:90])
    for c, (v, n) in sorted(d.items()):
        print("   %-28s %14.1f  (n=%d)" % (c, v, n))
