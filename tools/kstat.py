#!/usr/bin/env python3
"""Print per-kernel average/min durations (us) from a rocprofv3 results .db (development aid)."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, count(*), avg(end-start), min(end-start) from kernels group by name order by 3 desc"))
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    print("%-72s %5d %9.1f %9.1f" % (r[0][:72], r[1], r[2] / 1e3, r[3] / 1e3))
