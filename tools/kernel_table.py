#!/usr/bin/env python3
"""Per-launch-shape kernel table from a rocprofv3 --kernel-trace csv: one row per (kernel, grid size), so that two
launches of one kernel with different grids (the reference pre-pass and the main sweep) do not share an average.
   kernel_table.py <dir or *_kernel_trace.csv> [top N]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def table(path):
    if os.path.isdir(path):
        path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[-1]
    acc = defaultdict(list)
    with open(path) as fh:
        for r in csv.DictReader(fh):
            grid = r.get("Grid_Size") or "x".join(r.get(k, "") for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
            acc[(r["Kernel_Name"], grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    rows = [(sum(v), len(v), sum(v) / len(v), min(v), k[0], k[1]) for k, v in acc.items()]
    rows.sort(reverse=True)
    return rows


if __name__ == "__main__":
    rows = table(sys.argv[1])
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
    tot = sum(r[0] for r in rows)
    print("%-84s %12s %6s %10s %10s %6s" % ("kernel", "grid", "calls", "avg us", "min us", "%"))
    for t, n, avg, mn, name, grid in rows[:top]:
        print("%-84s %12s %6d %10.1f %10.1f %6.1f" % (name[:84], grid, n, avg, mn, 100 * t / tot))
