#!/usr/bin/env python3
"""Per-launch-shape kernel table from a rocprofv3 --kernel-trace csv: one row per (kernel, grid size), and when the
launches of one kernel on one grid fall into two clearly separate groups of durations (the reference pre-pass and the
main sweep of the single-sweep form launch the same kernel, at ne120 x 72 x 30 even with the same grid) one row per
group -- so that no average mixes a 0.1 ms launch with a 9 ms one.
   kernel_table.py <dir or *_kernel_trace.csv> [top N]"""
import csv
import glob
import math
import os
import sys
from collections import defaultdict


def grid_total(r):
    if r.get("Grid_Size"):
        return int(r["Grid_Size"])
    return int(r.get("Grid_Size_X") or 1) * int(r.get("Grid_Size_Y") or 1) * int(r.get("Grid_Size_Z") or 1)


def grid_text(r):
    if r.get("Grid_Size_X"):
        return "x".join(r.get(k, "1") for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
    return str(r.get("Grid_Size", ""))


def clusters(vals, ratio=4.0):
    """[(label, values)]: one group, or two when the values separate by more than `ratio` around sqrt(max * min)."""
    lo, hi = min(vals), max(vals)
    if lo <= 0 or hi / lo < ratio:
        return [("", vals)]
    cut = math.sqrt(lo * hi)
    a, b = [v for v in vals if v < cut], [v for v in vals if v >= cut]
    if not a or not b or min(b) / max(a) < 2.0:
        return [("", vals)]
    return [(" (long)", b), (" (short)", a)]


def table(path):
    if os.path.isdir(path):
        path = max(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
    acc = defaultdict(list)
    text = {}
    with open(path) as fh:
        for r in csv.DictReader(fh):
            k = (r["Kernel_Name"], grid_total(r))
            text[k] = grid_text(r)
            acc[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    rows = []
    for k, v in acc.items():
        for label, vals in clusters(v):
            rows.append((sum(vals), len(vals), sum(vals) / len(vals), min(vals), k[0], text[k] + label))
    rows.sort(reverse=True)
    return rows


if __name__ == "__main__":
    rows = table(sys.argv[1])
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
    tot = sum(r[0] for r in rows)
    print("%-84s %20s %6s %10s %10s %6s" % ("kernel", "grid", "calls", "avg us", "min us", "%"))
    for t, n, avg, mn, name, grid in rows[:top]:
        print("%-84s %20s %6d %10.1f %10.1f %6.1f" % (name[:84], grid, n, avg, mn, 100 * t / tot))
