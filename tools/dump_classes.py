#!/usr/bin/env python3
"""Development aid: the latitude classes of a cubed-sphere grid as tools/sweep_lab.hip reads them (LAB_CLASSES):
dump_classes.py ne out.bin [split_equator=0|1] [order=size|lat]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pytemdiags_amd import synth
ne = int(sys.argv[1]); out = sys.argv[2]
split = len(sys.argv) > 3 and sys.argv[3] == "1"
order = sys.argv[4] if len(sys.argv) > 4 else "size"
lat, lon = synth.cubed_sphere_gll(ne)
a = np.round(np.abs(lat), 10)
u, inv = np.unique(a, return_inverse=True)
idx = np.argsort(inv, kind="stable")
bounds = np.searchsorted(inv[idx], np.arange(len(u) + 1))
cls = []
for c in range(len(u)):
    rows = idx[bounds[c]:bounds[c + 1]]
    n = np.sort(rows[lat[rows] >= -1e-12]); s = np.sort(rows[lat[rows] < -1e-12])
    if split and (len(n) + 3) // 4 + (len(s) + 3) // 4 > 16:
        parts = max((len(n) + 7) // 8, (len(s) + 7) // 8)
        for k in range(parts):
            cls.append((n[8 * k:8 * k + 8], s[8 * k:8 * k + 8]))
    else:
        cls.append((n, s))
nb = lambda m: (len(m) + 3) // 4
if order in ("size", "shuffle", "shuffle4"):
    cls.sort(key=lambda c: (-nb(c[0]), -nb(c[1]), int(c[0][0]) if len(c[0]) else int(c[1][0])))
if order == "shuffle":           # classes in random order inside each (size) stratum
    rng = np.random.default_rng(1)
    keys = [(nb(c[0]), nb(c[1])) for c in cls]
    i = 0
    while i < len(cls):
        j = i
        while j < len(cls) and keys[j] == keys[i]:
            j += 1
        blk = cls[i:j]
        perm = rng.permutation(len(blk))
        cls[i:j] = [blk[k] for k in perm]
        i = j
if order == "shuffle4":          # groups of 4 consecutive classes stay together, the groups are shuffled
    rng = np.random.default_rng(1)
    keys = [(nb(c[0]), nb(c[1])) for c in cls]
    i = 0
    while i < len(cls):
        j = i
        while j < len(cls) and keys[j] == keys[i]:
            j += 1
        blk = cls[i:j]
        ng = (len(blk) + 3) // 4
        perm = rng.permutation(ng)
        cls[i:j] = [c for g in perm for c in blk[4 * g:4 * g + 4]]
        i = j
with open(out, "wb") as fh:
    fh.write(np.int32(len(cls)).tobytes())
    for n, s in cls:
        fh.write(np.array([len(n), len(s)], dtype=np.int32).tobytes())
        fh.write(n.astype(np.int32).tobytes()); fh.write(s.astype(np.int32).tobytes())
print(len(cls), "classes ->", out)
