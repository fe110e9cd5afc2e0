#!/usr/bin/env python3
"""Run the TEM pipeline of one shape a few times (for rocprofv3 --kernel-trace --stats):
  shape_probe.py 240x128x1 f32 [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.argv, a = sys.argv[:1], sys.argv[1:]
import torch
import quick_bench as q
ne, nlev, nt = (int(x) for x in a[0].split("x"))
q.run(ne, nlev, nt, reps=int(a[2]) if len(a) > 2 else 20, dtype=torch.float32 if len(a) > 1 and a[1] == "f32" else torch.float64)
