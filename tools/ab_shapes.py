#!/usr/bin/env python3
"""Development aid: time a list of shapes with the library named by TEMX_LIB (A/B builds).
  TEMX_LIB=tools/ab/libtemx_A.so python tools/ab_shapes.py 30x72x1 240x128x1:f32 ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.argv, args = sys.argv[:1], sys.argv[1:]
import torch
import quick_bench as q

print("lib:", os.environ.get("TEMX_LIB", "default"), {k: v for k, v in os.environ.items() if k.startswith("TEMX_") and k != "TEMX_LIB"}, flush=True)
for a in args:
    shape, _, dt = a.partition(":")
    ne, nlev, nt = (int(x) for x in shape.split("x"))
    q.run(ne, nlev, nt, reps=50 if ne * ne * nlev * nt < 5e6 else 8, dtype=torch.float32 if dt == "f32" else torch.float64)
