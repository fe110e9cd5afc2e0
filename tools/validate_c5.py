#!/usr/bin/env python3
"""One-off full-size parity check of BASELINE configs[4]: ne240 (3.1 M columns) x 128 x 1 with fp32
inputs, GPU pipeline vs the CPU oracle on the same arrays (development aid; needs ~40 GB of host RAM)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import tem_oracle as orc
from pytemdiags_amd import _lib, engine, synth

ne, nlev, nt = 240, 128, 1
lat, lon = synth.cubed_sphere_gll(ne)
plev = synth.pressure_levels(nlev)
f = engine.synth_fields(0, lat, lon, plev, nt, dtype=torch.float32, seed=0)
host = [x.cpu().numpy() for x in f]
t0 = time.time()
ref = orc.TEMOracle(*host, lat, plev, mode="factorised")
t_cpu = time.time() - t0
plan = engine.Plan(lat, ref.lat, 50)
plan.set_tem(nlev, nt, plev * 100)
res, _ = plan.tem_run(*f)
bad = plan.status()
res = res.cpu().numpy()
worst = 0.0
for i, n in enumerate(_lib.RESULT_NAMES):
    r = getattr(ref, n)().astype(np.float64)
    e = float(np.max(np.abs(res[i] - r)) / np.max(np.abs(r)))
    worst = max(worst, e)
    print("%-10s %.3e" % (n, e), flush=True)
print("ne240x128x1 fp32 inputs: N=%d, one_pass=%s, oracle %.1f s, worst field-normalised error %.3e, nonfinite=%s"
      % (lat.size, plan.one_pass, t_cpu, worst, bad))
