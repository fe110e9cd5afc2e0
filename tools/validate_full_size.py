#!/usr/bin/env python3
"""One-off full-size parity checks of BASELINE shapes, GPU pipeline vs the CPU oracle on the same arrays:
  validate_full_size.py                 configs[4]: ne240 (3.1 M columns) x 128 x 1, fp32 inputs (~40 GB host RAM)
  validate_full_size.py 30x72x91 f64    one rank's block of configs[2]
(development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import tem_oracle as orc
from pytemdiags_amd import _lib, engine, synth

ne, nlev, nt = (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "240x128x1").split("x"))
td = torch.float64 if len(sys.argv) > 2 and sys.argv[2] == "f64" else torch.float32
lat, lon = synth.cubed_sphere_gll(ne)
plev = synth.pressure_levels(nlev)
f = engine.synth_fields(0, lat, lon, plev, nt, dtype=td, seed=0)
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
print("ne%dx%dx%d %s inputs: N=%d, one_pass=%s, oracle %.1f s, worst field-normalised error %.3e, nonfinite=%s"
      % (ne, nlev, nt, str(td)[6:], lat.size, plan.one_pass, t_cpu, worst, bad))
