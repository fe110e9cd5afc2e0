#!/usr/bin/env python3
"""Development aid: time temx_tracer_run (after a TEM run on the same fields), one-pass vs two-pass.
  tracer_bench.py 120x72x30"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pytemdiags_amd import engine, synth

ne, nlev, nt = (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "30x72x30").split("x"))
lat, lon = synth.cubed_sphere_gll(ne)
plev = synth.pressure_levels(nlev)
e = np.arange(-90, 91, 1.0); lat_zm = (e[1:] + e[:-1]) / 2
f = engine.synth_fields(0, lat, lon, plev, nt, dtype=torch.float64, seed=0)
q = engine.synth_fields(0, lat, lon, plev, nt, dtype=torch.float64, seed=5)[2]      # a T-like field as the tracer
for form in ("one", "two"):
    os.environ.pop("TEMX_TWO_PASS", None)
    if form == "two":
        os.environ["TEMX_TWO_PASS"] = "1"
    plan = engine.Plan(lat, lat_zm, 50)
    plan.set_tem(nlev, nt, plev * 100)
    plan.tem_run(*f)
    for _ in range(2):
        plan.tracer_run(q, f[1], f[3])
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    a.record()
    for _ in range(reps):
        plan.tracer_run(q, f[1], f[3])
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    pts = lat.size * nlev * nt
    print("ne%dx%dx%d tracer_run %s (plan.one_pass=%s): %.3f ms -> %.3g grid-points/s, %.2f TB/s of the 3 fields"
          % (ne, nlev, nt, form, plan.one_pass, ms, pts / ms * 1e3, 3 * 8 * pts / ms / 1e9), flush=True)
    plan.close()
