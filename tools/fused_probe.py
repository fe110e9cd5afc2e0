#!/usr/bin/env python3
"""Development aid: sweep times of tem_run and tem_tracer_run on the cubed sphere and on a lat-lon grid of the
same size whose latitude classes all have 8 + 8 members."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pytemdiags_amd import engine, synth

nlev, nt = 72, 30
plev = synth.pressure_levels(nlev)
e = np.arange(-90, 91, 1.0); lat_zm = (e[1:] + e[:-1]) / 2
def run(name, lat, lon):
    plan = engine.Plan(lat, lat_zm, 50)
    plan.set_tem(nlev, nt, plev * 100)
    f = engine.synth_fields(0, lat, lon, plev, nt, dtype=torch.float64, seed=0)
    q = engine.synth_fields(0, lat, lon, plev, nt, dtype=torch.float64, seed=5)[2]
    out = {}
    for what, fn in (("tem_run", lambda: plan.tem_run(*f)), ("tem_tracer_run", lambda: plan.tem_tracer_run(*f, q))):
        for _ in range(2):
            fn()
        plan.kernel_timing(True)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            fn()
        b.record(); torch.cuda.synchronize()
        sw, _ = plan.kernel_timing_read(0)
        plan.kernel_timing(False)
        out[what] = (a.elapsed_time(b) / 5, sw)
    print("%s: N=%d mode=%d one_pass=%s | tem_run %.3f ms (sweep %.3f) | tem_tracer_run %.3f ms (sweep %.3f)"
          % (name, lat.size, plan.sweep_mode, plan.one_pass, *out["tem_run"], *out["tem_tracer_run"]), flush=True)
    plan.close()
lat, lon = synth.cubed_sphere_gll(120)
run("cubed sphere ne120", lat, lon)
nl = 777600 // 32
xs = (np.arange(nl) + 0.5) / nl * 89.9
lat = np.repeat(np.concatenate([xs, -xs]), 16)
lon = np.tile(np.arange(16) * 22.5, 2 * nl)
run("lat-lon 16 per latitude", lat, lon)
