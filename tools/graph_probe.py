#!/usr/bin/env python3
"""Development aid: one step of a small shape eager vs replayed from a HIP graph (graph_probe.py ne30x72x1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pytemdiags_amd import engine, synth

ne, nlev, nt = (int(x) for x in sys.argv[1].lower().replace("ne", "").split("x"))
lat, lon = synth.cubed_sphere_gll(ne, mirror=False)
plev = synth.pressure_levels(nlev)
lat_zm = (np.arange(-90, 91, 1.0)[1:] + np.arange(-90, 91, 1.0)[:-1]) / 2
plan = engine.Plan(lat, lat_zm, 50)
plan.set_tem(nlev, nt, plev * 100)
f = engine.synth_fields(0, lat, lon, plev, nt)
out = plan._alloc_results(False)


def timed(fn, reps=400):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


print("eager   %.1f us/step" % timed(lambda: plan.tem_run(*f, out=out)), flush=True)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    plan.tem_run(*f, out=out)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        plan.tem_run(*f, out=out)
print("graph   %.1f us/step" % timed(g.replay), flush=True)
