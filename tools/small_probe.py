#!/usr/bin/env python3
"""Small-workload probe (ne30 x 72 x 1): stream launches vs HIP-graph replay (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pytemdiags_amd import engine, synth

ne, nlev, nt = (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "30x72x1").split("x"))
reps = 200
lat, lon = synth.cubed_sphere_gll(ne)
plev = synth.pressure_levels(nlev)
lat_zm = (np.arange(-90, 91, 1.0)[1:] + np.arange(-90, 91, 1.0)[:-1]) / 2
plan = engine.Plan(lat, lat_zm, 50)
plan.set_tem(nlev, nt, plev * 100)
f = engine.synth_fields(0, lat, lon, plev, nt)
out = plan._alloc_results(False)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3):
        plan.tem_run(*f, out=out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        plan.tem_run(*f, out=out)
    b.record(); torch.cuda.synchronize()
    ms_stream = a.elapsed_time(b) / reps
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        plan.tem_run(*f, out=out)
    g.replay(); torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        g.replay()
    b.record(); torch.cuda.synchronize()
    ms_graph = a.elapsed_time(b) / reps
pts = lat.size * nlev * nt
print("ne%dx%dx%d paired=%s: stream %.4f ms (%.3g pts/s, %.1f%%) | graph %.4f ms (%.3g pts/s, %.1f%%)" % (
    ne, nlev, nt, plan.paired, ms_stream, pts / ms_stream * 1e3, pts / ms_stream * 1e3 / 7e8,
    ms_graph, pts / ms_graph * 1e3, pts / ms_graph * 1e3 / 7e8))
