#!/usr/bin/env python3
"""Time one rank's share of an ncol-sharded job on one GPU (development aid):
shard_probe.py 120x72x30 8  -> ranks 0 and 7 of 8, latitude-class shards."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pytemdiags_amd import engine, synth, sharding

ne, nlev, nt = (int(x) for x in sys.argv[1].split("x"))
world = int(sys.argv[2])
dtype = torch.float32 if len(sys.argv) > 3 and sys.argv[3] == "f32" else torch.float64
lat, lon = synth.cubed_sphere_gll(ne)
plev = synth.pressure_levels(nlev)
lat_zm = (np.arange(-90, 91, 1.0)[1:] + np.arange(-90, 91, 1.0)[:-1]) / 2
parts = sharding.symmetric_ncol_shards(lat, world)
for r in sorted({0, world // 2, world - 1}):
    mine = parts[r]
    plan = engine.Plan(lat[mine], lat_zm, 50, defer_finalize=True)
    G = np.eye(51)                      # timing only: any SPD matrix finalises the plan
    plan.finalize(G)
    plan.set_tem(nlev, nt, plev * 100)
    f = engine.synth_fields(0, lat[mine], lon[mine], plev, nt, dtype=dtype)
    out = plan._alloc_results(False)
    for _ in range(2):
        plan.tem_run(*f, out=out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        plan.tem_run(*f, out=out)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    pts = mine.size * nlev * nt
    print("rank %d/%d: %d cols, mode %d, %.3f ms -> %.3g pts/s per GPU" % (r, world, mine.size, plan.sweep_mode, ms, pts / ms * 1e3), flush=True)
    plan.close()
    del f, out
