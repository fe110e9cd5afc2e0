"""Development aid: configs[2]-like workload (ne240 x 128 lev x 1 snapshot, fp32) -- time per run and per kernel slot."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pytemdiags_amd import engine, synth
ne, nlev, nt = (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "240x128x1").split("x"))
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
lat, lon = synth.cubed_sphere_gll(ne)
plev = synth.pressure_levels(nlev)
e = np.arange(-90, 91, 1.0); lat_zm = (e[1:] + e[:-1]) / 2
f = engine.synth_fields(0, lat, lon, plev, nt, dtype=torch.float32, seed=0)
plan = engine.Plan(lat, lat_zm, 50)
plan.set_tem(nlev, nt, plev * 100)
for _ in range(3):
    plan.tem_run(*f)
torch.cuda.synchronize()
plan.kernel_timing(True)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(reps):
    plan.tem_run(*f)
b.record(); torch.cuda.synchronize()
sw, _ = plan.kernel_timing_read(0); fl, _ = plan.kernel_timing_read(1)
print("ne%dx%dx%d f32: %.3f ms per run (sweep %.3f, flux %.3f); single_sweep=%s one_pass=%s" %
      (ne, nlev, nt, a.elapsed_time(b) / reps, sw or 0, fl or 0, plan.single_sweep, plan.one_pass), flush=True)
