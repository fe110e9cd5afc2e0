#!/usr/bin/env python3
"""Development aid: run one workload through temx_tem_run a few times (for rocprofv3 --kernel-trace).
   run_shape.py ne30x72x91 [f64|f32] [form] [steps]     form: auto | two-pass | class-sums | single-sweep"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pytemdiags_amd import engine, synth

ne, nlev, nt = (int(x) for x in sys.argv[1].lower().replace("ne", "").split("x"))
dt = torch.float32 if len(sys.argv) > 2 and sys.argv[2] == "f32" else torch.float64
form = sys.argv[3] if len(sys.argv) > 3 else "auto"
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
lat, lon = synth.cubed_sphere_gll(ne, mirror=False)
plev = synth.pressure_levels(nlev)
lat_zm = (np.arange(-90, 91, 1.0)[1:] + np.arange(-90, 91, 1.0)[:-1]) / 2
plan = engine.Plan(lat, lat_zm, 50, form=None if form == "auto" else form)
plan.set_tem(nlev, nt, plev * 100)
f = engine.synth_fields(0, lat, lon, plev, nt, dtype=dt)
out = plan._alloc_results(False)
for _ in range(2):
    plan.tem_run(*f, out=out)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(steps):
    plan.tem_run(*f, out=out)
b.record()
torch.cuda.synchronize()
print("%s %s form=%s single_sweep=%s one_pass=%s: %.4f ms/step" % (sys.argv[1], "f32" if dt == torch.float32 else "f64", form,
                                                                 plan.single_sweep, plan.one_pass, a.elapsed_time(b) / steps), flush=True)
