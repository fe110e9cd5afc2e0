#!/usr/bin/env python3
"""Development aid: TEM + N tracers on the single-sweep path -- one tracer per sweep (temx_tracer_run each) against two
per sweep (temx_tracers_run: (q1, q2, v, omega) read once).   tracers_bench.py 120x72x30 [f64|f32] [ntracers=2]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pytemdiags_amd import engine, synth

ne, nlev, nt = (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "30x72x30").split("x"))
dt = torch.float32 if len(sys.argv) > 2 and sys.argv[2] == "f32" else torch.float64
ntr = int(sys.argv[3]) if len(sys.argv) > 3 else 2
lat, lon = synth.cubed_sphere_gll(ne, mirror=False)
plev = synth.pressure_levels(nlev)
e = np.arange(-90, 91, 1.0); lat_zm = (e[1:] + e[:-1]) / 2
f = engine.synth_fields(0, lat, lon, plev, nt, dtype=dt, seed=0)
qs = [engine.synth_fields(0, lat, lon, plev, nt, dtype=dt, seed=5 + i)[2] for i in range(ntr)]     # T-like fields as tracers
plan = engine.Plan(lat, lat_zm, 50)
plan.set_tem(nlev, nt, plev * 100)
out = plan._alloc_results(False)


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


tem = timeit(lambda: plan.tem_run(*f, out=out))
one = timeit(lambda: (plan.tem_run(*f, out=out), [plan.tracer_run(q, f[1], f[3]) for q in qs]))
two = timeit(lambda: (plan.tem_run(*f, out=out), plan.tracers_run(qs, f[1], f[3])))
bad = plan.status()
print("ne%dx%dx%d %s single_sweep=%s: TEM %.3f ms | TEM + %d tracers: one per sweep %.3f ms (%.3f per tracer), two per sweep %.3f ms "
      "(%.3f per tracer) nonfinite=%s" % (ne, nlev, nt, "f32" if dt == torch.float32 else "f64", plan.single_sweep, tem, ntr, one,
                                         (one - tem) / ntr, two, (two - tem) / ntr, bad), flush=True)
plan.close()
