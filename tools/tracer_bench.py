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

# TEM + one tracer: the two runs one after the other against the fused sweep (temx_tem_tracer_run)
os.environ.pop("TEMX_TWO_PASS", None)
plan = engine.Plan(lat, lat_zm, 50)
plan.set_tem(nlev, nt, plev * 100)
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
sep = timeit(lambda: (plan.tem_run(*f), plan.tracer_run(q, f[1], f[3])))
plan.kernel_timing(True)
fus = timeit(lambda: plan.tem_tracer_run(*f, q))
sw, n = plan.kernel_timing_read(0)
print("ne%dx%dx%d TEM + one tracer: separate runs %.3f ms, one fused sweep %.3f ms (its sweep %.3f ms = %.2f TB/s of the 5 "
      "fields); one_pass=%s" % (ne, nlev, nt, sep, fus, sw or float("nan"), 5 * 8 * lat.size * nlev * nt / (sw or 1) / 1e9, plan.one_pass), flush=True)
plan.close()

# the class-sum forms against the single-sweep forms (TEM run, then the tracer run)
for form, env in (("class-sum forms", "0"), ("single-sweep forms", "1")):
    os.environ["TEMX_SINGLE_SWEEP"] = env
    plan = engine.Plan(lat, lat_zm, 50)
    plan.set_tem(nlev, nt, plev * 100)
    plan.tem_run(*f)
    t_tr = timeit(lambda: plan.tracer_run(q, f[1], f[3]))
    t_all = timeit(lambda: plan.tem_tracer_run(*f, q))
    print("ne%dx%dx%d %s (single_sweep=%s): tracer_run %.3f ms, tem_tracer_run %.3f ms" % (ne, nlev, nt, form, plan.single_sweep, t_tr, t_all), flush=True)
    plan.close()
