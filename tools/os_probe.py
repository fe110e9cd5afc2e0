#!/usr/bin/env python3
"""Development aid: the single-sweep form of temx_tem_run (TEMX_SINGLE_SWEEP=1) against the class-sum path:
parity per result and timing.  os_probe.py 30x72x8 [oracle]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pytemdiags_amd import engine, synth, _lib

ne, nlev, nt = (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "30x72x8").split("x"))
lat, lon = synth.cubed_sphere_gll(ne)
plev = synth.pressure_levels(nlev)
e = np.arange(-90, 91, 1.0); lat_zm = (e[1:] + e[:-1]) / 2
f = engine.synth_fields(0, lat, lon, plev, nt, dtype=torch.float64, seed=0)
if os.environ.get("PROBE_RANDOM") == "1":       # does the data matter for the sweep's time?
    g = torch.Generator(device="cuda:0"); g.manual_seed(1)
    for i, (b, a) in enumerate(((10.0, 20.0), (-3.0, 10.0), (250.0, 30.0), (0.05, 0.2))):
        f[i].uniform_(b - a / 2, b + a / 2, generator=g)
os.environ["TEMX_ONE_PASS"] = "1"
out = {}
for form in ("csum", "single"):
    os.environ["TEMX_SINGLE_SWEEP"] = "1" if form == "single" else "0"
    plan = engine.Plan(lat, lat_zm, 50)
    plan.set_tem(nlev, nt, plev * 100)
    res, zon = plan.tem_run(*f, want_zonal=True)
    bad = plan.status()
    for _ in range(2):
        plan.tem_run(*f)
    torch.cuda.synchronize()
    plan.kernel_timing(True)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    a.record()
    for _ in range(reps):
        plan.tem_run(*f)
    b.record(); torch.cuda.synchronize()
    sw, _ = plan.kernel_timing_read(0); fl, _ = plan.kernel_timing_read(1)
    print("%-6s one_pass=%s nonfinite=%s: %.3f ms per run (sweep %.3f, flux / contraction %.3f)" % (form, plan.one_pass, bad, a.elapsed_time(b) / reps, sw or 0, fl or 0), flush=True)
    out[form] = (res.cpu().numpy(), zon.cpu().numpy())
    plan.close()
fn = lambda x, r: float(np.max(np.abs(x - r)) / np.max(np.abs(r)))
for i, n in enumerate(_lib.RESULT_NAMES):
    print("  result %-10s single vs csum %.2e" % (n, fn(out["single"][0][i], out["csum"][0][i])))
for i, n in enumerate(_lib.ZONAL_NAMES[:7]):
    print("  zonal  %-10s single vs csum %.2e" % (n, fn(out["single"][1][i], out["csum"][1][i])))
if len(sys.argv) > 2:
    from oracle import tem_oracle as orc
    ref = orc.TEMOracle(*[x.cpu().numpy() for x in f], lat, plev, mode="factorised")
    for form in out:
        w = max(fn(out[form][0][i], np.asarray(getattr(ref, n)(), float)) for i, n in enumerate(_lib.RESULT_NAMES))
        print("  %-6s vs oracle: worst of the ten results %.2e" % (form, w))
