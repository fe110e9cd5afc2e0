#!/usr/bin/env python3
"""Development aid: worst field-normalised error of the ten outputs against the CPU oracle for the
forms of the class path (one pass / two passes) and between the two, on one shape.
  parity_probe.py 120x72x2 [f64|f32]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import tem_oracle as orc
from pytemdiags_amd import _lib, engine, synth

ne, nlev, nt = (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "30x72x4").split("x"))
td = torch.float32 if len(sys.argv) > 2 and sys.argv[2] == "f32" else torch.float64
lat, lon = synth.cubed_sphere_gll(ne)
plev = synth.pressure_levels(nlev)
f = engine.synth_fields(0, lat, lon, plev, nt, dtype=td, seed=0)
ref = orc.TEMOracle(*[x.cpu().numpy() for x in f], lat, plev, mode="factorised")
out = {}
for form in ("one", "two"):
    os.environ.pop("TEMX_ONE_PASS", None); os.environ.pop("TEMX_TWO_PASS", None)
    os.environ["TEMX_ONE_PASS" if form == "one" else "TEMX_TWO_PASS"] = "1"
    plan = engine.Plan(lat, ref.lat, 50)
    plan.set_tem(nlev, nt, plev * 100)
    res, _ = plan.tem_run(*f)
    assert not plan.status()
    out[form] = (res.cpu().numpy(), plan.one_pass)
    plan.close()
for form in ("one", "two"):
    res, op = out[form]
    errs = {n: float(np.max(np.abs(res[i] - getattr(ref, n)().astype(np.float64))) / np.max(np.abs(getattr(ref, n)())))
            for i, n in enumerate(_lib.RESULT_NAMES)}
    w = max(errs, key=errs.get)
    print("ne%dx%dx%d %s one_pass=%s vs oracle: worst %.3e (%s)" % (ne, nlev, nt, form, op, errs[w], w), flush=True)
a, b = out["one"][0], out["two"][0]
d = {n: float(np.max(np.abs(a[i] - b[i])) / np.max(np.abs(b[i]))) for i, n in enumerate(_lib.RESULT_NAMES)}
w = max(d, key=d.get)
print("one-pass vs two-pass: worst %.3e (%s)" % (d[w], w))
