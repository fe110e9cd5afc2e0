#!/usr/bin/env python3
"""Where does the error of test_weights_mode_tem_pipeline_on_a_class_grid come from?  (the
test holds the ten results to 1e-9, the seven zonal means to 1e-10.)  Same 64 x 16 Gaussian grid and fields;
per quantity: GPU vs oracle, and -- the ten results recomputed by the ORACLE's epilogue from the GPU's seven
zonal means -- how much of a result's error is the epilogue amplifying the (tiny) differences of its inputs.
  weighted_probe.py [L=30]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import tem_oracle as orc
from pytemdiags_amd import _lib, engine, synth

L = int(sys.argv[1]) if len(sys.argv) > 1 else 30
nlat, nlon, nlev, nt = 64, 16, 9, 8
xg, wg = np.polynomial.legendre.leggauss(nlat)
lat = np.repeat(np.rad2deg(np.arcsin(xg)), nlon)
lon = np.tile(np.arange(nlon) * (360.0 / nlon), nlat)
w = np.repeat(wg / (2.0 * nlon), nlon)
plev = synth.pressure_levels(nlev)
f = synth.analytic_fields(lat, lon, plev, nt, seed=4)
ref = orc.TEMOracle(*f, lat, plev, L=L, zm_dlat=3, weights=w.copy())
plan = engine.Plan(lat, ref.lat, L, defer_finalize=True)
plan.set_weights(w)
plan.set_tem(nlev, nt, plev * 100)
res, zon = plan.tem_run(*[torch.as_tensor(x, device="cuda:0") for x in f], want_zonal=True)
res, zon = res.cpu().numpy(), zon.cpu().numpy()
fn = lambda x, r: float(np.max(np.abs(np.asarray(x, float) - np.asarray(r, float))) / np.max(np.abs(r)))
print("L = %d, grid %d x %d, weights mode (unfused second sweep)" % (L, nlat, nlon))
for i, n in enumerate(_lib.ZONAL_NAMES):
    print("  zonal  %-16s GPU vs oracle %.2e" % (n, fn(zon[i], getattr(ref, n))))
epi = orc.TEMOracle.from_zonal_means({n: zon[i] for i, n in enumerate(_lib.ZONAL_NAMES[:7])}, plev, zm_dlat=3)
wz = max(fn(zon[i], getattr(ref, n)) for i, n in enumerate(_lib.ZONAL_NAMES[:7]))
for i, n in enumerate(_lib.RESULT_NAMES):
    e_gpu, e_epi, e_gap = fn(res[i], getattr(ref, n)()), fn(getattr(epi, n)(), getattr(ref, n)()), fn(res[i], getattr(epi, n)())
    print("  result %-10s GPU vs oracle %.2e | oracle epilogue on the GPU's zonal means vs oracle %.2e (x %.0f the "
          "worst zonal-mean error) | GPU epilogue vs oracle epilogue on the same inputs %.2e" % (n, e_gpu, e_epi, e_epi / wz, e_gap))
