"""Development aid: does the placement of the four field arrays matter for the single sweep's time?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pytemdiags_amd import engine, synth
ne, nlev, nt = 120, 72, 30
lat, lon = synth.cubed_sphere_gll(ne)
plev = synth.pressure_levels(nlev)
e = np.arange(-90, 91, 1.0); lat_zm = (e[1:] + e[:-1]) / 2
def sweep_ms(plan, f, reps=6):
    for _ in range(2):
        plan.tem_run(*f)
    torch.cuda.synchronize()
    plan.kernel_timing(True)
    for _ in range(reps):
        plan.tem_run(*f)
    torch.cuda.synchronize()
    sw, n = plan.kernel_timing_read(0)
    plan.kernel_timing(False)
    return sw
plan = engine.Plan(lat, lat_zm, 50)
plan.set_tem(nlev, nt, plev * 100)
f = engine.synth_fields(0, lat, lon, plev, nt, dtype=torch.float64, seed=0)
print("fields after the plan, separate tensors: %.3f ms" % sweep_ms(plan, f), [hex(x.data_ptr()) for x in f], flush=True)
big = torch.empty((4,) + tuple(f[0].shape), dtype=torch.float64, device="cuda:0")
for i in range(4):
    big[i].copy_(f[i])
g = [big[i] for i in range(4)]
print("one block of four: %.3f ms" % sweep_ms(plan, g), [hex(x.data_ptr()) for x in g], flush=True)
print("separate again: %.3f ms" % sweep_ms(plan, f), flush=True)
# the same field four times (one quarter of the footprint, the same bytes per launch)
print("one array four times: %.3f ms" % sweep_ms(plan, [f[0]] * 4), flush=True)
del big, g
torch.cuda.empty_cache()
pad = torch.empty(3 * 2**30 + 12345 * 512, dtype=torch.uint8, device="cuda:0")
h = [x.clone() for x in f]
print("clones after a 3 GiB odd-sized pad: %.3f ms" % sweep_ms(plan, h), [hex(x.data_ptr()) for x in h], flush=True)
plan.close()
plan = engine.Plan(lat, lat_zm, 50)
plan.set_tem(nlev, nt, plev * 100)
print("a new plan (built after the fields): %.3f ms" % sweep_ms(plan, f), flush=True)
