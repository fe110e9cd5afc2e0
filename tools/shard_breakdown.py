#!/usr/bin/env python3
"""Development aid: where one rank's step of the ncol-sharded job goes (stage by stage, HIP events):
shard_breakdown.py 120x72x30 8 [rank]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pytemdiags_amd import engine, synth, sharding, _lib

ne, nlev, nt = (int(x) for x in sys.argv[1].split("x"))
world = int(sys.argv[2])
r = int(sys.argv[3]) if len(sys.argv) > 3 else 0
lat, lon = synth.cubed_sphere_gll(ne)
plev = synth.pressure_levels(nlev)
lat_zm = (np.arange(-90, 91, 1.0)[1:] + np.arange(-90, 91, 1.0)[:-1]) / 2
pg = engine.Plan(lat, lat_zm, 50)
G = pg.matrix(_lib.MAT_GRAM).cpu().numpy()
pg.close()
mine = sharding.symmetric_ncol_shards(lat, world)[r]
plan = engine.Plan(lat[mine], lat_zm, 50, defer_finalize=True)
plan.finalize(G)
plan.set_tem(nlev, nt, plev * 100)
f = engine.synth_fields(0, lat[mine], lon[mine], plev, nt)
for _ in range(3):
    B4 = plan.tem_stage1(*f); B3 = plan.tem_stage2_from_sums(B4) if plan.one_pass else plan.tem_stage2(*f, B4); plan.tem_stage3(B3)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
acc = np.zeros(3)
reps = 20
plan.kernel_timing(True)
for _ in range(reps):
    ev[0].record(); B4 = plan.tem_stage1(*f)
    ev[1].record(); B3 = plan.tem_stage2_from_sums(B4) if plan.one_pass else plan.tem_stage2(*f, B4)
    ev[2].record(); plan.tem_stage3(B3)
    ev[3].record(); torch.cuda.synchronize()
    acc += [ev[i].elapsed_time(ev[i + 1]) for i in range(3)]
acc /= reps
p_ms, _ = plan.kernel_timing_read(0); e_ms, _ = plan.kernel_timing_read(1)
print("rank %d/%d: %d cols, mode %d one_pass %s | stage1 %.3f (sweep %.3f) stage2 %.3f (flux/eddy %.3f) stage3 %.3f | total %.3f ms"
      % (r, world, mine.size, plan.sweep_mode, plan.one_pass, acc[0], p_ms, acc[1], e_ms, acc[2], acc.sum()), flush=True)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
out = plan._alloc_results(False)
a.record()
for _ in range(reps):
    plan.tem_run(*f, out=out)
b.record(); torch.cuda.synchronize()
print("tem_run back to back: %.3f ms" % (a.elapsed_time(b) / reps))
