import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pytemdiags_amd import engine, synth
lat, lon = synth.cubed_sphere_gll(int(sys.argv[1]))
e = np.arange(-90, 91, 1.0)
plan = engine.Plan(lat, (e[1:] + e[:-1]) / 2, 50)
print("plan built; single_sweep", plan.single_sweep)
