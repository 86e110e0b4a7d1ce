import sys, time, os
sys.path.insert(0, "/root/repo")
import numpy as np
from reactive_pb_nn_md_amd import api, forcefield, synth, system
s = synth.make_system(32, n_diabat=8, seed=0)
p = forcefield.build_params(forcefield.ch3so3h_water_forcefield(), moltype_order=["h2o", "so3h"])
spec = system.from_synth(s, p, rc=9.0, rv=10.5, alpha=0.3, pme_grid=64)
h = api.MsEvbHip(spec)
h.set_option("overlap", 0); h.set_option("graphs", 0); h.set_option("phase_timing", 1); h.set_option("phase_timing_stride", 1)
h.calculate_total_force_energy(want_force=False)
h.L.msevb_reset_phase_times(h.ctx)
for _ in range(20): h.calculate_total_force_energy(want_force=False)
print({k: round(v[0], 4) for k, v in h.phase_times().items() if v[1] > 0})
