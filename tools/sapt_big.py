#!/usr/bin/env python3
"""config B with the live SAPT / Tang-Toennies test parameters (k_pair<true, ...>): schedules agree, MD runs, step cost."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from reactive_pb_nn_md_amd import api, forcefield, synth, system
s = synth.make_system(32, n_diabat=8, seed=0)
p = forcefield.build_params(forcefield.with_test_sapt(forcefield.ch3so3h_water_forcefield()), moltype_order=["h2o", "so3h"])
spec = system.from_synth(s, p, rc=9.0, rv=10.5, alpha=0.3, pme_grid=64)
h = api.MsEvbHip(spec)
e1, F1 = h.ms_evb_calculate_total_force_energy()
h.set_option("overlap", 0); h.set_option("graphs", 0); h.set_option("pair_prune", 0)
e0, F0 = h.ms_evb_calculate_total_force_energy()
print("D %d  E %.6f  serial/unpruned schedule: dE/E %.2e dF %.2e" % (h.evb_state()["D"], e1, abs(e0 - e1) / abs(e1), np.abs(F0 - F1).max()), flush=True)
h.set_option("overlap", 1); h.set_option("graphs", 1); h.set_option("pair_prune", 1)
h.set_velocities(synth.maxwell_boltzmann(spec.mass, 300.0, 1))
h.ms_evb_calculate_total_force_energy(want_force=False)
h.md_integrate_atomic(10, spec.delta_t, use_evb=True); h.synchronize()
t0 = time.time(); en = h.md_integrate_atomic(100, spec.delta_t, use_evb=True); h.synchronize(); dt = time.time() - t0
print("100 steps: %.3f ms/step = %.0f steps/s; E_tot %.2f -> %.2f" % (dt * 10, 100 / dt, en[0].sum(), en[-1].sum()), flush=True)
h.close()
