#!/usr/bin/env python3
"""SURVEY config C on one GPU (69^3 sites -> ~985k atoms, 128^3 grid, 16 planted diabats): does everything scale (24-bit list entries,
LDS tiles, batched FFT of 15 grids), do the two schedules agree, what does a step cost."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from reactive_pb_nn_md_amd import api, forcefield, synth, system
t0 = time.time()
s = synth.make_system(69, n_diabat=16, seed=0, max_chain=4)
p = forcefield.build_params(forcefield.ch3so3h_water_forcefield(), moltype_order=["h2o", "so3h"])
spec = system.from_synth(s, p, rc=9.0, rv=10.5, alpha=0.3, pme_grid=128, max_chain=4)
print("system: %d atoms, box %.1f A, built in %.1f s" % (spec.n_atom, spec.box[0], time.time() - t0), flush=True)
h = api.MsEvbHip(spec)
e1, F1 = h.ms_evb_calculate_total_force_energy()
st = h.evb_state()
print("D = %d, E = %.6f, |F|max = %.3f" % (st["D"], e1, np.abs(F1).max()), flush=True)
h.set_option("overlap", 0); h.set_option("graphs", 0); h.set_option("pair_prune", 0)
e0, F0 = h.ms_evb_calculate_total_force_energy()
print("serial schedule: dE/E = %.2e, dF = %.2e" % (abs(e0 - e1) / abs(e1), np.abs(F0 - F1).max()), flush=True)
h.set_option("overlap", 1); h.set_option("graphs", 1); h.set_option("pair_prune", 1)
h.set_velocities(synth.maxwell_boltzmann(spec.mass, 300.0, 1))
h.ms_evb_calculate_total_force_energy(want_force=False)
h.md_integrate_atomic(10, spec.delta_t, use_evb=True)
h.synchronize()
t0 = time.time()
en = h.md_integrate_atomic(50, spec.delta_t, use_evb=True)
h.synchronize()
dt = time.time() - t0
print("50 steps: %.2f ms/step = %.1f steps/s; E_tot first/last %.3f %.3f; D now %d" % (dt / 50 * 1e3, 50 / dt, en[0].sum(), en[-1].sum(), h.evb_state()["D"]), flush=True)
h.close()
