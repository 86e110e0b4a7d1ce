#!/usr/bin/env python3
"""Longer EVB MD on the bench system: robustness (list rebuilds, principal-diabat switches, graph replays)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from reactive_pb_nn_md_amd import api, forcefield, synth, system
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
s = synth.make_system(32, n_diabat=8, seed=0, max_chain=3)
p = forcefield.build_params(forcefield.ch3so3h_water_forcefield(), moltype_order=["h2o", "so3h"])
spec = system.from_synth(s, p, rc=9.0, rv=10.5, alpha=0.3, pme_grid=64, max_chain=3)
h = api.MsEvbHip(spec)
h.set_velocities(synth.maxwell_boltzmann(spec.mass, 300.0, 12345))
h.ms_evb_calculate_total_force_energy(want_force=False)
hyd0 = h.topology()["hydronium_molecule"]
t0 = time.time()
done = 0
switches = 0
while done < n:
    tr = h.md_integrate_atomic(50, spec.delta_t, use_evb=True)
    done += 50
    st = h.evb_state()
    hyd = h.topology()["hydronium_molecule"]
    if hyd != hyd0:
        switches += 1; hyd0 = hyd
    print("step %4d  Etot %.1f  D %d  principal %d  hydronium %d" % (done, tr[-1].sum(), st["D"], st["principle_diabat"], hyd), flush=True)
print("%.1f steps/s, %d hydronium changes seen at block boundaries" % (n / (time.time() - t0), switches))
