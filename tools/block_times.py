#!/usr/bin/env python3
"""Per-block wall time of the default bench workload (blocks of 10 steps): where the 200-step run loses time against the first 100 steps."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reactive_pb_nn_md_amd import api, forcefield, synth, system

NS = int(os.environ.get("BT_NSIDE", "32")); ND = int(os.environ.get("BT_DIABATS", "8")); KG = int(os.environ.get("BT_GRID", "64"))
EVB = os.environ.get("BT_EVB", "1") == "1"
s = synth.make_system(NS, n_diabat=ND, seed=0, max_chain=3)
p = forcefield.build_params(forcefield.ch3so3h_water_forcefield(), moltype_order=["h2o", "so3h"])
spec = system.from_synth(s, p, rc=9.0, rv=10.5, alpha=0.3, pme_grid=KG, max_chain=3)
h = api.MsEvbHip(spec)
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    h.set_option(k, float(v))
h.set_velocities(synth.maxwell_boltzmann(spec.mass, 300.0, 12345))
(h.ms_evb_calculate_total_force_energy if EVB else h.calculate_total_force_energy)(want_force=False)
h.md_integrate_atomic(20, spec.delta_t, use_evb=EVB)
for b in range(22):
    h.synchronize()
    t0 = time.perf_counter()
    en = h.md_integrate_atomic(10, spec.delta_t, use_evb=EVB)
    h.synchronize()
    dt = time.perf_counter() - t0
    st = h.evb_state() if EVB else {"D": 1}
    ke = h.calculate_kinetic_energy()
    print("steps %3d-%3d  %.3f ms/step  D=%d  T=%.0f K" % (20 + 10 * b, 30 + 10 * b, dt * 100.0, st["D"], ke / (1.5 * 0.0083144626 * spec.n_atom)), flush=True)
