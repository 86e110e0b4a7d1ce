#!/usr/bin/env python3
"""Thermalise the liquid-like box (velocity rescaling, as tools/equilibrate.py), then time single steps and print the diabat count of
each: which transitions of the diabat count cost host time?   python tools/step_times.py [blocks_before] [single_steps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from reactive_pb_nn_md_amd import api, forcefield, synth, system
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 150
KB = 0.0083144626
s = synth.make_system(32, n_diabat=8, seed=0, max_chain=3, liquid_like=True)
p = forcefield.build_params(forcefield.ch3so3h_water_forcefield(), moltype_order=["h2o", "so3h"])
spec = system.from_synth(s, p, rc=9.0, rv=10.5, alpha=0.3, pme_grid=64, max_chain=3)
h = api.MsEvbHip(spec)
h.set_velocities(synth.maxwell_boltzmann(spec.mass, 300.0, 12345))
h.ms_evb_calculate_total_force_energy(want_force=False)
ndf = 3 * spec.n_atom - 3
for b in range(nb):
    en = h.md_integrate_atomic(50, spec.delta_t, use_evb=True)
    T = 2 * en[-1, 1] / (ndf * KB)
    h.set_velocities(h.get_velocities() * np.sqrt(300.0 / max(T, 1.0)))
Dp = h.evb_state()["D"]
rows = []
for i in range(ns):
    h.synchronize()
    t0 = time.perf_counter()
    h.md_integrate_atomic(1, spec.delta_t, use_evb=True)
    h.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    D = h.evb_state()["D"]
    rows.append((i, dt, Dp, D))
    Dp = D
med = np.median([r[1] for r in rows])
print("median single-step call %.3f ms" % med)
for i, dt, a, b in rows:
    if a != b or dt > 1.5 * med:
        print("step %4d  %.3f ms  D %d -> %d" % (i, dt, a, b))
print(" ".join("%d:%.2f" % (r[3], r[1]) for r in rows))
if os.environ.get("MSEVB_STAMPS"):
    h.close()
