#!/usr/bin/env python3
"""Times one stage call of the library on the bench system without integrating (positions stay put), so experimental kernel
variants that produce wrong forces can still be timed:  python tools/time_stage.py [pair|pme|total] [--opt k=v ...]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reactive_pb_nn_md_amd import api, forcefield, synth, system

ap = argparse.ArgumentParser()
ap.add_argument("stage", nargs="?", default="pair")
ap.add_argument("--n-side", type=int, default=32)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--opt", action="append", default=[])
a = ap.parse_args()
s = synth.make_system(a.n_side, n_diabat=8, seed=0, max_chain=3)
p = forcefield.build_params(forcefield.ch3so3h_water_forcefield(), moltype_order=["h2o", "so3h"])
spec = system.from_synth(s, p, rc=9.0, rv=10.5, alpha=0.3, pme_grid=64, max_chain=3)
h = api.MsEvbHip(spec)
for kv in a.opt:
    k, v = kv.split("=")
    h.set_option(k, float(v))
h.construct_verlet_list()
fn = {"pair": h.real_space_energy_force, "pme": h.pme_reciprocal_space_energy_force, "total": lambda: h.calculate_total_force_energy(want_force=False)}[a.stage]
fn()
h.set_option("phase_timing", 1.0)
t0 = time.time()
for _ in range(a.reps):
    fn()
h.synchronize()
dt = (time.time() - t0) / a.reps
print("stage %s: wall %.3f ms/call; phases %s" % (a.stage, dt * 1e3, {k: round(v[0] / max(v[1], 1), 4) for k, v in h.phase_times().items() if v[1]}))
