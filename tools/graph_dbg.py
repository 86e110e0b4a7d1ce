#!/usr/bin/env python3
"""diagnostic: a small MS-EVB box, one force call + a few MD steps with the step graph on; run under AMD_LOG_LEVEL=3 to see the HIP calls of the capture"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from reactive_pb_nn_md_amd import api, forcefield, synth, system
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
grid = int(sys.argv[2]) if len(sys.argv) > 2 else 32
s = synth.make_system(n, n_diabat=4, seed=3)
p = forcefield.build_params(forcefield.ch3so3h_water_forcefield(), moltype_order=["h2o", "so3h"])
spec = system.from_synth(s, p, rc=7.0, rv=8.0, pme_grid=grid, nslist=16)
h = api.MsEvbHip(spec)
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    h.set_option(k, float(v))
h.set_velocities(synth.maxwell_boltzmann(spec.mass, 300.0, 5))
e, F = h.ms_evb_calculate_total_force_energy()
print("call 1 done", e, flush=True)
sys.stderr.write("=== MD steps start\n"); sys.stderr.flush()
en = h.md_integrate_atomic(4, spec.delta_t, use_evb=True)
print("md done", en[:, 0], flush=True)
for k in ("graph_launches", "graph_captures", "graph_mismatches", "spec_steps", "spec_misses", "evb_calls"):
    print(k, h.info(k))
h.close()
