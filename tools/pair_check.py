#!/usr/bin/env python3
"""The real-space stage on the bench system (98 k atoms) through the pair kernels the library has, against each other:
k_pair_half (pair_flat=0), k_pair_flat on a list it has just written / on the list of the call before (prune_freeze), and its
one-table-row variant (pair_taylor).  Prints energies, max |dF| and the time per stage call."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reactive_pb_nn_md_amd import api, forcefield, synth, system

n_side = int(sys.argv[1]) if len(sys.argv) > 1 else 32
s = synth.make_system(n_side, n_diabat=8, seed=0, max_chain=3)
p = forcefield.build_params(forcefield.ch3so3h_water_forcefield(), moltype_order=["h2o", "so3h"])
spec = system.from_synth(s, p, rc=9.0, rv=10.5, alpha=0.3, pme_grid=64, max_chain=3)
h = api.MsEvbHip(spec)
h.construct_verlet_list()

def run(opts, reps=30):
    for k, v in opts.items():
        h.set_option(k, float(v))
    h.zero_forces(); e = h.real_space_energy_force(); F = h.get_forces().copy()
    if opts.get("prune_freeze"):      # second call: the list of the first
        h.zero_forces(); e = h.real_space_energy_force(); F = h.get_forces().copy()
    h.synchronize(); t0 = time.time()
    for _ in range(reps):
        h.real_space_energy_force()
    h.synchronize()
    return np.array(e[:2]), F, (time.time() - t0) / reps * 1e3

base = run({"pair_flat": 0, "prune_freeze": 0, "pair_taylor": 0})
print("k_pair_half                          E_elec %.9f E_vdw %.9f   %.3f ms/call" % (base[0][0], base[0][1], base[2]))
for split in [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else "0,1,2".split(","))]:
    for name, o in (("list rewritten", {"prune_freeze": 0, "pair_taylor": 1}), ("list kept", {"prune_freeze": 1, "pair_taylor": 1}), ("list kept, two table rows", {"prune_freeze": 1, "pair_taylor": 0})):
        o = dict(o, pair_flat=1, pair_split=split)
        r = run(o)
        print("flat split=%d %-24s E_elec %.9f E_vdw %.9f   %.3f ms/call   dE %.2e %.2e   max|dF| %.3e (max|F| %.1f)" %
              (split, name, r[0][0], r[0][1], r[2], r[0][0] - base[0][0], r[0][1] - base[0][1], np.abs(r[1] - base[1]).max(), np.abs(base[1]).max()))
