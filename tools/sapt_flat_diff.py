#!/usr/bin/env python3
"""Diagnostic: real-space forces of the flat-list pair kernel against the row kernel on a system with a live SAPT term; prints the atoms
that differ most and their types:  python tools/sapt_flat_diff.py [n_side]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from reactive_pb_nn_md_amd import api, forcefield, synth, system
n = int(sys.argv[1]) if len(sys.argv) > 1 else 17
s = synth.make_system(n, n_diabat=2, seed=7)
ff = forcefield.with_test_sapt(forcefield.ch3so3h_water_forcefield())
p = forcefield.build_params(ff, moltype_order=["h2o", "so3h"])
spec = system.from_synth(s, p, rc=9.0, rv=10.5, pme_grid=32)
out = {}
for name, opts in (("rows", {"pair_flat": 0}), ("flat", {}), ("flat_notaylor_rows", {"pair_taylor": 0})):
    h = api.MsEvbHip(spec)
    for k, v in opts.items():
        h.set_option(k, v)
    h.construct_verlet_list()
    h.zero_forces()
    e = h.real_space_energy_force()
    out[name] = (np.array(e), h.get_forces().copy(), h.info("pair_kernel"))
    h.close()
F0 = out["rows"][1]
for name in ("flat", "flat_notaylor_rows"):
    e, F, k = out[name]
    d = np.abs(F - F0).max(axis=1)
    print(name, "kernel", k, "energies", e - out["rows"][0], "max |dF|", d.max(), "atoms above 1e-9:", int((d > 1e-9).sum()), "of", len(d))
    idx = np.argsort(-d)[:12]
    at = np.asarray(spec.atype)
    for i in idx:
        print("  atom %6d type %2d mol %6d  dF %.3e  F %s" % (i, at[i], np.asarray(spec.mol_of)[i] if hasattr(spec, "mol_of") else -1, d[i], np.array2string(F0[i], precision=4)))
    print("  types of the atoms above 1e-9:", np.unique(at[d > 1e-9], return_counts=True))
