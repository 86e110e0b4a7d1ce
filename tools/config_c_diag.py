#!/usr/bin/env python3
"""config C (985k atoms, 128^3): GPU principal-diabat force call vs the CPU oracle, term by term."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import oracle_py
from reactive_pb_nn_md_amd import api, forcefield, synth, system
n_side = int(sys.argv[1]) if len(sys.argv) > 1 else 69
K = int(sys.argv[2]) if len(sys.argv) > 2 else 128
s = synth.make_system(n_side, n_diabat=16 if n_side > 40 else 8, seed=0, max_chain=4)
p = forcefield.build_params(forcefield.ch3so3h_water_forcefield(), moltype_order=["h2o", "so3h"])
spec = system.from_synth(s, p, rc=9.0, rv=10.5, alpha=0.3, pme_grid=K, max_chain=4)
h = api.MsEvbHip(spec)
h.construct_verlet_list()
h.zero_forces(); e_rs = h.real_space_energy_force(); F_rs = h.get_forces()
h.zero_forces(); e_pme = h.pme_reciprocal_space_energy_force(); F_pme = h.get_forces()
h.zero_forces(); e_b = h.intra_molecular_energy_force(); F_b = h.get_forces()
vp, nl = h.get_verlet_list()
o = oracle_py.Oracle(spec, threads=min(64, os.cpu_count() or 8))
t0 = time.time(); o.verlet_build(); print("oracle verlet %.1f s" % (time.time() - t0), flush=True)
vpo, nlo = o.verlet_get()
print("list sizes", len(nl), len(nlo), "verlet_point equal", np.array_equal(vp, vpo), flush=True)
if np.array_equal(vp, vpo):
    same = all(np.array_equal(np.sort(nl[vp[i]-1:vp[i+1]-1]), np.sort(nlo[vpo[i]-1:vpo[i+1]-1])) for i in range(0, spec.n_atom, 1))
    print("pair set identical:", same, flush=True)
else:
    d = np.nonzero(np.diff(vp) != np.diff(vpo))[0]
    print("rows differing:", len(d), d[:10], flush=True)
    for i in d[:5]:
        a, b = set(nl[vp[i]-1:vp[i+1]-1]), set(nlo[vpo[i]-1:vpo[i+1]-1])
        for j in (a ^ b):
            dr = spec.xyz[i] - spec.xyz[j - 1]; dr -= spec.box * np.round(dr / spec.box)
            print("  atom", i + 1, "partner", j, "in gpu" if j in a else "in oracle", "r = %.15f" % np.sqrt((dr * dr).sum()))
Ep, Fp = o.pair(); Ei, Fi = o.intra_nonbonded(); Er, Fr = o.pme(); Eb, Fb = o.bonded()
print("pair   elec %.3e vdw %.3e | dF %.3e" % (e_rs[0] - Ep[0], e_rs[1] - Ep[1], np.abs(F_rs - Fp - Fi).max()))
print("intra  elec %.3e vdw %.3e" % (e_rs[2] - Ei[0], e_rs[3] - Ei[1]))
print("recip  %.3e (%.6f) | dF %.3e" % (e_pme - Er[0], Er[0], np.abs(F_pme - Fr).max()))
print("bonded %.3e %.3e %.3e | dF %.3e" % (e_b[0] - Eb[2], e_b[1] - Eb[3], e_b[2] - Eb[4], np.abs(F_b - Fb).max()))
print("raw", e_rs, Ep, Ei, Eb)
