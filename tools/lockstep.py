#!/usr/bin/env python3
"""GPU library and CPU oracle stepped side by side on a small EVB box: first step where diabat count, hydronium or energies part.
    python tools/lockstep.py n_diabat seed liquid_like proton_shift n_steps [T]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ctypes as C
import numpy as np
import oracle_py
from reactive_pb_nn_md_amd import api, forcefield, synth, system
nd, seed, liquid, shift, n = int(sys.argv[1]), int(sys.argv[2]), bool(int(sys.argv[3])), float(sys.argv[4]), int(sys.argv[5])
T = float(sys.argv[6]) if len(sys.argv) > 6 else 300.0
p = forcefield.build_params(forcefield.ch3so3h_water_forcefield(), moltype_order=["h2o", "so3h"])
s = synth.make_system(6, n_diabat=nd, seed=seed, liquid_like=liquid, proton_shift=shift)
spec = system.from_synth(s, p, rc=7.0, rv=8.0, pme_grid=16, nslist=16)
o = oracle_py.Oracle(spec, threads=8)
vel = synth.maxwell_boltzmann(spec.mass, T, seed + 100).copy()
o.verlet_build()
r = o.evb(); F = r["F"].copy()
o.L.orc_evb_apply_principal(o.h)
h = api.MsEvbHip(spec)
h.set_velocities(vel)
e, Fg = h.ms_evb_calculate_total_force_energy()
print("step 0: oracle E %.9f D %d newh %d | gpu E %.9f D %d hyd %d | dF %.2e" % (r["E"], r["D"], r["new_hydronium"], e, h.evb_state()["D"], h.topology()["hydronium_molecule"], np.abs(Fg - F).max()))
bad = 0
for st in range(1, n + 1):
    eo = o.md_step(vel, F, spec.delta_t, evb=True)
    Do = o.L.orc_evb_n_diabat(o.h)
    nh = C.c_int(0); o.L.orc_evb_principal(o.h, C.byref(nh))
    en = h.md_integrate_atomic(1, spec.delta_t, use_evb=True)
    stg = h.evb_state(); hyd = h.topology()["hydronium_molecule"]
    dF = np.abs(h.get_forces() - F).max(); dx = np.abs(h.get_positions() - o.get_xyz()).max()
    flag = Do != stg["D"] or (nh.value + 1) != hyd or abs(en[0, 0] - eo) > 1e-6 * abs(eo)
    if flag or st % 25 == 0 or st > n - 3:
        print("step %4d: oracle E %.9f D %d hyd %d | gpu E %.9f D %d hyd %d | dF %.2e dx %.2e %s" % (st, eo, Do, nh.value + 1, en[0, 0], stg["D"], hyd, dF, dx, "<-- DIFF" if flag else ""), flush=True)
    if flag:
        bad += 1
        if bad >= 4:
            break
