#!/usr/bin/env python3
"""Diagnostic: the thermalised configs[2] box in short blocks, printing energy, temperature, diabat count and hydronium molecule per block.
   python tools/liquid_md_diag.py [steps=800] [block=10] [opt=value ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from reactive_pb_nn_md_amd import api, forcefield, synth, system
N = int(sys.argv[1]) if len(sys.argv) > 1 else 800
B = int(sys.argv[2]) if len(sys.argv) > 2 else 10
KB = 0.0083144626
s = synth.make_system(32, n_diabat=8, seed=0, max_chain=3, liquid_like=True)
p = forcefield.build_params(forcefield.ch3so3h_water_forcefield(), moltype_order=["h2o", "so3h"])
spec = system.from_synth(s, p, rc=9.0, rv=10.5, alpha=0.3, pme_grid=64, max_chain=3)
h = api.MsEvbHip(spec)
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    h.set_option(k, float(v))
h.set_velocities(synth.maxwell_boltzmann(spec.mass, 300.0, 12345))
h.ms_evb_calculate_total_force_energy(want_force=False)
ndf = 3 * spec.n_atom - 3
for _b in range(8):
    en = h.md_integrate_atomic(20, spec.delta_t, use_evb=True)
    T = 2 * en[-1, 1] / (ndf * KB)
    h.set_velocities(h.get_velocities() * np.sqrt(300.0 / max(T, 1.0)))
h.ms_evb_calculate_total_force_energy(want_force=False)
done = 0
prev = None
while done < N:
    try:
        tr = h.md_integrate_atomic(B, spec.delta_t, use_evb=True)
    except Exception as ex:
        print("step", done, "FAILED:", str(ex)[:400]); break
    done += B
    st = h.evb_state()
    top = h.topology()
    v = h.get_velocities(); F = h.get_forces()
    vmax = np.abs(v).max(); ivm = int(np.abs(v).max(axis=1).argmax()); fmax = np.abs(F).max(); ifm = int(np.abs(F).max(axis=1).argmax())
    key = (int(st["D"]), int(top["hydronium_molecule"]), int(st["principle_diabat"]))
    print("step %4d  Epot %.2f KE %.2f Etot %.2f  T %.1f  D %d hyd %d principal %d  |v|max %.1f (atom %d)  |F|max %.1f (atom %d)%s" % (done, tr[-1, 0], tr[-1, 1], tr[-1].sum(), 2 * tr[-1, 1] / (ndf * KB), key[0], key[1], key[2], vmax, ivm, fmax, ifm, "   <-- changed" if prev is not None and key != prev else ""), flush=True)
    prev = key
h.close()
