#!/usr/bin/env python3
"""Long NVE run of the thermalised configs[2] box (the bench's `thermalised_box` protocol, then N steps in blocks): robustness of the
device-resident MS-EVB loop over list rebuilds, diabat-count changes, tree changes (speculative misses) and principal-diabat switches,
and the energy drift.   python tools/long_liquid_md.py [steps=20000] [block=500] [n_exclusions=2]
(n_exclusions = 3: 1-4 pairs excluded, src/intra_bonded_interactions.f90 [ exclusions ] generation - the acid proton then does not see its own sulfonate oxygens)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from reactive_pb_nn_md_amd import api, forcefield, synth, system
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 500
NEX = int(sys.argv[3]) if len(sys.argv) > 3 else 2
KB = 0.0083144626
s = synth.make_system(32, n_diabat=8, seed=0, max_chain=3, liquid_like=True)
p = forcefield.build_params(forcefield.ch3so3h_water_forcefield(), moltype_order=["h2o", "so3h"], n_exclusions=NEX)
spec = system.from_synth(s, p, rc=9.0, rv=10.5, alpha=0.3, pme_grid=64, max_chain=3)
h = api.MsEvbHip(spec)
h.set_velocities(synth.maxwell_boltzmann(spec.mass, 300.0, 12345))
h.ms_evb_calculate_total_force_energy(want_force=False)
ndf = 3 * spec.n_atom - 3
for _b in range(8):
    en = h.md_integrate_atomic(20, spec.delta_t, use_evb=True)
    T = 2 * en[-1, 1] / (ndf * KB)
    h.set_velocities(h.get_velocities() * np.sqrt(300.0 / max(T, 1.0)))
h.ms_evb_calculate_total_force_energy(want_force=False)
hyd0 = h.topology()["hydronium_molecule"]
e0 = None
done, switches, dset = 0, 0, set()
t0 = time.time()
print("%d atoms, dt %.2f fs, blocks of %d steps" % (spec.n_atom, spec.delta_t * 1e3, B))
while done < N:
    try:
        tr = h.md_integrate_atomic(B, spec.delta_t, use_evb=True)
    except api.MsevbError as err:
        print("run ended after %d + < %d steps: %s" % (done, B, str(err)[:200]), flush=True)
        break
    done += B
    e = tr.sum(1)
    if e0 is None:
        e0 = e[0]
    st = h.evb_state()
    hyd = h.topology()["hydronium_molecule"]
    if hyd != hyd0:
        switches += 1; hyd0 = hyd
    dset.add(int(st["D"]))
    assert np.all(np.isfinite(tr))
    if done % (4 * B) == 0 or done >= N:
        print("step %6d  Etot %.2f  (E - E0)/|E0| %+.2e  T %.1f K  D %d  hydronium molecule %d" % (done, e[-1], (e[-1] - e0) / abs(e0), 2 * tr[-1, 1] / (ndf * KB), st["D"], hyd), flush=True)
dt = time.time() - t0
nh, log = h.hop_log() if hasattr(h, "hop_log") else (0, None)
print("%d steps in %.1f s = %.0f steps/s incl. the host's per-block bookkeeping; principal-diabat switches %d; diabat counts seen %s; relative energy change %+.2e" % (done, dt, done / dt, switches, sorted(dset), (e[-1] - e0) / abs(e0)))
h.close()
