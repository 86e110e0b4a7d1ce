#!/usr/bin/env python3
"""How hot does the synthetic box run, and what does a thermalised one cost per step?  Blocks of MD with velocity rescaling to 300 K
between blocks (the library itself is NVE, like the reference), then a timed NVE stretch.
    python tools/equilibrate.py [liquid_like 0/1] [blocks] [block_len]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from reactive_pb_nn_md_amd import api, forcefield, synth, system
liquid = bool(int(sys.argv[1])) if len(sys.argv) > 1 else False
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 40
blen = int(sys.argv[3]) if len(sys.argv) > 3 else 50
vscale = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0      # > 1: provoke a blow-up (must end in an error, not a GPU fault)
KB = 0.0083144626
s = synth.make_system(32, n_diabat=8, seed=0, max_chain=3, liquid_like=liquid)
p = forcefield.build_params(forcefield.ch3so3h_water_forcefield(), moltype_order=["h2o", "so3h"])
spec = system.from_synth(s, p, rc=9.0, rv=10.5, alpha=0.3, pme_grid=64, max_chain=3)
h = api.MsEvbHip(spec)
h.set_velocities(vscale * synth.maxwell_boltzmann(spec.mass, 300.0, 12345))
h.ms_evb_calculate_total_force_energy(want_force=False)
ndf = 3 * spec.n_atom - 3
for b in range(blocks):
    h.L.msevb_reset_phase_times(h.ctx)
    t0 = time.time()
    try:
        en = h.md_integrate_atomic(blen, spec.delta_t, use_evb=True)
    except Exception as ex:
        print("ERROR:", ex)
        x, v, f = h.get_positions(), h.get_velocities(), h.get_forces()
        L = spec.box[0] if np.ndim(spec.box) == 1 else spec.box[0][0]
        bad = np.unique(np.nonzero(~np.isfinite(x) | (x < -L) | (x > 2 * L))[0])
        print("box", L, "atoms with bad coordinates:", len(bad), bad[:20])
        big = np.argsort(-np.abs(np.nan_to_num(v, nan=1e30)).max(1))[:10]
        for a in big:
            print("atom %6d mol %6d  x %s  v %s  f %s" % (a, np.searchsorted(spec.mol_first, a, side="right") - 1, x[a], v[a], f[a]))
        print("hydronium", h.topology()["hydronium_molecule"], h.evb_state()["D"])
        sys.exit(1)
    dt = time.time() - t0
    T = 2 * en[-1, 1] / (ndf * KB)      # columns: potential, kinetic
    st = h.evb_state()
    reb = int(h.phase_times().get("verlet list construction", (0, 0))[1])
    print("block %3d  T %8.1f K  PE %.4e  D %2d  rebuilds %2d/%d  %.0f steps/s" % (b, T, en[-1, 0], st["D"], reb, blen, blen / dt), flush=True)
    if b < blocks - 6:      # the last blocks run NVE
        v = h.get_velocities()
        h.set_velocities(v * np.sqrt(300.0 / max(T, 1.0)))
h.close()
