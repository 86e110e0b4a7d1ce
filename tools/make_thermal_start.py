#!/usr/bin/env python3
"""Makes the thermalised start state of bench.py --start thermal (tests/golden/thermal_start_n32_d8.npz): the configs[2] box built with
synth.make_system(32, 8, seed 0, liquid_like=True) (rigid molecules at the force field's equilibrium geometry), relaxed on the GPU by
blocks of NVE steps with a velocity rescale to 300 K in between, until the kinetic temperature stays near 300 K without help, the planted
tree still gives 8 diabats and a 250-step NVE probe keeps them and rebuilds its Verlet list at a liquid's pace.  Runs on the GPU box:
    python tools/make_thermal_start.py OUT.npz [max_blocks] [block_len]
The result is DATA (positions, velocities of a synthetic system); it only fixes the initial condition of a benchmark line."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from reactive_pb_nn_md_amd import api, forcefield, synth, system
out = sys.argv[1]
max_blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 60
blen = int(sys.argv[3]) if len(sys.argv) > 3 else 20
KB = 0.0083144626
s = synth.make_system(32, n_diabat=8, seed=0, max_chain=3, liquid_like=True)
p = forcefield.build_params(forcefield.ch3so3h_water_forcefield(), moltype_order=["h2o", "so3h"])
spec = system.from_synth(s, p, rc=9.0, rv=10.5, alpha=0.3, pme_grid=64, max_chain=3)
h = api.MsEvbHip(spec)
h.set_velocities(synth.maxwell_boltzmann(spec.mass, 300.0, 12345))
h.ms_evb_calculate_total_force_energy(want_force=False)
ndf = 3 * spec.n_atom - 3
best = None
for b in range(max_blocks):
    en = h.md_integrate_atomic(blen, spec.delta_t, use_evb=True)
    T = 2 * en[-1, 1] / (ndf * KB)
    D = h.evb_state()["D"]
    print("block %3d  T %8.1f K  PE %.5e  D %d" % (b, T, en[-1, 0], D), flush=True)
    v = h.get_velocities()
    h.set_velocities(v * np.sqrt(300.0 / max(T, 1.0)))
    if b >= 8 and D == 8 and abs(T - 300.0) < 25.0:      # candidate: probe it
        x0, v0 = h.get_positions(), h.get_velocities()
        hp = api.MsEvbHip(system.from_synth(s, p, rc=9.0, rv=10.5, alpha=0.3, pme_grid=64, max_chain=3))
        hp.set_positions(x0); hp.set_velocities(v0)
        hp.ms_evb_calculate_total_force_energy(want_force=False)
        ok, Ts = True, []
        for k in range(25):
            e = hp.md_integrate_atomic(10, spec.delta_t, use_evb=True)
            Ts.append(2 * e[-1, 1] / (ndf * KB))
            if hp.evb_state()["D"] != 8:
                ok = False; break
        reb = int(hp.phase_times().get("verlet list construction", (0, 0))[1])
        hp.close()
        print("   probe: %s, D kept for %d steps, T %.0f..%.0f K, %d rebuilds in 250 steps" % ("OK" if ok else "no", 10 * len(Ts), min(Ts), max(Ts), reb), flush=True)
        if ok and reb <= 8 and max(Ts) < 400.0:
            best = (x0, v0, b, reb, Ts)
            break
if best is None:
    print("no state passed the probe"); sys.exit(1)
x0, v0, b, reb, Ts = best
np.savez_compressed(out, xyz=x0, velocity=v0, gen=np.array([32, 8, 0]), blocks=np.array([b, blen]), probe_rebuilds=np.array(reb), probe_T=np.array(Ts))
print("saved", out, os.path.getsize(out) / 1e6, "MB")
h.close()
