#!/usr/bin/env python3
"""Cost of a principal-diabat switch for a host that re-packs its atom arrays, at BASELINE.json configs[2] size.

The reference's evb_change_diabat_data_structure_topology (src/ms_evb.f90:843) moves the transferring proton's array slot and every atom in between;
the Fortran shim then hands the new host order to the library (msevb_set_host_order, fortran/total_energy_forces_hip.f90:msevb_hip_remap) instead of
tearing the context down.  Timed here through the same C entry points: the per-step host traffic of the Fortran path (positions up, MS-EVB call,
forces down) on an ordinary step, and the same with a fresh host order in front - against a context re-creation, which is what round 2 did.

    python tools/hop_cost.py
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from reactive_pb_nn_md_amd import api, forcefield, synth, system  # noqa: E402


def main():
    s = synth.make_system(32, n_diabat=8, seed=0, max_chain=3)
    p = forcefield.build_params(forcefield.ch3so3h_water_forcefield(), moltype_order=["h2o", "so3h"])
    spec = system.from_synth(s, p, rc=9.0, rv=10.5, alpha=0.3, pme_grid=64, max_chain=3)
    t0 = time.perf_counter()
    h = api.MsEvbHip(spec)
    h.construct_verlet_list()
    h.ms_evb_calculate_total_force_energy()
    t_create = time.perf_counter() - t0
    N = spec.n_atom
    xyz = h.get_positions()

    def step(order=None):
        t = time.perf_counter()
        if order is not None:
            h.set_host_order(order)
            x = np.empty_like(xyz); x[:] = xyz[order]      # the host's arrays in its new order
        else:
            x = xyz
        h.set_positions(x)
        e, F = h.ms_evb_calculate_total_force_energy()
        return time.perf_counter() - t, e, F

    for _ in range(5):
        step()
    plain = min(step()[0] for _ in range(20))
    e0, F0 = step()[1:]
    rng = np.random.default_rng(3)
    hops = []
    for k in range(10):
        # a hop as the reference re-packs it: one atom leaves slot a and is inserted at slot b, everything in between shifts by one
        a, b = sorted(rng.integers(0, N, 2))
        order = np.concatenate([np.arange(0, a), np.arange(a + 1, b + 1), [a], np.arange(b + 1, N)]).astype(np.int64)
        dt, e, F = step(order)
        hops.append(dt)
        back = np.empty(N, dtype=np.int64); back[order] = np.arange(N)
        assert abs(e - e0) < 1e-9 * abs(e0) and np.abs(F[back] - F0).max() < 1e-9, "forces in the new host order differ"
        h.set_host_order(None)
    print("context creation + first list build + first call: %.1f ms" % (t_create * 1e3))
    print("ordinary step through the host interface (positions up, MS-EVB call, forces down): %.3f ms" % (plain * 1e3))
    print("first hop of the run (allocates the order array on the device): %.3f ms above an ordinary step" % ((hops[0] - plain) * 1e3))
    hops = hops[1:]
    print("step with a new host atom order in front (msevb_set_host_order + the host's own re-pack of its coordinates): %.3f ms  (hop cost %.3f ms; slowest of the other 9: %.3f ms)"
          % (np.median(hops) * 1e3, (np.median(hops) - plain) * 1e3, (max(hops) - plain) * 1e3))
    h.close()


if __name__ == "__main__":
    main()
