#!/usr/bin/env python3
"""Soak run for the intermittent MSEVB_ERR_NUMERIC of round 3: N fresh contexts at BASELINE.json configs[2] size
(98 220 atoms, 8 diabats, 64^3), each doing what the two runs that failed had been doing - the MS-EVB force call with the MFMA spreading
option (tests: test_mfma_charge_spreading_config_B) and the force call + partial blend of one of eight virtual ranks
(tools/virtual_rank_times.py) - plus the default schedule with a short trajectory.  Every error is logged with its message (which since
round 4 names the atom, its position / velocity / force and the kernel that flagged it) and the iteration's mode; nothing stops the run.

    python tools/soak.py [iterations] [out.json]
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from reactive_pb_nn_md_amd import api, forcefield, synth, system  # noqa: E402


def main():
    n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    out = sys.argv[2] if len(sys.argv) > 2 else None
    s = synth.make_system(32, n_diabat=8, seed=0, max_chain=3)
    p = forcefield.build_params(forcefield.ch3so3h_water_forcefield(), moltype_order=["h2o", "so3h"])
    spec = system.from_synth(s, p, rc=9.0, rv=10.5, alpha=0.3, pme_grid=64, max_chain=3)
    vel = synth.maxwell_boltzmann(spec.mass, 300.0, 12345)
    h = api.MsEvbHip(spec)
    e_ref, F_ref = h.ms_evb_calculate_total_force_energy()
    ev = h.evb_state()["eigvec"].copy()
    h.close()
    # round 5: the linear step graphs (captures, replays, a tree change on the way) and the device path of a rank with a communicator on a virtual rank of 4
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
    import virtual_rank_times as vrt
    vrt.OPTS = []
    pay4 = vrt.gathered_payloads(spec, 4)
    modes = ["default + 12 MD steps", "spread_mfma", "virtual rank of 8", "gather_mfma + stored weights", "step_graph + 40 MD steps", "virtual rank of 4, device path"]      # spread_mfma: + 12 MD steps as well
    errors, counts, t0 = [], {m: 0 for m in modes}, time.time()
    for it in range(n_it):
        mode = modes[it % len(modes)]
        counts[mode] += 1
        try:
            h = api.MsEvbHip(spec)
            if mode == "spread_mfma":
                h.set_option("spread_mfma", 1)
            elif mode == "gather_mfma + stored weights":
                h.set_option("gather_mfma", 1)
                h.set_option("pme_weightless", 0)
            elif mode == "virtual rank of 8":
                h.set_option("shard_principal", 1)
                h.set_option("evb_device_solve_ranks", 0)
                h.comm_set_virtual((it // len(modes)) % 8, 8)
            elif mode == "step_graph + 40 MD steps":
                h.set_option("step_graph", 1)
            elif mode == "virtual rank of 4, device path":
                h.set_option("shard_principal", 1)
                h.comm_set_virtual((it // len(modes)) % 4, 4)
                h.comm_set_virtual_payload(pay4)
            for _ in range(3):
                e, F = h.ms_evb_calculate_total_force_energy()
                if mode == "virtual rank of 8":
                    h.evb_partial_blend(ev, download=False)
                elif mode == "virtual rank of 4, device path":
                    evr = h.evb_state()["eigvec"]
                    if min(np.abs(evr - ev).max(), np.abs(evr + ev).max()) > 1e-9 or not np.all(np.isfinite(F)):
                        errors.append({"iteration": it, "mode": mode, "error": "eigenvector of the gathered Hamiltonian differs: %.3e" % min(np.abs(evr - ev).max(), np.abs(evr + ev).max())})
                elif abs(e - e_ref) > 1e-9 * abs(e_ref) or not np.all(np.isfinite(F)) or np.abs(F - F_ref).max() > 1e-7:
                    errors.append({"iteration": it, "mode": mode, "error": "result differs: dE %.3e max dF %.3e" % (e - e_ref, np.abs(F - F_ref).max())})
            if mode in ("default + 12 MD steps", "spread_mfma", "step_graph + 40 MD steps"):      # (the round-3 failure of the spread_mfma test came in its 25-step trajectory)
                h.set_velocities(vel)
                en = h.md_integrate_atomic(40 if mode.startswith("step_graph") else 12, spec.delta_t, use_evb=True)
                if mode.startswith("step_graph") and (h.info("graph_launches") < 20 or h.info("graph_mismatches") > 0):
                    errors.append({"iteration": it, "mode": mode, "error": "graph launches %d, mismatches %d" % (h.info("graph_launches"), h.info("graph_mismatches"))})
                if not np.all(np.isfinite(en)):
                    errors.append({"iteration": it, "mode": mode, "error": "non-finite trajectory energies"})
            h.close()
        except api.MsevbError as ex:
            errors.append({"iteration": it, "mode": mode, "error": str(ex)})
            try:
                h.close()
            except Exception:
                pass
        if (it + 1) % 10 == 0:
            print("soak: %d / %d iterations, %d errors, %.0f s" % (it + 1, n_it, len(errors), time.time() - t0), flush=True)
    rec = {"workload": "98220 atoms, 8 diabats, 64^3 - fresh context per iteration, 3 MS-EVB force calls each (+ 12 MD steps in the default mode)",
           "iterations": n_it, "per_mode": counts, "errors": errors, "seconds": round(time.time() - t0, 1)}
    print(json.dumps(rec))
    if out:
        with open(out, "w") as f:
            json.dump(rec, f, indent=1)
    return 1 if errors else 0


if __name__ == "__main__":
    sys.exit(main())
