#!/usr/bin/env python3
"""RCCL with MORE THAN ONE rank, checked cheaply: the MS-EVB force call of a small reference fixture (evb_n6_d8_s1: 8 diabats) sharded over the
ranks - diabats dealt to the ranks, Hamiltonian entries all-gathered, partial Hellmann-Feynman forces all-reduced (csrc/comm.hip) - against the
same call on one rank and against the reference's numbers.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/rccl_two_rank_check.py

Rank r uses GPU r.  Rank 0 prints ONE JSON line: {"ranks_seen": R, "max_abs_dF_vs_single_rank": ..., "abs_dE_vs_single_rank": ..., "max_abs_dH": ...,
"max_abs_dF_vs_reference": ..., "steps_same_trajectory": ..., "ok": true|false}; exit status 0 only when ok.  Needs R visible GPUs (R = WORLD_SIZE)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    from reactive_pb_nn_md_amd import api, system
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    dist.init_process_group("gloo")      # rendezvous + the id broadcast only: the data path under test is RCCL inside the library
    fixture = os.environ.get("MSEVB_CHECK_FIXTURE", "evb_n6_d8_s1")
    g = np.load(os.path.join(ROOT, "tests", "golden", fixture + ".npz"))
    spec = system.from_golden(g)
    # single-rank values: every rank computes them on its own GPU first (no communicator)
    h1 = api.MsEvbHip(spec, device=local)
    e1, F1 = h1.ms_evb_calculate_total_force_energy()
    H1 = h1.evb_state()["H"].copy()
    h1.close()
    # the same call on `world` ranks
    h = api.MsEvbHip(spec, device=local)
    idbuf = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        idbuf = torch.from_numpy(np.frombuffer(api.MsEvbHip.comm_unique_id(h.L), dtype=np.uint8).copy())
    dist.broadcast(idbuf, 0)
    h.comm_init(idbuf.numpy().tobytes(), rank, world)
    e, F = h.ms_evb_calculate_total_force_energy()
    st = h.evb_state()
    en = h.md_integrate_atomic(5, float(g["md_dt"][0]) if "md_dt" in g else 0.0005, use_evb=True)      # a few steps: both collectives every step
    x_end = h.get_positions()
    h.close()
    # every rank must hold the same bytes (the trajectories are replicated)
    t = torch.from_numpy(np.concatenate([F.ravel(), x_end.ravel(), [e]]))
    tmax, tmin = t.clone(), t.clone()
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX); dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
    identical = bool(torch.equal(tmax, tmin))
    ranks = torch.ones(1); dist.all_reduce(ranks)
    out = {"ranks_seen": int(ranks.item()), "fixture": fixture, "n_diabats": int(st["D"]),
           "max_abs_dF_vs_single_rank": float(np.abs(F - F1).max()), "abs_dE_vs_single_rank": float(abs(e - e1)), "max_abs_dH": float(np.abs(st["H"] - H1).max()),
           "max_abs_dF_vs_reference": float(np.abs(F - g["F_adiabatic"]).max()) if "F_adiabatic" in g else None,
           "replicas_bit_identical_after_5_steps": identical, "energies_finite": bool(np.all(np.isfinite(en)))}
    out["ok"] = bool(out["ranks_seen"] == world and out["max_abs_dF_vs_single_rank"] < 1e-9 and out["abs_dE_vs_single_rank"] < 1e-9 * max(1.0, abs(e1)) and out["max_abs_dH"] < 1e-9 * max(1.0, float(np.abs(H1).max()))
                     and identical and out["energies_finite"])
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)
    return 0 if out["ok"] else 1


if __name__ == "__main__":
    sys.exit(main())
