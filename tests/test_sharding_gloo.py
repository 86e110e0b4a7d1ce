"""world_size-2 `gloo` test of the diabat sharding used on N>1 GPUs: rank r owns diabats d>=2 with
(d-2) mod R == r, slots are laid out [rank][slot], one all-gather completes the store, every rank blends.  The per-element
numbers come from the CPU oracle here (no GPU in this container); the layout logic under test is bench.py's / comm.hip's."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, golden


def _worker(rank, world, port, name, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import torch.distributed as dist
    import oracle_py
    from reactive_pb_nn_md_amd import sharding, system
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    spec = system.from_golden(g)
    o = oracle_py.Oracle(spec)
    o.verlet_build()
    r = o.evb()                     # every rank could compute everything; keep only what this rank owns
    D, N = r["D"], spec.n_atom
    S = sharding.slots_per_rank(D, world)
    Hloc = np.zeros((S, 2))
    Floc = np.zeros((S, 2, N, 3))
    for d in range(1, D):
        if sharding.owner(d, world) == rank:
            s = sharding.local_slot(d, world)
            Hloc[s] = (r["H"][d, d], r["H"][r["parent"][d], d])
            Floc[s, 0], Floc[s, 1] = r["F_diag"][d], r["F_off"][d]
    Hall = [torch.zeros(S, 2, dtype=torch.float64) for _ in range(world)]
    Fall = [torch.zeros(S, 2, N, 3, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(Hall, torch.from_numpy(Hloc))
    dist.all_gather(Fall, torch.from_numpy(Floc))
    H = np.zeros((D, D))
    H[0, 0] = r["H"][0, 0]
    Fd, Fo = np.zeros((D, N, 3)), np.zeros((D, N, 3))
    Fd[0] = r["F_diag"][0]
    for d in range(1, D):
        ow, s = sharding.owner(d, world), sharding.local_slot(d, world)
        H[d, d], H[r["parent"][d], d] = Hall[ow][s].numpy()
        Fd[d], Fo[d] = Fall[ow][s, 0].numpy(), Fall[ow][s, 1].numpy()
    E, F, c = sharding.diagonalise_and_blend(H, r["parent"], Fd, Fo)
    q.put((rank, float(E), float(np.abs(F - r["F"]).max()), float(abs(E - r["E"])), float(np.abs(H - r["H"]).max())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["evb_n6_d8_s1", "evb_n6_d4_s2"])
def test_two_rank_allgather_reproduces_single_rank(name):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, name, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
    g = golden(name)
    for rank, E, dF, dE, dH in res:
        assert dH == 0.0 and dE < 1e-9 and dF < 1e-9
        assert abs(E - g["E_adiabatic"][0]) < 1e-9 * abs(E)
