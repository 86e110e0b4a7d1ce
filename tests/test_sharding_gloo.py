"""world_size-2 `gloo` test of the diabat sharding used on N>1 GPUs: rank r owns diabats d>=2 with
(d-2) mod R == r; the Hamiltonian entries are all-gathered (2 doubles per slot), every rank diagonalises the same matrix, blends ITS
diabats and the partial forces are all-reduced.  The per-element numbers come from the CPU oracle here (no GPU in this container);
the protocol under test is comm.hip's / evb.hip's (blend_terms)."""
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
    for d in range(1, D):
        if sharding.owner(d, world) == rank:
            Hloc[sharding.local_slot(d, world)] = (r["H"][d, d], r["H"][r["parent"][d], d])
    Hall = [torch.zeros(S, 2, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(Hall, torch.from_numpy(Hloc))                    # msevb_comm_allgather_evb
    H = np.zeros((D, D))
    H[0, 0] = r["H"][0, 0]
    for d in range(1, D):
        ow, s = sharding.owner(d, world), sharding.local_slot(d, world)
        H[d, d], H[r["parent"][d], d] = Hall[ow][s].numpy()
    E, c = sharding.ground_state(H)
    # this rank only has ITS stores: zero the others to prove they are not needed
    Fd, Fo = np.zeros((D, N, 3)), np.zeros((D, N, 3))
    for d in range(D):
        if (d == 0 and rank == 0) or (d > 0 and sharding.owner(d, world) == rank):
            Fd[d], Fo[d] = r["F_diag"][d], r["F_off"][d]
    Fp = torch.from_numpy(sharding.partial_blend(c, r["parent"], Fd, Fo, rank, world))
    dist.all_reduce(Fp)                                               # msevb_comm_allreduce_forces
    F = Fp.numpy()
    q.put((rank, float(E), float(np.abs(F - r["F"]).max()), float(abs(E - r["E"])), float(np.abs(H - r["H"]).max())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["evb_n6_d8_s1", "evb_n6_d4_s2"])
def test_two_rank_exchange_reproduces_single_rank(name):
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
