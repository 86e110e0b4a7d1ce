"""world_size-2 `gloo` test of the diabat sharding used on N>1 GPUs: rank r owns diabats d>=2 with
(d-2) mod R == r; the Hamiltonian entries are all-gathered (2 doubles per slot), every rank diagonalises the same matrix, blends ITS
diabats and the partial forces are all-reduced.  The per-element numbers come from the CPU oracle here (no GPU in this container);
the payload packing under test is the library's own (csrc/comm.hip, host functions), the blend layout evb.hip's (blend_terms)."""
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
    # the library's OWN payload code (csrc/comm.hip: msevb_evb_payload_pack / _unpack, what msevb_comm_allgather_evb wraps around its
    # ncclAllGather) - host functions of libmsevb_hip.so, no GPU needed; gloo carries the bytes here
    import ctypes as C
    from reactive_pb_nn_md_amd import api
    L = api.load_library()
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
    P = L.msevb_evb_payload_size(D, world)
    assert P == 2 * sharding.slots_per_rank(D, world) + 9
    parent = np.ascontiguousarray(r["parent"], dtype=np.int32)
    Hmine = np.zeros((D, D))                                           # what THIS rank computed: its own diabats' entries, H'_11
    Hmine[0, 0] = r["H"][0, 0]
    for d in range(1, D):
        if sharding.owner(d, world) == rank:
            Hmine[d, d], Hmine[parent[d], d] = r["H"][d, d], r["H"][parent[d], d]
    e8 = np.arange(8, dtype=np.float64) * (rank + 1)                   # stand-ins for the principal energy slots: summed over the ranks
    mine = np.zeros(P)
    assert L.msevb_evb_payload_pack(D, world, rank, dp(Hmine), ip(parent), dp(e8), dp(mine)) == 0
    Hall = [torch.zeros(P, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(Hall, torch.from_numpy(mine))                    # msevb_comm_allgather_evb
    allp = np.ascontiguousarray(np.concatenate([t.numpy() for t in Hall]))
    H = np.zeros((D, D))
    esum = np.zeros(8)
    assert L.msevb_evb_payload_unpack(D, world, dp(allp), ip(parent), dp(H), dp(esum)) == 0
    assert np.array_equal(esum, np.arange(8) * sum(range(1, world + 1)))
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
