#!/usr/bin/env python3
"""One GPU, R virtual ranks (the Amdahl term of the strong-scaling layout without an 8-GPU node).

For R in 1, 2, 4, 8 and every rank r < R: a context of its own with `msevb_comm_set_virtual(r, R)` + `shard_principal` (what msevb_comm_init
switches on) runs the MS-EVB force call of BASELINE.json configs[2] (98 220 atoms, 8 diabats, 64^3) and then its share of the blend with a
fixed ground-state vector - everything a rank does between the two collectives of a step, nothing of the collectives themselves (the
all-gather carries 2S + 9 doubles, the all-reduce 24 B/atom = 2.4 MB).  Reported: the slowest rank's time over the one-rank time.

    python tools/virtual_rank_times.py [calls] [B|C] [R,R,...] [host|device] [option=value ...]      (C = configs[4]'s size: 985 527 atoms, 16 diabats, 128^3; default ranks 1,2,4,8)
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from reactive_pb_nn_md_amd import api, forcefield, synth, system  # noqa: E402


def gathered_payloads(spec, R):
    """the R payloads an all-gather would deliver, from one context that runs the multi-rank device path as rank 0 of 1 (tests/test_gpu_parity.py: _gathered_payloads)"""
    import ctypes as C
    h = api.MsEvbHip(spec)
    for kv in OPTS:
        k, v = kv.split("=")
        h.set_option(k, float(v))
    h.comm_set_virtual(0, 1)
    h.ms_evb_calculate_total_force_energy(want_force=False)
    st = h.evb_state()
    D = st["D"]
    P1 = h.evb_payload_size(D, 1)
    h.comm_set_virtual_payload(np.zeros(P1))
    h.ms_evb_calculate_total_force_energy(want_force=False)
    one = h.comm_get_payload(P1)
    Hp = np.zeros((D, D))
    parent = np.ascontiguousarray(st["coupling_matrix"] - 1, dtype=np.int32)
    parent[0] = -1
    dp, ip = (lambda a: a.ctypes.data_as(C.POINTER(C.c_double))), (lambda a: a.ctypes.data_as(C.POINTER(C.c_int)))
    h._chk(h.L.msevb_evb_payload_unpack(D, 1, dp(one), ip(parent), dp(Hp), None))
    pay = np.zeros((R, h.evb_payload_size(D, R)))
    for r in range(R):
        h._chk(h.L.msevb_evb_payload_pack(D, R, r, dp(Hp), ip(parent), None, dp(pay[r])))
    h.close()
    return pay


def rank_time(spec, r, R, eigvec, calls):
    if MODE == "device" and R > 1:      # the path of a rank with a communicator, minus the collectives: solve on the gathered entries on the device, blend pre-enqueued
        pay = gathered_payloads(spec, R)
        h = api.MsEvbHip(spec)
        for kv in OPTS:
            k, v = kv.split("=")
            h.set_option(k, float(v))
        h.set_option("shard_principal", 1)
        h.comm_set_virtual(r, R)
        h.comm_set_virtual_payload(pay)
        h.construct_verlet_list()
        for _ in range(5):
            h.ms_evb_calculate_total_force_energy(want_force=False)
        t0 = time.perf_counter()
        for _ in range(calls):
            h.ms_evb_calculate_total_force_energy(want_force=False)
        h.synchronize()
        dt = (time.perf_counter() - t0) / calls
        D = h.evb_state()["D"]
        h.close()
        return dt, D
    h = api.MsEvbHip(spec)
    for kv in OPTS:
        k, v = kv.split("=")
        h.set_option(k, float(v))
    if R > 1:
        h.set_option("shard_principal", 1)
        h.comm_set_virtual(r, R)
    h.construct_verlet_list()
    for _ in range(5):
        h.ms_evb_calculate_total_force_energy(want_force=False)
        if R > 1:
            h.evb_partial_blend(eigvec, download=False)
    t0 = time.perf_counter()
    for _ in range(calls):
        h.ms_evb_calculate_total_force_energy(want_force=False)
        if R > 1:
            h.evb_partial_blend(eigvec, download=False)
    dt = (time.perf_counter() - t0) / calls
    D = h.evb_state()["D"]
    h.close()
    return dt, D


_ARGV = sys.argv if __name__ == "__main__" else [sys.argv[0]]
MODE = "host" if "host" in _ARGV[1:] else "device"      # device (default): virtual ranks get the gathered payloads and run the device-side solve + pre-enqueued blend; host: round 4's path (the caller merges with a given eigenvector)
if __name__ == "__main__":
    sys.argv = [a for a in sys.argv if a not in ("host", "device")]
OPTS = [a for a in _ARGV[1:] if "=" in a]      # library options, name=value (e.g. step_graph=0)
if __name__ == "__main__":
    sys.argv = [a for a in sys.argv if "=" not in a]


def main():
    calls = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    cfg = sys.argv[2] if len(sys.argv) > 2 else "B"
    n_side, n_dia, grid, chain = (69, 16, 128, 4) if cfg == "C" else (32, 8, 64, 3)      # B = bench.py's default workload
    s = synth.make_system(n_side, n_diabat=n_dia, seed=0, max_chain=chain)
    p = forcefield.build_params(forcefield.ch3so3h_water_forcefield(), moltype_order=["h2o", "so3h"])
    spec = system.from_synth(s, p, rc=9.0, rv=10.5, alpha=0.3, pme_grid=grid, max_chain=chain)
    h = api.MsEvbHip(spec)
    h.ms_evb_calculate_total_force_energy(want_force=False)
    ev = h.evb_state()["eigvec"].copy()
    h.close()
    out = {}
    ranks = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 4, 8]
    if 1 not in ranks:      # the ratios are over ONE rank's time, whatever the list asks for (a list starting with 4 used to take rank 0 of R = 4 as the baseline)
        ranks = [1] + ranks
    ranks = [1] + [R for R in ranks if R != 1]
    t1 = None
    for R in ranks:
        ts = []
        for r in range(R):
            dt, D = rank_time(spec, r, R, ev, calls)
            ts.append(dt * 1e6)
        if R == 1:
            t1 = ts[0]
        out[R] = {"us_per_rank": [round(t, 1) for t in ts], "slowest_over_one_rank": round(max(ts) / t1, 3), "n_diabats": int(D)}
        print("R=%d  slowest %.1f us  (x%.3f of one rank)  per rank: %s" % (R, max(ts), max(ts) / t1, " ".join("%.0f" % t for t in ts)), flush=True)
    print(json.dumps({"workload": "%d atoms, %d diabats, %d^3, force call + own share of the blend, no collectives" % (spec.n_atom, n_dia, grid), "calls": calls, "ranks": out}))


if __name__ == "__main__":
    main()
