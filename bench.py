#!/usr/bin/env python3
"""bench.py - MD timesteps/s of the MS-EVB force/energy hot path on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): 32^3-site lattice water + CH3SO3H + planted
H-bond tree -> ~98k atoms, 8 diabats, 64^3 PME grid, r_c 9 / r_v 10.5 A, dt 0.5 fs, NVE.  A step = one
velocity-Verlet step of msevb_md_integrate_atomic with the EVB force call (principal diabat, 7 diabat deltas, Hamiltonian,
Hellmann-Feynman blend), positions/velocities/forces resident in HBM.  With N > 1 the diabats are sharded over the ranks
(and the principal diabat's pair loop / per-molecule terms by atom ranges); per step one small RCCL all-gather of Hamiltonian entries and one all-reduce of the partial
Hellmann-Feynman forces: total work is fixed -> "scaling": "strong".
Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from reactive_pb_nn_md_amd import api, forcefield, synth, system  # noqa: E402

FP64_VALU_PEAK_TFLOPS = 78.6      # MI355X fp64 vector: 256 CUs x 64 lanes x 2 x 2.4 GHz
HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s (6.3 TB/s achievable)


TRAFFIC_SOURCES = ("ctx.hpp", "devmath.hpp", "gather_dev.hpp", "molecule.hip", "molecule_dev.hpp", "neighbor.hip", "pair.hip", "pair_flat.hip", "pme.hip")      # same list in tools/make_traffic_json.py


def algorithmic_bytes(N, K, D, D_local, half_list_entries):
    """SURVEY.md 8d per-step algorithmic bytes of each phase (fp64 data, int32 indices)"""
    K3 = K ** 3
    nd = max(D_local - 1, 0)
    return {
        "real space (pair loop)": 4 * half_list_entries + 60 * N,
        "Q_grid (spread)": 32 * N + 8 * K3,
        "Fourier transforms + B*C*FQ": 64 * K3 + 40 * K3,
        "Q grid derivatives (gather)": 56 * N + 8 * K3,
        "intra-molecular (non-bonded + bonded)": 52 * N,
        "ms-evb reciprocal space": nd * ((8 + 64 + 40 + 8) * K3 + 56 * N),
        "ms-evb diabat real space": nd * (60 * N + 56 * N + 24 * N),
        "ms-evb Hellmann-Feynman blend": 24 * N * (2 * D - 1) + 24 * N,
    }


def moved_bytes(N, K, D, D_local, half_list_entries, linear=True):
    """the same accounting for what THIS build does per step when evb_recip_linear is on (default): no transform pair and no all-atom gather per
    non-principal diabat (their reciprocal energies / special-atom forces come from stencil sums over a few dozen atoms) but ONE extra
    solve of the blended charge grid; the diabats' real-space terms touch the candidates near the special atoms only (counted as SURVEY does: upper bound)"""
    ab = algorithmic_bytes(N, K, D, D_local, half_list_entries)
    if linear and D > 1:
        K3 = K ** 3
        ab["ms-evb reciprocal space"] = (64 + 40) * K3      # the blended grid's transform pair + convolution; its gather is the ONE gather counted above
    return sum(ab.values())


def _timeit(fn):
    t0 = time.time()
    fn()
    return time.time() - t0


def cpu_baseline(spec, steps_per_rebuild):
    """The CPU oracle ("port") timed on this host, on a bounded sample of the same workload:
      * thread sweep of the principal-diabat force call (calculate_total_force_energy restatement) - the oracle keeps the reference's
        OpenMP structure (per-thread force replicas summed serially, src/pair_int_real_space.f90:146-147,:319-327; serial spread), so it
        stops scaling long before the host runs out of cores; the best thread count is used for the rest;
      * value: TIMED velocity-Verlet steps of the reference's delta-update formulation (orc_evb_force_energy_delta: principal diabat in full, every further
        diabat as the change of its donor / acceptor molecules + one reciprocal-space pass of its own - src/ms_evb.f90:1421-1560, :1962-2090);
      * from_scratch_steps_per_s: a step of the oracle's other formulation (every diabat from scratch in its own topology - more work than
        the reference's delta update: a lower bound of CPU steps/s); principal_only_steps_per_s: an upper bound.  The Verlet build is amortised
        over the bench's own rebuild interval."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cpu = "?"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    sweep, best = [], None
    o = oracle_py.Oracle(spec, threads=min(cores, 32))
    t0 = time.time()
    o.verlet_build()
    t_list = time.time() - t0
    for th in sorted({t for t in (8, 16, 32, 64, 128, cores) if t <= cores}):
        o.L.orc_set_threads(o.h, th)
        o.total()                                   # first call at this thread count: scratch allocation, page faults
        t0 = time.time(); o.total(); dt = time.time() - t0
        sweep.append((th, dt))
        if best is None or dt < best[1]:
            best = (th, dt)
    th = best[0]
    o.L.orc_set_threads(o.h, th)
    # what the REFERENCE does per step (src/ms_evb.f90:375-470): the principal diabat's full force call, then per further diabat a DELTA update - the donor /
    # acceptor molecules against all atoms in both topologies, their stencils on a copy of the principal charge grid (modify_Q_grid), and one reciprocal-space pass of
    # its own (transform, convolution, back-transform, force update over all atoms: calculate_reciprocal_space_pme, :1962-2090).  The oracle has that formulation
    # (orc_evb_force_energy_delta, pinned against the real reference's Hamiltonians and force stores in tests/test_oracle_vs_reference.py) and it is TIMED here,
    # whole velocity-Verlet steps of it, not assembled from stages.
    t_tot = min(_timeit(o.total) for _k in range(3))
    vel = synth.maxwell_boltzmann(spec.mass, 300.0, 12345)
    F = o.evb(delta=True)["F"].copy()
    D = o.n_diabat()
    n_evb, t0 = 0, time.time()
    while n_evb < 3 or (time.time() - t0 < 12.0 and n_evb < 12):
        o.md_step(vel, F, spec.delta_t, 2)      # evb = 2: the delta-update formulation
        n_evb += 1
    t_evb = (time.time() - t0) / n_evb
    t_delta = t_evb + t_list / max(steps_per_rebuild, 1.0)
    t0 = time.time()
    o.md_step(vel, F, spec.delta_t, 1)      # one step of the from-scratch formulation (every diabat in its own topology with its own Verlet list), for the record
    t_scratch = time.time() - t0 + t_list / max(steps_per_rebuild, 1.0)
    return {"value": 1.0 / t_delta, "unit": "steps/s", "cores": th, "kind": "port",
            "sample": "CPU oracle on the same %d-atom box, %d diabats, %d OpenMP threads (%s, %d hardware threads visible, no pinning; fastest of a sweep %s s at %s threads for the "
                      "principal force call): %d TIMED velocity-Verlet steps of the reference's DELTA-UPDATE formulation (principal-diabat force call %.3f s of it; per further diabat the "
                      "donor / acceptor real-space and intra-molecular deltas by all-atom scans, modify_Q_grid, one transform pair + convolution and the all-atom reciprocal force "
                      "update; couplings, Jacobi, Hellmann-Feynman blend, integrator) = %.3f s per step, + Verlet build %.2f s amortised over %.0f steps.  "
                      "from_scratch_steps_per_s: one step of the oracle's other formulation (every diabat from scratch, %.2f s) - more work than the reference does.  The oracle keeps "
                      "the reference's per-thread force replicas with a serial sum and its serial charge spreading, so it stops scaling long before the host runs out of cores"
                      % (spec.n_atom, D, th, cpu, cores, "/".join("%.2f" % d for _t, d in sweep), "/".join(str(t) for t, _d in sweep), n_evb, t_tot, t_evb, t_list,
                         steps_per_rebuild, t_scratch),
            "from_scratch_steps_per_s": 1.0 / t_scratch,
            "principal_only_steps_per_s": 1.0 / best[1]}


def thermalised_line(args, device):
    """Second measurement on a thermalised box (rank 0, N = 1 only): the same composition built with liquid_like=True (rigid molecules at the
    force field's equilibrium geometry, random orientations) and relaxed on the GPU, outside the timed region, by 8 blocks of 20 NVE steps
    with a velocity rescale to 300 K in between (tools/make_thermal_start.py measured this protocol: T 315-360 K afterwards, the planted
    8-diabat tree lives ~230 more steps, Verlet rebuilds at a liquid's pace); then the warm-up and min(steps, 200) timed steps."""
    KB = 0.0083144626
    s = synth.make_system(args.n_side, n_diabat=args.diabats, seed=0, max_chain=3 if args.diabats <= 8 else 4, liquid_like=True)
    p = forcefield.build_params(forcefield.ch3so3h_water_forcefield(), moltype_order=["h2o", "so3h"])
    spec = system.from_synth(s, p, rc=9.0, rv=10.5, alpha=0.3, pme_grid=args.pme_grid, max_chain=3 if args.diabats <= 8 else 4)
    h = api.MsEvbHip(spec, device=device)
    for kv in args.opt:
        k, v = kv.split("=")
        h.set_option(k, float(v))
    h.set_velocities(synth.maxwell_boltzmann(spec.mass, 300.0, 12345))
    h.ms_evb_calculate_total_force_energy(want_force=False)
    ndf = 3 * spec.n_atom - 3
    for _b in range(8):
        en = h.md_integrate_atomic(20, spec.delta_t, use_evb=True)
        T = 2 * en[-1, 1] / (ndf * KB)
        h.set_velocities(h.get_velocities() * np.sqrt(300.0 / max(T, 1.0)))
    h.ms_evb_calculate_total_force_energy(want_force=False)
    if args.warmup > 0:
        h.md_integrate_atomic(args.warmup, spec.delta_t, use_evb=True)
    h.set_option("phase_timing", 1.0)      # counts the list rebuilds
    h.L.msevb_reset_phase_times(h.ctx)
    D0 = int(h.evb_state()["D"])
    steps = min(args.steps, 200)
    h.synchronize()
    t0 = time.perf_counter()
    en = h.md_integrate_atomic(steps, spec.delta_t, use_evb=True)
    h.synchronize()
    dt = time.perf_counter() - t0
    reb = int(h.phase_times().get("verlet list construction", (0, 0))[1])
    out = {"value": round(steps / dt, 3), "unit": "steps/s", "ms_per_step": round(dt / steps * 1e3, 4), "steps": steps,
           "workload": "same composition (%d atoms, %d^3 grid), liquid_like start relaxed by 8 x 20 steps with velocity rescaling to 300 K before the warm-up" % (spec.n_atom, spec.pme_grid),
           "n_diabats_start_end": [D0, int(h.evb_state()["D"])], "verlet_rebuilds": reb,
           "temperature_first_last_K": [round(2 * en[0, 1] / (ndf * KB), 1), round(2 * en[-1, 1] / (ndf * KB), 1)],
           "energy_first_last": [float(en[0].sum()), float(en[-1].sum())]}
    h.close()
    return out


def long_run_line(args, device, n_steps=2000, block=500):
    """Third measurement (rank 0, N = 1 only): SUSTAINED throughput.  The hot lattice of `value` lives ~300 steps and the thermalised box ~280 (the acid
    proton, which has no repulsive wall outside the EVB terms, falls onto a 1-4 sulfonate oxygen of its own molecule once the diabat tree has dissolved - the reference walks
    the same surface, tests/golden/long_evb_n12_liquid_s40.npz).  With the 1-4 pairs EXCLUDED (n_exclusions = 3, the reference's [ exclusions ] generation,
    src/intra_bonded_interactions.f90; fixture std_n6_ex3_s12) the same 98 220-atom composition survives: thermalised as `thermalised_box`, then n_steps NVE steps in blocks,
    timed as a whole - list rebuilds, diabat-count changes, tree changes (device-side re-planning) and the host's per-block bookkeeping included."""
    KB = 0.0083144626
    s = synth.make_system(args.n_side, n_diabat=args.diabats, seed=0, max_chain=3 if args.diabats <= 8 else 4, liquid_like=True)
    p = forcefield.build_params(forcefield.ch3so3h_water_forcefield(), moltype_order=["h2o", "so3h"], n_exclusions=3)
    spec = system.from_synth(s, p, rc=9.0, rv=10.5, alpha=0.3, pme_grid=args.pme_grid, max_chain=3 if args.diabats <= 8 else 4)
    h = api.MsEvbHip(spec, device=device)
    for kv in args.opt:
        k, v = kv.split("=")
        h.set_option(k, float(v))
    h.set_velocities(synth.maxwell_boltzmann(spec.mass, 300.0, 12345))
    h.ms_evb_calculate_total_force_energy(want_force=False)
    ndf = 3 * spec.n_atom - 3
    for _b in range(8):
        en = h.md_integrate_atomic(20, spec.delta_t, use_evb=True)
        T = 2 * en[-1, 1] / (ndf * KB)
        h.set_velocities(h.get_velocities() * np.sqrt(300.0 / max(T, 1.0)))
    h.ms_evb_calculate_total_force_energy(want_force=False)
    h.md_integrate_atomic(20, spec.delta_t, use_evb=True)
    h.set_option("phase_timing", 1.0)      # counts the list rebuilds
    h.L.msevb_reset_phase_times(h.ctx)
    info0 = {k: int(h.info(k)) for k in ("evb_calls", "spec_misses", "device_plans", "host_dfs_calls")}
    dcounts, e_first, e_last, T_last, done, err = set(), None, None, None, 0, None
    h.synchronize()
    t0 = time.perf_counter()
    while done < n_steps:
        try:
            tr = h.md_integrate_atomic(min(block, n_steps - done), spec.delta_t, use_evb=True)
        except api.MsevbError as e:
            err = str(e)[:160]
            break
        done += len(tr)
        if e_first is None:
            e_first = float(tr[0].sum())
        e_last, T_last = float(tr[-1].sum()), 2 * tr[-1, 1] / (ndf * KB)
        dcounts.add(int(h.evb_state()["D"]))
    h.synchronize()
    dt = time.perf_counter() - t0
    reb = int(h.phase_times().get("verlet list construction", (0, 0))[1])
    info1 = {k: int(h.info(k)) for k in info0}
    nh = 0
    try:
        nh = int(h.hop_log()[0])
    except Exception:
        pass
    out = {"value": round(done / dt, 3) if done else 0.0, "unit": "steps/s", "ms_per_step": round(dt / max(done, 1) * 1e3, 4), "steps": done, "steps_asked": n_steps,
           "workload": "same composition (%d atoms, %d^3 grid), 1-4 pairs excluded (n_exclusions = 3), liquid_like start relaxed as thermalised_box; %d NVE steps in blocks of %d timed as a whole" % (spec.n_atom, spec.pme_grid, n_steps, block),
           "verlet_rebuilds": reb, "principal_diabat_switches": nh, "diabat_counts_seen": sorted(dcounts),
           "tree_changes_replanned_on_the_device": info1["device_plans"] - info0["device_plans"], "host_dfs_calls": info1["host_dfs_calls"] - info0["host_dfs_calls"],
           "speculation_misses": info1["spec_misses"] - info0["spec_misses"], "evb_force_calls": info1["evb_calls"] - info0["evb_calls"],
           "relative_energy_change": None if e_first is None else (e_last - e_first) / abs(e_first), "temperature_last_K": None if T_last is None else round(T_last, 1), "ended_early": err}
    h.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--n-side", type=int, default=32)
    ap.add_argument("--diabats", type=int, default=8)
    ap.add_argument("--pme-grid", type=int, default=64)
    ap.add_argument("--repeats", type=int, default=5, help="timed blocks of --steps steps (each from the initial state + warm-up); the MEDIAN block is reported")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-evb", action="store_true")
    ap.add_argument("--no-thermal", action="store_true", help="skip the second (thermalised box) measurement")
    ap.add_argument("--no-long-run", action="store_true", help="skip the third measurement (2 000 steps of a box that survives: sustained throughput)")
    ap.add_argument("--no-phase-timing", action="store_true", help="skip the per-phase HIP events (roofline block becomes null)")
    ap.add_argument("--rccl-single", action="store_true", help="diagnostic: N=1 with a one-rank RCCL communicator (cost of the two collectives per step)")
    ap.add_argument("--opt", action="append", default=[], help="library option name=value (msevb_set_option), e.g. pair_tiles=0")
    args = ap.parse_args()
    # stdout carries exactly ONE line, the JSON record: libraries that write to file descriptor 1 on their own (RCCL prints a five-line
    # version banner there when a communicator is created) go to stderr for the whole run
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo")      # rendezvous / barriers / timing reduction only; data path is RCCL inside the library
        ndev = torch.cuda.device_count()      # does not initialise the GPU
        if ndev > 0:
            local_rank %= ndev                # a launcher that shows each rank one device only

    s = synth.make_system(args.n_side, n_diabat=args.diabats, seed=0, max_chain=3 if args.diabats <= 8 else 4)
    p = forcefield.build_params(forcefield.ch3so3h_water_forcefield(), moltype_order=["h2o", "so3h"])
    spec = system.from_synth(s, p, rc=9.0, rv=10.5, alpha=0.3, pme_grid=args.pme_grid, max_chain=3 if args.diabats <= 8 else 4)
    h = api.MsEvbHip(spec, device=local_rank)
    if world > 1:
        import torch
        idbuf = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            raw = (C.c_ubyte * 128)()
            h._chk(h.L.msevb_comm_unique_id(raw))
            idbuf = torch.tensor(list(raw), dtype=torch.uint8)
        dist.broadcast(idbuf, 0)
        raw = (C.c_ubyte * 128)(*idbuf.tolist())
        h._chk(h.L.msevb_comm_init(h.ctx, raw, rank, world))
    elif args.rccl_single:
        h.comm_init(api.MsEvbHip.comm_unique_id(h.L), 0, 1)
    vel = synth.maxwell_boltzmann(spec.mass, 300.0, 12345)
    h.set_velocities(vel)
    use_evb = not args.no_evb

    def barrier():
        h.synchronize()
        if dist is not None:
            dist.barrier()
        h.synchronize()
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.set_device(local_rank)      # never torch's default device 0 from every rank
                torch.cuda.synchronize()
        except Exception:
            pass

    # The synthetic box is an unrelaxed lattice: it heats up, its planted H-bond tree dissolves after ~300 steps and around step 450 the
    # example force field's acid proton (no repulsion outside the EVB terms) collapses onto a water oxygen - the library then stops with
    # "unstable trajectory", as the reference does.  Long runs therefore go in segments of at most SEG steps, each from the initial state;
    # the reset (two 2.4 MB uploads + one force call) happens inside the timed region and is counted.
    SEG = 300
    state = {"since": 0, "resets": 0}

    def reset():
        h.set_positions(spec.xyz)
        h.set_velocities(vel)
        h.construct_verlet_list()
        if use_evb:
            h.ms_evb_calculate_total_force_energy(want_force=False)
        else:
            h.calculate_total_force_energy(want_force=False)
        state["since"] = 0

    def run_steps(n):
        out = []
        while n > 0:
            if state["since"] >= SEG:
                reset()
                state["resets"] += 1
            m = min(n, SEG - state["since"])
            out.append(h.md_integrate_atomic(m, spec.delta_t, use_evb=use_evb))
            state["since"] += m
            n -= m
        return np.concatenate(out) if out else np.zeros((0, 2))

    # library options of this run (diagnostics, A/B runs) BEFORE the first force call: the warm-up runs the schedule that is timed (and a profiler's counter passes,
    # which serialise kernels, never see the default schedule's device-side waits)
    for kv in args.opt:
        k, v = kv.split("=")
        h.set_option(k, float(v))
    # initial forces + warm-up
    if use_evb:
        h.ms_evb_calculate_total_force_energy(want_force=False)
    else:
        h.calculate_total_force_energy(want_force=False)
    if args.warmup > 0:
        run_steps(args.warmup)
    h.set_option("phase_timing", 0.0 if args.no_phase_timing else 1.0)      # HIP events on the library's stream around every phase (live roofline input)
    h.L.msevb_reset_phase_times(h.ctx)
    D_start = int(h.evb_state()["D"]) if use_evb else 1      # diabats when the timed region starts (the planted tree of the hot lattice box dissolves over time)
    # `repeats` timed blocks of EXACTLY --steps steps, each bracketed by barrier + synchronize on both sides and each started from the same state
    # (initial coordinates and velocities, list build, force call, warm-up - all outside the timed region); the MEDIAN block is the reported one
    # and all of them are listed: the driver's --steps 20 is a 9 ms region with one list rebuild in it, one block of that decides nothing
    blocks = []
    for rep_i in range(max(args.repeats, 1)):
        if rep_i > 0:
            reset()
            if args.warmup > 0:
                run_steps(args.warmup)
        barrier()
        reb0 = int(h.phase_times().get("verlet list construction", (0, 0))[1])
        t0 = time.perf_counter()
        state["resets"] = 0
        en_b = run_steps(args.steps)
        barrier()
        dt_b = time.perf_counter() - t0
        reb_b = int(h.phase_times().get("verlet list construction", (0, 0))[1]) - reb0
        if dist is not None:
            import torch
            t = torch.tensor([dt_b], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_b = float(t[0])
        blocks.append((dt_b, en_b, state["resets"], reb_b))
    order = sorted(range(len(blocks)), key=lambda k: blocks[k][0])
    dt, en, state["resets"], rebuilds = blocks[order[len(order) // 2]]
    phases = h.phase_times()
    st = h.evb_state() if use_evb else {"D": 1}
    D_end = int(st["D"])
    D = D_start
    vp_n = C.c_long()
    h._chk(h.L.msevb_get_verlet_list(h.ctx, None, None, C.byref(vp_n)))
    D_local = 1 + sum(1 for d in range(1, D) if (d - 1) % world == rank)
    ab = algorithmic_bytes(spec.n_atom, spec.pme_grid, D, D_local, vp_n.value)
    kern = {k: v for k, v in phases.items() if v[1] > 0 and k in ab}
    # every phase runs once per step; the library records its HIP events on one force call in eight (sampling keeps the ~25 event
    # records per step off the host's critical path), so the averages are over the timed region, the counts are not per-step counts
    per_step = {k: v[0] for k, v in kern.items()}
    # the roofline block is about the dominant KERNEL: phases that are one kernel launch (the EVB phases are chains of kernels on side
    # streams, their event brackets include waits for other chains)
    single = ("real space (pair loop)", "Q_grid (spread)", "Q grid derivatives (gather)", "intra-molecular (non-bonded + bonded)")
    cand = {k: v for k, v in per_step.items() if k in single}
    dom = max(cand, key=cand.get) if cand else None
    roofline = None
    if dom:
        launches_per_step = 1.0
        avg_ms = kern[dom][0]
        achieved = ab[dom] / launches_per_step / (avg_ms * 1e-3) / 1e9
        traffic = None        # HBM bytes per launch from the committed PMC passes (profiles/traffic.json, tools/make_traffic_json.py)
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            wl = tj.get("_workload", {})
            hh = hashlib.sha256()      # the sources of the kernels traffic.json covers (pair loop, PME spread / gather, per-molecule terms) and of the list they read
            for name in TRAFFIC_SOURCES:
                hh.update(open(os.path.join(ROOT, "reactive_pb_nn_md_amd", "csrc", name), "rb").read())
            if dom in tj and world == 1 and wl.get("n_atoms") == spec.n_atom and wl.get("pme_grid") == spec.pme_grid and use_evb and not args.opt and tj.get("_build") == hh.hexdigest():
                traffic = int(tj[dom]["hbm_bytes_per_launch"])
        except (OSError, ValueError, KeyError):
            pass
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "avg_launch_ms": round(avg_ms, 5), "algorithmic_bytes_per_launch": int(ab[dom] / launches_per_step)}
        if dom == "real space (pair loop)":
            # the pair arithmetic is fp64 VALU, report that fraction beside the HBM one.  Algorithmic flops =
            # 50 per pair inside r_c, pairs counted once (P_rc = N * 4/3 pi r_c^3 rho / 2); peak = 256 CUs x 64 lanes x 2 (FMA) x 2.4 GHz.
            p_rc = spec.n_atom * (4.0 / 3.0 * np.pi * spec.rc ** 3 * spec.n_atom / float(np.prod(spec.box))) / 2.0
            tf = 50.0 * p_rc / (avg_ms * 1e-3) / 1e12
            roofline["fp64_valu"] = {"algorithmic_flops_per_launch": int(50.0 * p_rc), "achieved": round(tf, 3), "peak": FP64_VALU_PEAK_TFLOPS,
                                     "unit": "TFLOP/s", "frac": round(tf / FP64_VALU_PEAK_TFLOPS, 5)}
    out = None
    if rank == 0:
        b_step = sum(ab.values())
        out = {
            "metric": "MD timesteps/sec", "value": round(args.steps / dt, 3), "unit": "steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[2] as SURVEY 8d prescribes it (jittered lattice, hot): CH3SO3H/H2O box %d atoms, %d diabats, %d^3 PME grid, r_c 9.0 / r_v 10.5 A, dt 0.5 fs, NVE, %s force call"
                                   % (spec.n_atom, D, spec.pme_grid, "MS-EVB" if use_evb else "non-EVB"),
                       "n_atoms": spec.n_atom, "n_diabats": D, "n_diabats_at_end": D_end, "restarts_from_initial_state_in_timed_region": state["resets"], "pme_grid": spec.pme_grid, "parallelism": "diabats sharded over %d GPU(s)" % world},
            "roofline": roofline,
            # twice: SURVEY 8d's yardstick (it credits a transform pair + gather per non-principal diabat, which this build replaces by stencil sums and ONE
            # blended solve) and the bytes this build's kernels are there to move - the second one is the achieved-bandwidth statement
            "step_roofline": {"algorithmic_bytes_per_step": int(b_step), "achieved_GBs": round(b_step / (dt / args.steps) / 1e9, 2),
                              "frac_of_hbm_peak": round(b_step / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 5),
                              "note": "SURVEY.md 8d formula, including (D-1) per-diabat transforms this build does not perform"},
            "step_roofline_moved": (lambda b: {"algorithmic_bytes_per_step": int(b), "achieved_GBs": round(b / (dt / args.steps) / 1e9, 2),
                                               "frac_of_hbm_peak": round(b / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 5),
                                               "note": "bytes of the kernels this build runs: SURVEY's terms with the per-diabat reciprocal term replaced by one more solve (the blended grid) and without the per-diabat all-atom gathers"})(
                                       moved_bytes(spec.n_atom, spec.pme_grid, D, D_local, vp_n.value, linear=use_evb)),
            "phases_ms_per_step": {k: round(v, 5) for k, v in sorted(per_step.items(), key=lambda kv: -kv[1])},
            "blocks_ms_per_step": [round(b[0] / args.steps * 1e3, 4) for b in blocks],      # every timed block; `value` is the median one
            "verlet_rebuilds": rebuilds,      # inside the reported (median) block
            "energy_first_last": [float(en[0, 0] + en[0, 1]), float(en[-1, 0] + en[-1, 1])],
            "library_sha256": hashlib.sha256(open(os.path.join(ROOT, "reactive_pb_nn_md_amd", "libmsevb_hip.so"), "rb").read()).hexdigest()[:16],      # which build this line is about
            "options": list(args.opt),
            "step_graph": {k: int(h.info(k)) for k in ("evb_calls", "spec_steps", "spec_misses", "graph_launches", "graph_captures", "graph_mismatches", "device_plans")},      # MS-EVB calls of the whole run (warm-ups and all blocks): how many went out as four linear graph launches
        }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(spec, args.steps / max(out["verlet_rebuilds"], 1))
    elif rank == 0:
        out["cpu_baseline"] = None
    h.close()
    if rank == 0 and world == 1 and use_evb and not args.no_thermal and args.n_side >= 16:
        out["thermalised_box"] = thermalised_line(args, local_rank)
    if rank == 0 and world == 1 and use_evb and not args.no_long_run and not args.no_thermal and args.n_side >= 16:
        out["long_run"] = long_run_line(args, local_rank)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        real_stdout.write(json.dumps(out) + "\n")
        real_stdout.flush()


if __name__ == "__main__":
    main()
