"""GPU parity tests (run on the MI355X box: pytest -m gpu).  Everything goes through the C-ABI
(reactive_pb_nn_md_amd.api -> libmsevb_hip.so).  Checkers: golden vectors of the real reference, and the CPU oracle.

Tolerances (fp64 path; the GPU sums in a different order and uses reciprocal multiplies in the pair kernel):
  forces   abs 1e-9 kJ/mol/A  (measured 1e-12)     energies rel 1e-11 (measured 1e-15)
  integer work (Verlet pair set, PME grid / table indices) bit-exact.
"""
import hashlib
import os

import numpy as np
import pytest

from conftest import all_golden, golden, grid_diff, pre_switch_spec, reference_atom_order
from reactive_pb_nn_md_amd import api, forcefield, synth, system

pytestmark = pytest.mark.gpu

F_TOL = 1e-9
E_RTOL = 1e-11


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def rel(a, b):
    return abs(a - b) / max(abs(b), 1e-300)


def production_kernels(h, name):
    """the n12 fixtures (>= 4 neighbour cells per axis) must be served by the kernels bench.py runs: tiled list build, k_pair_flat on the
    half list, tiled weight-free spread / gather - asserted, so that a change of the dispatch rules cannot silently take them off the fixtures"""
    if "_n12_" not in name:
        return
    assert h.info("cells_per_axis_min") >= 4 and h.info("list_build_tiles") == 1 and h.info("pair_kernel") == 2
    # the PME tile of a cell (K / cells + 8 points per axis) must fit 64 KB of LDS: 32^3 on 4 cells per axis does (16^3), 64^3 does not (24^3; the
    # benchmark's 64^3 on 9 cells per axis does: 16^3) - those fixtures take the plain spread / gather
    assert h.info("pme_tiles") == (1 if h.spec.pme_grid <= 32 else 0)


@pytest.mark.parametrize("name", ["std_n8_s0", "std_n6_s3"])
def test_tables_bit_exact(name):
    g = golden(name)
    h = api.MsEvbHip(system.from_golden(g))
    for key in ("B6_spline", "B5_spline", "erfc_table", "ewaldscale_table", "Tang_Toennies_table", "dTang_Toennies_table", "CB"):
        assert sha(h.table(key)) == str(g[key + "_sha256"]), key
    assert h.ewald_self() == g["Ewald_self"][0]
    h.close()


@pytest.mark.parametrize("name", ["std_n8_s0", "std_n6_s3", "std_n12_k32_s26"])
def test_verlet_pair_set_identical(name):
    g = golden(name)
    spec = system.from_golden(g)
    h = api.MsEvbHip(spec)
    h.construct_verlet_list()
    vp, nl = h.get_verlet_list()
    assert len(nl) == int(g["neighbor_count"]) and np.array_equal(vp, g["verlet_point"])
    hh = hashlib.sha256()
    for i in range(spec.n_atom):
        hh.update(np.int32(i + 1).tobytes())
        hh.update(np.sort(nl[vp[i] - 1:vp[i + 1] - 1]).astype("<i4").tobytes())
    assert hh.hexdigest() == str(g["neighbor_pairs_sha256"])
    h.close()


@pytest.mark.parametrize("name", ["std_n8_s0", "std_n6_s3", "std_n6_sapt_s6", "std_n6_lb_s9", "std_n6_rb_s10", "std_n6_ex3_s12", "std_n6_k20_s13", "std_n6_k64_s17", "std_n6_ortho_s19", "std_n12_k32_s26", "std_n12_sapt_s32"])
def test_stages_against_reference(name):
    g = golden(name)
    h = api.MsEvbHip(system.from_golden(g))
    h.construct_verlet_list()
    production_kernels(h, name)
    h.zero_forces()
    e = h.real_space_energy_force()
    assert np.abs(h.get_forces() - (g["F_pair"] + g["F_intra_nb"])).max() < F_TOL
    assert rel(e[0], g["E_elec_pair"][0]) < E_RTOL and rel(e[1], g["E_vdw_pair"][0]) < E_RTOL and rel(e[2], g["E_elec_intra_nb"][0]) < E_RTOL
    h.zero_forces()
    er = h.pme_reciprocal_space_energy_force()
    assert np.abs(h.get_forces() - g["F_recip"]).max() < F_TOL and rel(er, g["E_recip"][0]) < E_RTOL
    Q, th, u, fr = h.pme_state()
    assert np.array_equal(u, g["xyz_scale"])                       # scaled coordinates bit-exact
    assert grid_diff(g, "Q_grid", Q) < 1e-14 and grid_diff(g, "theta_conv_Q", th) < 1e-13      # (64^3: the library's own LDS transform, fft_lds.hip, against MKL's)
    near, spl = h.pme_indices()
    assert np.array_equal(near, np.floor(g["xyz_scale"]).astype(np.int32))          # PME grid indices bit-exact
    arg = g["xyz_scale"][:, :, None] - (np.floor(g["xyz_scale"])[:, :, None] - np.arange(6)[None, None, :])
    assert np.array_equal(spl, np.ceil(arg / 6.0 * 100000.0).astype(np.int32))      # B-spline table rows bit-exact
    h.zero_forces()
    eb = h.intra_molecular_energy_force()
    assert np.abs(h.get_forces() - g["F_bonded"]).max() < F_TOL
    assert rel(eb[0], g["E_bond"][0]) < E_RTOL and rel(eb[1], g["E_angle"][0]) < E_RTOL and rel(eb[2], g["E_dihedral"][0]) < E_RTOL
    en, F = h.calculate_total_force_energy()
    assert np.abs(F - g["F_total"]).max() < F_TOL and rel(en["potential"], g["E_total"][0]) < E_RTOL
    assert rel(en["elec"], g["E_elec_total"][0]) < E_RTOL and rel(en["vdw"], g["E_vdw_total"][0]) < E_RTOL
    h.close()


@pytest.mark.parametrize("name", ["std_n6_s3", "std_n6_sapt_s6", "evb_n6_d4_s2"])
def test_mfma_charge_spreading_against_reference(name):
    """option spread_mfma: grid_Q (src/pme.f90:184-262) as a per-cell contraction on the matrix cores (v_mfma_f64_16x16x4_f64, 16^3 tile per
    cell) - the real reference's Q_grid, theta and reciprocal forces; then the EVB force call on top of it"""
    g = golden(name)
    for bricks in (1, 0):      # 1: owner-computes MFMA bricks (a cell's workgroup writes its <= 8^3 brick of the grid, no atomics); 0: the 16^3-tile MFMA kernel
        h = api.MsEvbHip(system.from_golden(g))
        h.set_option("spread_bricks", bricks)
        h.set_option("spread_mfma", 1)
        h.construct_verlet_list()
        h.zero_forces()
        er = h.pme_reciprocal_space_energy_force()
        Q, th, u, fr = h.pme_state()
        assert np.abs(Q - g["Q_grid"]).max() < 1e-14 and np.abs(th - g["theta_conv_Q"]).max() < 1e-13
        assert np.abs(h.get_forces() - g["F_recip"]).max() < F_TOL and rel(er, g["E_recip"][0]) < E_RTOL
        if name.startswith("evb"):
            e, F = h.ms_evb_calculate_total_force_energy()
            assert rel(e, g["E_adiabatic"][0]) < E_RTOL and np.abs(F - g["F_adiabatic"]).max() < F_TOL
        h.close()


def test_mfma_charge_spreading_config_B():
    """... and at BASELINE.json configs[2] (98 220 atoms, 64^3: 729 cells, 16^3 tiles) against the LDS-atomic kernel: same grid, same forces,
    also along a trajectory on which atoms drift out of their cell's tile (those are spread point by point)"""
    spec = _synthetic(32, 8, 11, 64)
    out = []
    for bricks, mfma in ((0, 0), (0, 1), (1, 0)):      # LDS-atomic tiles (the default), MFMA tiles, owner-computes MFMA bricks
        h = api.MsEvbHip(spec)
        h.set_option("spread_bricks", bricks)
        h.set_option("spread_mfma", mfma)
        h.construct_verlet_list()
        h.zero_forces()
        er = h.pme_reciprocal_space_energy_force()
        Q = h.pme_state()[0]
        F = h.get_forces()
        h.set_velocities(synth.maxwell_boltzmann(spec.mass, 300.0, 99))
        h.ms_evb_calculate_total_force_energy(want_force=False)
        tr = h.md_integrate_atomic(25, spec.delta_t, use_evb=True)
        out.append((er, Q, F, tr, h.get_positions()))
        h.close()
    for k in (1, 2):
        assert rel(out[k][0], out[0][0]) < 1e-13 and np.abs(out[k][1] - out[0][1]).max() < 1e-13 and np.abs(out[k][2] - out[0][2]).max() < F_TOL, k
        assert np.abs(out[k][3] - out[0][3]).max() < 1e-8 * np.abs(out[0][3]).max() and np.abs(out[k][4] - out[0][4]).max() < 1e-8, k


def test_mfma_force_interpolation_config_B():
    """option gather_mfma: derivative_grid_Q (src/pme.f90:346-493) with the x contraction of every 16-atom block on the matrix cores
    (k_pme_gather_mfma, the transpose of the MFMA spread) at BASELINE.json configs[2] - the reciprocal forces of the default gather, also
    along a trajectory on which stencils leave their cell's tile (those atoms are gathered from global memory)"""
    spec = _synthetic(32, 8, 11, 64)
    out = []
    for mfma in (0, 1):
        h = api.MsEvbHip(spec)
        h.set_option("gather_mfma", mfma)
        h.construct_verlet_list()
        h.zero_forces()
        er = h.pme_reciprocal_space_energy_force()
        F = h.get_forces()
        h.set_velocities(synth.maxwell_boltzmann(spec.mass, 300.0, 99))
        e, Fe = h.ms_evb_calculate_total_force_energy()
        tr = h.md_integrate_atomic(25, spec.delta_t, use_evb=True)
        out.append((er, F, e, Fe, tr, h.get_positions()))
        h.close()
    a, b = out
    assert rel(a[0], b[0]) < 1e-13 and np.abs(a[1] - b[1]).max() < F_TOL and np.abs(a[1]).max() > 1.0
    assert rel(a[2], b[2]) < 1e-13 and np.abs(a[3] - b[3]).max() < 1e-10
    assert np.abs(a[4] - b[4]).max() < 1e-8 * np.abs(a[4]).max() and np.abs(a[5] - b[5]).max() < 1e-8


def test_own_cell_sort_against_radix_sort():
    """the list rebuild's cell sort (construct_verlet_list_grid's cell assignment, src/general_routines.f90:1408-1470): the library's stable
    counting sort (three kernels) against hipcub's radix sort (option cell_sort_own=0) at BASELINE.json configs[2] size and on a small
    fixture - the same half list, the same energies, the same 20-step trajectory (which rebuilds the list on the way)"""
    for spec in (_synthetic(32, 8, 11, 64), system.from_golden(golden("evb_n6_d4_s2"))):
        out = []
        for own in (1, 0):
            h = api.MsEvbHip(spec)
            h.set_option("cell_sort_own", own)
            h.construct_verlet_list()
            vp, nl = h.get_verlet_list()
            en = h.calculate_total_force_energy(want_force=False)[0]
            h.set_velocities(synth.maxwell_boltzmann(spec.mass, 600.0, 5))
            tr = h.md_integrate_atomic(20, spec.delta_t, use_evb=False)
            out.append((vp, nl, en["potential"], tr))
            h.close()
        a, b = out
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and rel(a[2], b[2]) < 1e-13
        assert np.abs(a[3] - b[3]).max() < 1e-9 * np.abs(b[3]).max()


def test_pme_without_the_weight_store_changes_nothing():
    """option pme_weightless (default on): k_pme_prep_light keeps (u, nearpt) only and the tiled spread / gather take the B-spline weights from the
    tables themselves (grid_Q / derivative_grid_Q, src/pme.f90:184-262, :346-493, table rows :220-247, :412-414) - the SAME bits as with the
    324 B/atom store of k_pme_prep, at BASELINE.json configs[2] size; the MS-EVB call with per-diabat transforms still gets its stored weights"""
    spec = _synthetic(32, 8, 11, 64)
    out = []
    for wl in (0, 1):
        h = api.MsEvbHip(spec)
        h.set_option("pme_weightless", wl)
        h.construct_verlet_list()
        h.zero_forces()
        er = h.pme_reciprocal_space_energy_force()
        Q, th, u, fr = h.pme_state()
        F = h.get_forces()
        e, Fe = h.ms_evb_calculate_total_force_energy()
        h.set_option("evb_recip_linear", 0)
        e2, Fe2 = h.ms_evb_calculate_total_force_energy()
        out.append((er, Q, th, u, F, e, Fe, e2, Fe2))
        h.close()
    a, b = out
    # (the weights are the same bits; the LDS atomics of the spread add them in order of arrival, so the grids agree to rounding)
    assert rel(a[0], b[0]) < 1e-14 and np.abs(a[1] - b[1]).max() < 1e-14 and np.abs(a[2] - b[2]).max() < 1e-13 and np.array_equal(a[3], b[3]) and np.abs(a[4] - b[4]).max() < 1e-12
    assert rel(a[5], b[5]) < 1e-14 and np.abs(a[6] - b[6]).max() < 1e-11      # (the EVB call sums with atomics: order of arrival, not bits)
    assert rel(a[7], b[7]) < 1e-14 and np.abs(a[8] - b[8]).max() < 1e-11


_SCHED_DEFAULT = {}


def _schedule_run(spec, opts):
    h = api.MsEvbHip(spec)
    for k, v in opts.items():
        h.set_option(k, v)
    e, F = h.ms_evb_calculate_total_force_energy()
    H = h.evb_state()["H"].copy()
    h.set_velocities(synth.maxwell_boltzmann(spec.mass, 300.0, 7))
    tr = h.md_integrate_atomic(12, spec.delta_t, use_evb=True)
    out = (e, F, H, tr, h.get_positions())
    h.close()
    return out


@pytest.mark.parametrize("opts", [{"evb_defer_fold": 0}, {"evb_defer_molecules": 1}, {"evb_defer_molecules": 2}, {"evb_region_main": 1},
                                  {"evb_coupling_blocks": 96}, {"evb_recip_linear": 0}, {"fft_own": 0, "evb_defer_fold": 0, "evb_device_solve": 0},
                                  {"overlap": 0, "graphs": 0}, {"evb_device_solve": 0}, {"evb_device_solve": 2}, {"evb_blend_aside": 0}, {"pair_pipe": 0}, {"pair_self_mode": 0},
                                  {"evb_device_solve": 0, "evb_blend_aside": 0}, {"step_fuse_refresh": 0}, {"evb_fft_patch": 0}, {"evb_fft_patch": 0, "evb_device_solve": 0},
                                  {"step_graph": 1}, {"evb_publish_spin": 0}, {"evb_device_solve": 0, "step_graph": 1}, {"step_graph": 1, "evb_defer_molecules": 2}])
def test_schedule_options_leave_the_forces_alone(opts):
    """the switchable pieces of the step schedule at BASELINE.json configs[2]: same Hamiltonian, same Hellmann-Feynman
    force, same 12-step trajectory as the default schedule (device-side solve by repeated squaring); among them the blend as one kernel behind the gather, the
    host's Jacobi (evb_device_solve = 0, the reference's row-cyclic sweep) and the device's round-robin Jacobi (= 2)"""
    spec = _synthetic(32, 8, 11, 64)
    if "default" not in _SCHED_DEFAULT:
        _SCHED_DEFAULT["default"] = _schedule_run(spec, {})
    a, b = _SCHED_DEFAULT["default"], _schedule_run(spec, opts)
    assert rel(b[0], a[0]) < 1e-13 and np.abs(b[1] - a[1]).max() < 1e-9 and np.abs(b[2] - a[2]).max() < 1e-9 * np.abs(a[2]).max()
    assert np.abs(b[3] - a[3]).max() < 1e-8 * np.abs(a[3]).max() and np.abs(b[4] - a[4]).max() < 1e-8


@pytest.mark.parametrize("name", all_golden("evb_") + ["md_evb_n6_d4_switch_s8"])
def test_device_side_enumeration_against_the_host(name):
    """f2: evb_conduct_proton_transfer_recursive / find_evb_reactive_neighbors (src/ms_evb.f90:498-620, :702-764) and the descriptor build behind
    them (:843-932) as ONE workgroup on the device (k_evb_plan) against the host running the same source (evb_device_plan = 0, the checker): diabat count,
    proton_log and coupling_matrix integer-equal, Hamiltonian and Hellmann-Feynman force to rounding of the atomics - on every EVB fixture of the real reference
    (whose own proton_log / coupling_matrix test_evb_against_reference compares with the default = device path) and, with three virtual ranks, on the rank layout."""
    g = golden(name)
    spec = system.from_golden(g)
    for ranks in (1, 3):
        res = {}
        for dev in (1, 0):
            h = api.MsEvbHip(spec)
            h.set_option("evb_device_plan", dev)
            if ranks > 1:
                h.set_option("shard_principal", 1)
                h.comm_set_virtual(1, ranks)
            e, F = h.ms_evb_calculate_total_force_energy()
            e2, F2 = h.ms_evb_calculate_total_force_energy()      # (a speculative call on the tree the first one left)
            st = h.evb_state()
            res[dev] = (e, F, st, e2, F2, h.info("host_dfs_calls"), h.info("device_plans"), h.info("spec_steps"))
            h.close()
        a, b = res[1], res[0]
        assert a[5] == 0 and a[6] >= 1 and b[5] >= 1 and b[6] == 0, (a[5:], b[5:])
        assert a[2]["D"] == b[2]["D"] and np.array_equal(a[2]["proton_log"], b[2]["proton_log"]) and np.array_equal(a[2]["coupling_matrix"], b[2]["coupling_matrix"])
        assert np.abs(a[2]["H"] - b[2]["H"]).max() < 1e-10 * max(np.abs(b[2]["H"]).max(), 1.0)
        assert rel(a[0], b[0]) < 1e-13 and np.abs(a[1] - b[1]).max() < 1e-9 and rel(a[3], a[0]) < 1e-13 and np.abs(a[4] - a[1]).max() < 1e-9


def test_step_graph_replays_the_force_call():
    """The enqueue part of a speculative MS-EVB call as ONE captured hipGraph (option step_graph; off by default - on ROCm 7.2 it saves no host time).  At BASELINE.json configs[2] size:
    30 MD steps + host-array force calls; nearly every call must have gone out as a graph launch, NO replay may have differed from what the direct enqueue
    would have issued (the library compares a hash of every launch's arguments), and energies / end state must equal the run without the graph."""
    spec = _synthetic(32, 8, 11, 64)
    out = {}
    for g in (1, 0):
        h = api.MsEvbHip(spec)
        h.set_option("step_graph", g)
        h.set_velocities(synth.maxwell_boltzmann(spec.mass, 300.0, 77))
        e0, F0 = h.ms_evb_calculate_total_force_energy()
        e1, F1 = h.ms_evb_calculate_total_force_energy()      # (the second host-array call replays or captures the non-fused variant)
        e2, F2 = h.ms_evb_calculate_total_force_energy()
        en = h.md_integrate_atomic(30, spec.delta_t, use_evb=True)
        out[g] = (e0, e1, e2, F2, en, h.get_positions(), int(h.info("graph_launches")), int(h.info("graph_captures")), int(h.info("graph_mismatches")), int(h.info("spec_misses")), int(h.info("evb_calls")))
        h.close()
    a, b = out[1], out[0]
    assert b[6] == 0 and b[7] == 0      # off means off
    assert a[8] == 0, "replays differed from the direct enqueue: %d" % a[8]
    assert a[6] >= a[10] - 3 * (a[9] + 1) - 8 and a[7] <= 2 * (a[9] + 2), (a[6:], "graph launches / captures / mismatches / speculation misses / calls")
    assert rel(a[0], b[0]) < 1e-13 and rel(a[1], a[0]) < 1e-13 and rel(a[2], a[0]) < 1e-13
    assert np.abs(a[3] - b[3]).max() < 1e-9
    assert np.abs(a[4][:, 0] - b[4][:, 0]).max() < 1e-9 * np.abs(b[4][:, 0]).max() and np.abs(a[5] - b[5]).max() < 1e-8


@pytest.mark.parametrize("stores", [0, 1])
def test_own_lds_transform_against_rocfft_and_reference(stores):
    """csrc/fft_lds.hip (64^3 grids: three kernels for FFT . CB . IFFT, src/pme.f90:73-125) against the rocFFT path (option fft_own=0) and the
    real reference on evb_n6_d4_k64_s18: potential, reciprocal energy and force, then the whole MS-EVB call (the Hellmann-Feynman force goes
    through a second solve, of the blended charge grid)"""
    g = golden("evb_n6_d4_k64_s18")
    res = []
    for own in (1, 0):
        h = api.MsEvbHip(system.from_golden(g))
        h.set_option("fft_own", own)
        h.set_option("evb_element_stores", stores)
        h.construct_verlet_list()
        h.zero_forces()
        er = h.pme_reciprocal_space_energy_force()
        Q, th, u, fr = h.pme_state()
        Fr = h.get_forces()
        e, F = h.ms_evb_calculate_total_force_energy()
        assert rel(e, g["E_adiabatic"][0]) < E_RTOL and np.abs(F - g["F_adiabatic"]).max() < F_TOL, own
        res.append((er, th, Fr, e, F))
        h.close()
    assert rel(res[0][0], res[1][0]) < 1e-14 and np.abs(res[0][1] - res[1][1]).max() < 1e-13 * np.abs(res[1][1]).max()
    assert np.abs(res[0][2] - res[1][2]).max() < 1e-11 and rel(res[0][3], res[1][3]) < 1e-14 and np.abs(res[0][4] - res[1][4]).max() < 1e-10


def test_own_128_transform_against_rocfft():
    """csrc/fft_lds.hip, 128^3 grids (BASELINE.json configs[4]'s grid: five kernels for FFT . CB . IFFT, src/pme.f90:73-125, a 128-point line as
    8 x 16 across 8 threads) against the rocFFT path (option fft_own=0) on a 10 125-atom box: potential, reciprocal energy and force, and the
    MS-EVB call, whose Hellmann-Feynman force goes through a second solve (of the blended charge grid).  The rocFFT path itself is pinned to the
    oracle at this grid size by test_config_C_against_oracle."""
    spec = _synthetic(15, 4, 5, 128)
    res = []
    for own in (1, 0):
        h = api.MsEvbHip(spec)
        h.set_option("fft_own", own)
        h.construct_verlet_list()
        h.zero_forces()
        er = h.pme_reciprocal_space_energy_force()
        Q, th, u, fr = h.pme_state()
        Fr = h.get_forces()
        e, F = h.ms_evb_calculate_total_force_energy()
        res.append((er, th, Fr, e, F))
        h.close()
    a, b = res
    assert np.abs(b[1]).max() > 1e-3 and np.abs(a[1] - b[1]).max() < 1e-13 * np.abs(b[1]).max()
    assert rel(a[0], b[0]) < 1e-13 and np.abs(a[2] - b[2]).max() < 1e-11 and rel(a[3], b[3]) < 1e-13 and np.abs(a[4] - b[4]).max() < 1e-10


@pytest.mark.parametrize("name", ["std_n6_s3", "std_n6_k20_s13"])
def test_pme_stage_entry_points_against_reference(name):
    """grid_Q / modify_Q_grid / derivative_grid_Q (src/pme.f90:184, :275, :346) at the reference's own granularity (host arrays and SCALED
    coordinates in and out - what reactive_pb_nn_md_amd/fortran/msevb_hip_stages.f90 binds) against the real reference's grids and forces"""
    g = golden(name)
    spec = system.from_golden(g)
    h = api.MsEvbHip(spec)
    u, q = g["xyz_scale"], spec.charge
    Q = h.grid_Q(q, u)
    assert np.abs(Q - g["Q_grid"]).max() < 1e-14
    # take five atoms off the grid and put them back with twice their charge: equals the full spread of the modified charges
    idx = np.array([0, 7, 100, 311, spec.n_atom - 1])
    Q2 = h.modify_Q_grid(Q.copy(), q[idx], u[idx], -1)
    Q2 = h.modify_Q_grid(Q2, 2.0 * q[idx], u[idx], +1)
    q2 = q.copy(); q2[idx] *= 2.0
    assert np.abs(Q2 - h.grid_Q(q2, u)).max() < 1e-13
    small = np.where(np.abs(q) > 1e-6, q, 0.0)      # modify_Q_grid skips |q| <= 1e-6 (src/pme.f90:296), grid_Q does not
    assert np.abs(h.modify_Q_grid(np.zeros_like(Q), q, u, +1) - h.grid_Q(small, u)).max() < 1e-14
    F = h.derivative_grid_Q(g["theta_conv_Q"], q, u, g["kk"], float(g["conv_e2A_kJmol"][0]))
    assert np.abs(F - g["F_recip"]).max() < F_TOL
    F1 = h.derivative_grid_Q(g["theta_conv_Q"], q[5:6], u[5:6], g["kk"], float(g["conv_e2A_kJmol"][0]))      # one atom per call, as the reference calls it
    assert np.abs(F1[0] - g["F_recip"][5]).max() < F_TOL
    h.close()


@pytest.mark.parametrize("stores", [0, 1])
@pytest.mark.parametrize("name", all_golden("evb_"))
def test_evb_against_reference(name, stores):
    """stores = 0: default path (one stencil gather of the blended potential, no per-element forces);
    stores = 1: F_dd / F_pd of every Hamiltonian element kept, as the reference does (evb_forces_store)"""
    g = golden(name)
    h = api.MsEvbHip(system.from_golden(g))
    h.set_option("evb_element_stores", stores)
    e, F = h.ms_evb_calculate_total_force_energy()
    production_kernels(h, name)
    st = h.evb_state()
    D = int(g["n_diabat"][0])
    assert st["D"] == D and ("_d16_" not in name or D == 16)
    assert np.array_equal(st["proton_log"][:, :g["evb_diabat_proton_log"].shape[1], :], g["evb_diabat_proton_log"])
    assert np.array_equal(st["coupling_matrix"], g["evb_diabat_coupling_matrix"])
    Href = g["evb_hamiltonian"]
    assert np.abs(st["H"] - Href).max() < E_RTOL * np.abs(Href).max()
    assert np.allclose(st["H"], np.triu(st["H"]))                  # upper triangle, tree couplings only
    assert rel(e, g["E_adiabatic"][0]) < E_RTOL and np.abs(F - g["F_adiabatic"]).max() < F_TOL
    assert st["principle_diabat"] == int(g["principle_diabat"][0]) and st["new_hydronium_molecule"] == int(g["new_hydronium_molecule"][0])
    assert abs(np.linalg.norm(st["eigvec"]) - 1.0) < 1e-12
    if stores == 0:
        with pytest.raises(api.MsevbError):
            h.evb_element_force(1, 1)
    if stores and "evb_forces_store" in g:
        lk, sto = g["evb_forces_lookup_index"], g["evb_forces_store"]
        for i in range(D):
            for j in range(i, D):
                if lk[i, j] > 0:
                    assert np.abs(h.evb_element_force(i + 1, j + 1) - sto[lk[i, j] - 1]).max() < F_TOL, (i, j)
    h.close()


@pytest.mark.parametrize("name", ["evb_n6_d4_s2", "evb_n6_d8_s1", "evb_n6_d4_k64_s18"])
def test_evb_translated_across_the_periodic_boundary(oracle_lib, name):
    """The fixtures hold the acid in the middle of the box.  Shifted by half / a quarter of a box (whole molecules, as the reference keeps them)
    the acid and its hydrogen-bonded waters straddle the periodic boundary: their PME stencils sit on both sides of the grid's wrap-around,
    which is where the stencil-pair sums of the diabats' reciprocal space (k_evb_recip_pairs) indexed their table with ONE wrap instead of two
    (round 3: a wrong table element, or - first element of the buffer - a read in front of it; found with the guard allocator).
    Checker: the CPU oracle (bit-pinned to the reference) on the SAME shifted coordinates.  The reference's numbers are not translation
    invariant to rounding: .gro-quantised coordinates put pairs exactly on r_c, and which side they fall is decided by the last bit (the oracle
    and the GPU both move by -0.19879676938 kJ/mol on evb_n6_d8_s1 at half a box); at zero shift the oracle equals the fixture."""
    import dataclasses
    g = golden(name)
    spec = system.from_golden(g)
    L = np.diag(spec.box) if np.ndim(spec.box) == 2 else np.asarray(spec.box)
    for frac in (0.5, 0.25):
        xyz = spec.xyz + frac * L
        for m in range(spec.n_mole):      # wrap molecule by molecule (first atom decides): molecules stay whole
            a0, n = spec.mol_first[m], spec.mol_natom[m]
            xyz[a0:a0 + n] -= L * np.floor(xyz[a0] / L)
        sp = dataclasses.replace(spec, xyz=xyz)
        o = oracle_lib.Oracle(sp)
        o.verlet_build()
        r = o.evb()
        h = api.MsEvbHip(sp)
        e, F = h.ms_evb_calculate_total_force_energy()
        st = h.evb_state()
        assert st["D"] == r["D"] == int(g["n_diabat"][0])
        Ho = np.triu(np.array(r["H"]))
        assert np.abs(st["H"] - Ho).max() < 1e-10 * np.abs(Ho).max(), frac
        assert rel(e, r["E"]) < 1e-10 and np.abs(F - r["F"]).max() < 1e-8, frac
        assert abs(r["E"] - g["E_adiabatic"][0]) < 0.5      # (the shifted system is the fixture's system)
        h.close()


@pytest.mark.parametrize("sapt", [False, True])
def test_half_list_lds_pair_kernel_against_oracle(oracle_lib, sapt):
    """The half-list LDS pair kernel (one workgroup per cell, tile-local list entries; the default whenever the box has >= 4 cells per
    axis - the small reference fixtures do not, they run the full-list kernel) at 1.5 x the size of BASELINE.json configs[1] (17^3 sites: five
    cells of 10.6 A per axis, so that the 14-cell half neighbourhood fits the flat-list kernel's 2560-atom tile; at 15^3 sites the four cells of
    11.7 A per axis do not and the full-list kernel runs): its list decodes
    to the oracle's pair set, forces / energies equal the oracle's AND the full-list kernel's, with and without the rolling pruned rows,
    also with a live SAPT (Buckingham + Tang-Toennies) term."""
    s = synth.make_system(17, n_diabat=2, seed=7)
    ff = forcefield.ch3so3h_water_forcefield()
    if sapt:
        ff = forcefield.with_test_sapt(ff)
    p = forcefield.build_params(ff, moltype_order=["h2o", "so3h"])
    spec = system.from_synth(s, p, rc=9.0, rv=10.5, pme_grid=32)
    o = oracle_lib.Oracle(spec, threads=8)
    o.verlet_build()
    vpo, nlo = o.verlet_get()
    rows_o = [np.sort(nlo[vpo[i]:vpo[i + 1]]) + 1 for i in range(spec.n_atom)]      # the oracle's arrays are 0-based
    Eo, Fo = o.pair()
    Ei, Fi = o.intra_nonbonded()
    e_tot, _E, F_tot = o.total()
    Fk = {}
    for tiles in (1, 0):
        h = api.MsEvbHip(spec)
        h.set_option("pair_tiles", tiles)
        h.construct_verlet_list()
        vp, nl = h.get_verlet_list()
        assert np.array_equal(vp, vpo + 1)
        assert all(np.array_equal(nl[vp[i] - 1:vp[i + 1] - 1], rows_o[i]) for i in range(spec.n_atom))      # the same set of pairs
        for prune in (0, 1):
            h.set_option("pair_prune", prune)
            h.zero_forces()
            e = h.real_space_energy_force()
            assert np.abs(h.get_forces() - (Fo + Fi)).max() < F_TOL, (tiles, prune)
            assert rel(e[0], Eo[0]) < E_RTOL and abs(e[1] - Eo[1]) < E_RTOL * max(abs(Eo[1]), 1.0), (tiles, prune, e, Eo)
        if tiles:      # with the pruned list on, the flat-list kernel serves both force fields (round 4: the SAPT / Tang-Toennies branch lives in k_pair_flat too)
            assert h.info("pair_kernel") == 2
        # the full force call three times at the same positions: the first writes the rolling pruned rows, the others run on them
        for rep in range(3):
            en, F = h.calculate_total_force_energy()
            assert rel(en["potential"], e_tot) < E_RTOL and np.abs(F - F_tot).max() < F_TOL, (tiles, rep)
        Fk[tiles] = F
        h.close()
    assert np.abs(Fk[1] - Fk[0]).max() < F_TOL


def test_tang_toennies_rows_on_table_boundaries(oracle_lib):
    """The Tang-Toennies damping table is a STEP function of B r (row = ceil(B r / 50 * 1000), no interpolation: src/pair_int_real_space.f90:674), so a
    pair whose B r lies on a row boundary takes the row the last bit of r picks.  Coordinates quantised to 0.01 A (what a .gro holds) put dozens of
    pairs exactly there (H...H at multiples of 0.1 A with B = 3.5).  All three pair kernels - k_pair (full list), k_pair_half (tile rows) and
    k_pair_flat - must take the row the reference's arithmetic on the raw coordinates gives (oracle = reference, bit-identical on the fixtures):
    a wrong row is an error of 1e-4 kJ/mol/A, five orders above the tolerance (VERDICT r04 'weak' 2)."""
    s = synth.make_system(17, n_diabat=2, seed=11)
    s.xyz = np.round(s.xyz, 2)
    p = forcefield.build_params(forcefield.with_test_sapt(forcefield.ch3so3h_water_forcefield()), moltype_order=["h2o", "so3h"])
    spec = system.from_synth(s, p, rc=9.0, rv=10.5, pme_grid=32)
    o = oracle_lib.Oracle(spec, threads=8)
    o.verlet_build()
    vpo, nlo = o.verlet_get()
    # how many pairs inside the cutoff sit on a row boundary (the test must not pass for lack of them)
    i_of = np.repeat(np.arange(spec.n_atom), np.diff(vpo))
    d = spec.xyz[i_of] - spec.xyz[nlo]
    L = np.diag(spec.box) if np.ndim(spec.box) == 2 else np.asarray(spec.box)
    d -= L * np.floor(d / L + 0.5)
    r = np.sqrt((d * d).sum(1))
    ti, tj = spec.atype[i_of], spec.atype[nlo]
    live = (p.vdw_type[ti, tj] == 1) & (r < spec.rc)
    xi = p.vdw_param[ti, tj, 1] * r / 50.0 * 1000.0
    n_tie = int((live & (np.abs(xi - np.rint(xi)) <= 1e-9 * np.maximum(xi, 1.0))).sum())
    assert n_tie >= 20, n_tie
    Eo, Fo = o.pair()
    Ei, Fi = o.intra_nonbonded()
    seen = set()
    for opts in ({"pair_tiles": 0}, {"pair_flat": 0}, {"pair_flat": 0, "pair_prune": 0}, {}):
        h = api.MsEvbHip(spec)
        for k, v in opts.items():
            h.set_option(k, v)
        h.construct_verlet_list()
        for rep in range(2):      # (the second call of the pruning kernels runs on the list the first one wrote)
            h.zero_forces()
            e = h.real_space_energy_force()
            assert np.abs(h.get_forces() - (Fo + Fi)).max() < F_TOL, (opts, rep, np.abs(h.get_forces() - (Fo + Fi)).max())
            assert rel(e[0], Eo[0]) < E_RTOL and abs(e[1] - Eo[1]) < E_RTOL * max(abs(Eo[1]), 1.0), (opts, rep)
        seen.add(int(h.info("pair_kernel")))
        h.close()
    assert seen == {0, 1, 2}      # every kernel was in the comparison


@pytest.mark.parametrize("name,evb", [("md_std_n6_s4", False), ("md_evb_n6_d2_s5", True), ("md_std_n6_frz_s14", False), ("md_evb_n12_d4_s29", True)])
def test_md_trajectory_against_reference(name, evb):
    g = golden(name)
    spec = system.from_golden(g)
    h = api.MsEvbHip(spec)
    h.set_velocities(g["md_velocity0"])
    if evb:
        h.ms_evb_calculate_total_force_energy(want_force=False)
    else:
        h.calculate_total_force_energy(want_force=False)
    production_kernels(h, name)
    assert np.abs(h.get_forces() - g["md_F0"]).max() < F_TOL
    n = int(g["md_nsteps"][0])
    en = h.md_integrate_atomic(n, spec.delta_t, use_evb=evb)
    assert np.abs(en[:, 0] - g["md_energy"][:, 0]).max() < 1e-9 * np.abs(g["md_energy"][:, 0]).max()
    assert np.abs(en[:, 1] - g["md_energy"][:, 1]).max() < 1e-8 * np.abs(g["md_energy"][:, 1]).max()
    assert np.abs(h.get_positions() - g["md_xyz"]).max() < 1e-10
    assert np.abs(h.get_velocities() - g["md_velocity"]).max() < 1e-9
    assert np.abs(h.get_forces() - g["md_force"]).max() < 1e-7
    h.close()


@pytest.mark.parametrize("name,evb", [("md_std_n6_rebuild_s21", False), ("md_evb_n6_d4_rebuild_s24", True), ("md_evb_n6_d4_drop_s22", True)])
def test_md_trajectory_across_rebuilds_against_reference(name, evb):
    """60 reference steps at 600 K that cross 6-7 Verlet-list rebuilds (update_verlet_displacements, src/general_routines.f90:1259-1337;
    the flag raised by one force call makes the NEXT one rebuild, src/total_energy_forces.f90:30-41) and changes of the diabat count
    (4 -> 5 -> 6 and 4 -> 1).  Step by step: energies, flag_verlet_list and the diabat count of the real reference; then the end state."""
    g = golden(name)
    spec = system.from_golden(g)
    h = api.MsEvbHip(spec)
    h.set_velocities(g["md_velocity0"])
    if evb:
        h.ms_evb_calculate_total_force_energy(want_force=False)
    else:
        h.calculate_total_force_energy(want_force=False)
    assert np.abs(h.get_forces() - g["md_F0"]).max() < F_TOL
    n = int(g["md_nsteps"][0])
    flags, nd = g["md_verlet_flag"], g["md_n_diabat"]
    assert flags.sum() >= 6 and (not evb or len(set(nd)) > 1)      # the fixture does cross the events
    for s in range(n):
        en = h.md_integrate_atomic(1, spec.delta_t, use_evb=evb)
        assert abs(en[0, 0] - g["md_energy"][s, 0]) < 1e-9 * abs(g["md_energy"][s, 0]), s
        assert abs(en[0, 1] - g["md_energy"][s, 1]) < 1e-8 * abs(g["md_energy"][s, 1]), s
        assert h.verlet_flag() == flags[s], s
        if evb:
            assert h.evb_state()["D"] == nd[s], s
    assert np.abs(h.get_positions() - g["md_xyz"]).max() < 1e-10
    assert np.abs(h.get_velocities() - g["md_velocity"]).max() < 1e-9
    assert np.abs(h.get_forces() - g["md_force"]).max() < 1e-7
    if evb:      # the DEFAULT schedule (speculative step, device-side solve, blend phase enqueued before the host has seen the verdict) went through tree changes:
        # calls that started from a stale tree were noticed and repeated, and the trajectory above is still the reference's
        assert h.info("spec_steps") > 0 and h.info("spec_misses") >= len(set(nd)) - 1, (h.info("spec_steps"), h.info("spec_misses"))
        # f2: every enumeration of this trajectory (first call, every tree change) ran on the device - no depth-first search, no descriptor build on the host
        assert h.info("host_dfs_calls") == 0 and h.info("device_plans") >= len(set(nd)), (h.info("host_dfs_calls"), h.info("device_plans"))
    h.close()
    # the same trajectory in ONE call (device-resident loop, no host look at the flag between steps)
    h = api.MsEvbHip(spec)
    h.set_velocities(g["md_velocity0"])
    (h.ms_evb_calculate_total_force_energy if evb else h.calculate_total_force_energy)(want_force=False)
    en = h.md_integrate_atomic(n, spec.delta_t, use_evb=evb)
    assert np.abs(en[:, 0] - g["md_energy"][:, 0]).max() < 1e-9 * np.abs(g["md_energy"][:, 0]).max()
    assert np.abs(h.get_positions() - g["md_xyz"]).max() < 1e-10 and np.abs(h.get_forces() - g["md_force"]).max() < 1e-7
    h.close()


def test_long_trajectory_to_the_references_own_stop():
    """tests/golden/long_evb_n12_liquid_s40 (oracle/make_collapse_trace.py): the REAL reference on a liquid-like 12^3-site box, 4 planted diabats, 300 K start -
    it runs 319 steps and then STOPS ITSELF ('force on atom 2568 is too big', src/md_integration.f90:523): the diabat tree dissolves (D = 4 -> 5 -> 1 around
    step 305) and the acid proton, which has no Lennard-Jones wall outside the EVB complex, falls onto an oxygen of its own molecule (|F|max 470 -> 4 000
    kJ/mol/A within six steps).  The same collapse ends bench.py's box after ~ 300 steps: it is the force field's, not the library's.
    Here the GPU path walks the reference's history step by step - potential and kinetic energy, Verlet flag, diabat count, largest force - up to the
    collapse and then ends the run the way the reference does."""
    g = golden("long_evb_n12_liquid_s40")
    s = synth.make_system(int(g["gen_n_side"]), n_diabat=int(g["gen_n_diabat"]), seed=int(g["gen_seed"]), liquid_like=True)
    assert np.array_equal(s.xyz, g["gen_xyz"])
    p = forcefield.build_params(forcefield.ch3so3h_water_forcefield(), moltype_order=["h2o", "so3h"])
    spec = system.from_synth(s, p, rc=float(g["rc"]), rv=float(g["rv"]), pme_grid=int(g["pme_grid"]), nslist=int(g["nslist"]))
    assert not bool(g["reference_finished"]) and "is too big" in str(g["reference_last_output"])
    h = api.MsEvbHip(spec)
    h.set_velocities(g["md_velocity0"])
    h.ms_evb_calculate_total_force_energy(want_force=False)
    n = len(g["trace_step"])
    worst = 0.0
    for k in range(n):
        en = h.md_integrate_atomic(1, spec.delta_t, use_evb=True)
        tol = 1e-9 if k < 100 else 1e-7 if k < 250 else 1e-5      # (fp64 atomics sum in another order than the reference's loops: the difference grows with the trajectory's own instability)
        de = abs(en[0, 0] - g["trace_epot"][k]) / abs(g["trace_epot"][k])
        dk = abs(en[0, 1] - g["trace_ekin"][k]) / abs(g["trace_ekin"][k])
        worst = max(worst, de / tol, dk / tol)
        assert de < tol and dk < tol, (k, de, dk)
        assert h.verlet_flag() == g["trace_verlet_flag"][k], k
        assert h.evb_state()["D"] == g["trace_n_diabat"][k], k
        fmax = np.sqrt((h.get_forces() ** 2).sum(1).max())
        assert abs(fmax - g["trace_fmax"][k]) < (1e-6 if k < 250 else 1e-3) * g["trace_fmax"][k], (k, fmax, g["trace_fmax"][k])
    assert g["trace_fmax"][-1] > 5 * g["trace_fmax"][-8] and g["trace_n_diabat"][-1] == 1      # the fixture ends in the collapse
    with pytest.raises(api.MsevbError, match="too big|not finite"):      # ... and so does the run: the reference stopped in its step 320
        for _k in range(4):
            h.md_integrate_atomic(1, spec.delta_t, use_evb=True)
    assert h.info("host_dfs_calls") == 0
    h.close()


def _gathered_payloads(spec, n_ranks, opts=None):
    """the R payloads an all-gather would deliver (msevb_evb_payload_pack layout), made from ONE context that runs the multi-rank device path as rank 0 of 1 (its
    energy-fetch kernel then writes every diabat's entries into the one slot of the buffer): (payloads [R, P], D, state of that context, its energy and force)"""
    import ctypes as C
    h = api.MsEvbHip(spec)
    for k, v in (opts or {}).items():
        h.set_option(k, v)
    h.comm_set_virtual(0, 1)
    h.ms_evb_calculate_total_force_energy(want_force=False)
    D = h.evb_state()["D"]
    P1 = h.evb_payload_size(D, 1)
    h.comm_set_virtual_payload(np.zeros(P1))      # (rank 0 of 1 owns every slot: nothing of these zeros survives the call)
    e, F = h.ms_evb_calculate_total_force_energy()
    st = h.evb_state()
    one = h.comm_get_payload(P1)
    Hp = np.zeros((D, D))
    parent = np.ascontiguousarray(st["coupling_matrix"] - 1, dtype=np.int32)
    parent[0] = -1
    h._chk(h.L.msevb_evb_payload_unpack(D, 1, one.ctypes.data_as(C.POINTER(C.c_double)), parent.ctypes.data_as(C.POINTER(C.c_int)), Hp.ctypes.data_as(C.POINTER(C.c_double)), None))
    P = h.evb_payload_size(D, n_ranks)
    pay = np.zeros((n_ranks, P))
    for r in range(n_ranks):
        h._chk(h.L.msevb_evb_payload_pack(D, n_ranks, r, Hp.ctypes.data_as(C.POINTER(C.c_double)), parent.ctypes.data_as(C.POINTER(C.c_int)), None, pay[r].ctypes.data_as(C.POINTER(C.c_double))))
    h.close()
    return pay, D, st, e, F


def _virtual_allgather_run(spec, n_ranks, shard_principal, opts=None):
    """what R ranks with a communicator do, played on one GPU: every virtual rank is handed the R gathered payloads (msevb_comm_set_virtual_payload) and runs its force
    call - Hamiltonian solved on the device from the gathered entries, blend pre-enqueued, no caller-side merge.  Returns per rank (epot, partial force, state, ...)."""
    pay, D, st1, e1, F1 = _gathered_payloads(spec, n_ranks, opts)
    out = []
    for r in range(n_ranks):
        h = api.MsEvbHip(spec)
        for k, v in (opts or {}).items():
            h.set_option(k, v)
        if shard_principal:
            h.set_option("shard_principal", 1)
        h.comm_set_virtual(r, n_ranks)
        h.comm_set_virtual_payload(pay)
        e, F = h.ms_evb_calculate_total_force_energy()
        e2, F2 = h.ms_evb_calculate_total_force_energy()      # (a second, speculative call: nothing of the first one may be left behind)
        assert np.abs(F2 - F).max() < 1e-9 and rel(e2, e) < 1e-13, (e, e2)
        own = h.comm_get_payload(pay.size).reshape(pay.shape)[r]
        assert np.abs(own[:-9] - pay[r][:-9]).max() < 1e-10 * np.abs(pay).max(), "the rank's own slot is not what the one-rank run computed for the same diabats"      # (the entries of its diabats; H'_11 counts from rank 0's slot)
        out.append((e, F, h.evb_state(), int(h.info("spec_steps"))))
        h.close()
    return out, (e1, F1, st1)


@pytest.mark.parametrize("n_ranks,shard", [(2, 1), (3, 0), (4, 1), (8, 1)])
def test_device_side_solve_on_the_gathered_hamiltonian(n_ranks, shard):
    """More than one rank (diagonalize_evb_hamiltonian src/ms_evb.f90:242-330 on entries that meet in an all-gather): the rank's energy-fetch kernel
    assembles the entries of ITS diabats into its payload slot, k_evb_solve_gathered builds H' from the R gathered payloads, finds the ground state and the rank's blend
    coefficients, the blend phase is pre-enqueued - no host Jacobi, no host turn.  Played with virtual ranks and the test as the all-gather: every rank must arrive at the
    single-rank Hamiltonian and eigenvector, and the ranks' partial forces must add up to the single-rank Hellmann-Feynman force (what the all-reduce sums)."""
    g = golden("evb_n6_d8_s1")
    spec = system.from_golden(g)
    h = api.MsEvbHip(spec)
    e1, F1 = h.ms_evb_calculate_total_force_energy()
    st1 = h.evb_state()
    h.close()
    res, one = _virtual_allgather_run(spec, n_ranks, shard)
    assert rel(one[0], e1) < 1e-13 and np.abs(one[1] - F1).max() < 1e-9      # rank 0 of 1 on the multi-rank path = the one-rank path
    Fsum = np.zeros_like(F1)
    for r, (e, F, st, nspec) in enumerate(res):
        assert st["D"] == st1["D"] and nspec >= 1
        H1 = st1["H"]
        # off-diagonal elements and DIFFERENCES of diagonal elements do not depend on the principal energy E0, which a virtual rank holds only in part
        assert np.abs(np.triu(st["H"], 1) - np.triu(H1, 1)).max() < 1e-10 * np.abs(H1).max(), (r, np.triu(st["H"], 1)[np.triu(H1, 1) != 0], np.triu(H1, 1)[np.triu(H1, 1) != 0], np.diag(st["H"]), np.diag(H1), st["eigvec"], st1["eigvec"])
        dd, d1 = np.diag(st["H"]) - st["H"][0, 0], np.diag(H1) - H1[0, 0]
        assert np.abs(dd - d1).max() < 1e-10 * np.abs(H1).max()
        assert min(np.abs(st["eigvec"] - st1["eigvec"]).max(), np.abs(st["eigvec"] + st1["eigvec"]).max()) < 1e-10
        Fsum += F
    assert np.abs(Fsum - F1).max() < 1e-9
    assert np.abs(Fsum - g["F_adiabatic"]).max() < F_TOL      # ... which is the reference's


@pytest.mark.parametrize("n_ranks", [2, 3, 8])
def test_diabat_sharding_is_a_partition(n_ranks):
    """virtual ranks: every Hamiltonian element is produced by exactly one rank and equals the single-rank result, and the ranks'
    partial Hellmann-Feynman forces (what msevb_comm_allreduce_forces sums) add up to the single-rank force"""
    g = golden("evb_n6_d8_s1")
    spec = system.from_golden(g)
    h = api.MsEvbHip(spec)
    h.set_option("evb_element_stores", 1)
    _e, Fblend = h.ms_evb_calculate_total_force_energy()
    full = h.evb_state()
    D = full["D"]
    Ffull = {(i, j): h.evb_element_force(i, j) for i in range(1, D + 1) for j in range(i, D + 1) if i == j or full["coupling_matrix"][j - 1] == i}
    assert np.abs(h.evb_partial_blend(full["eigvec"]) - Fblend).max() < 1e-10      # one rank: the partial blend is the blend
    h.close()
    Hsum = np.zeros((D, D))
    Fsum = np.zeros_like(Fblend)
    for r in range(n_ranks):
        hr = api.MsEvbHip(spec)
        hr.set_option("evb_element_stores", 1)
        hr.comm_set_virtual(r, n_ranks)
        hr.ms_evb_calculate_total_force_energy(want_force=False)
        st = hr.evb_state()
        for d in range(2, D + 1):
            owner = (d - 2) % n_ranks
            p = full["coupling_matrix"][d - 1]
            if owner == r:
                assert rel(st["H"][d - 1, d - 1], full["H"][d - 1, d - 1]) < 1e-13 and rel(st["H"][p - 1, d - 1], full["H"][p - 1, d - 1]) < 1e-13
                assert np.abs(hr.evb_element_force(d, d) - Ffull[(d, d)]).max() < 1e-10
                assert np.abs(hr.evb_element_force(p, d) - Ffull[(p, d)]).max() < 1e-10
                Hsum[d - 1, d - 1] += st["H"][d - 1, d - 1]
            else:
                assert st["H"][d - 1, d - 1] == 0.0 and np.abs(hr.evb_element_force(d, d)).max() == 0.0
        Fsum += hr.evb_partial_blend(full["eigvec"])
        hr.close()
    assert np.allclose(np.diag(Hsum)[1:], np.diag(full["H"])[1:], rtol=1e-13)
    assert np.abs(Fsum - Fblend).max() < 1e-10
    # the default path (no per-element stores: one gather of the rank's blended potential, self-cleaning delta store), twice per
    # rank so that a delta store left dirty by the first call would show in the second
    Fsum = np.zeros_like(Fblend)
    for r in range(n_ranks):
        hr = api.MsEvbHip(spec)
        hr.comm_set_virtual(r, n_ranks)
        for _rep in range(2):
            hr.ms_evb_calculate_total_force_energy(want_force=False)
            Fr = hr.evb_partial_blend(full["eigvec"])
        hr.ms_evb_calculate_total_force_energy(want_force=False)      # ... and once without the hook consuming it
        hr.ms_evb_calculate_total_force_energy(want_force=False)
        assert np.abs(hr.evb_partial_blend(full["eigvec"]) - Fr).max() < 1e-10
        Fsum += Fr
        hr.close()
    assert np.abs(Fsum - Fblend).max() < 1e-10


@pytest.mark.parametrize("stores", [0, 1])
@pytest.mark.parametrize("n_ranks", [2, 3, 8])
def test_sharded_principal_partial_sums(n_ranks, stores):
    """multi-rank layout with the principal diabat's pair loop / per-molecule terms / reciprocal-force accumulation split over the ranks
    (what msevb_comm_init switches on): the ranks' partial blends and partial H11 add up to the single-rank force and energy"""
    g = golden("evb_n6_d8_s1")
    spec = system.from_golden(g)
    h = api.MsEvbHip(spec)
    _e, Fblend = h.ms_evb_calculate_total_force_energy()
    full = h.evb_state()
    en_full, F_full = h.calculate_total_force_energy()
    h.close()
    Fsum, Esum = np.zeros_like(Fblend), 0.0
    for r in range(n_ranks):
        hr = api.MsEvbHip(spec)
        hr.set_option("evb_element_stores", stores)
        hr.set_option("shard_principal", 1)
        hr.comm_set_virtual(r, n_ranks)
        e_part, _ = hr.ms_evb_calculate_total_force_energy(want_force=False)      # virtual ranks return their share of H11
        Esum += e_part
        Fsum += hr.evb_partial_blend(full["eigvec"])
        # a full (unsharded) force call on the same context afterwards: the pruned pair rows of the rank's atom range must not be reused
        en_r, F_r = hr.calculate_total_force_energy()
        assert rel(en_r["potential"], en_full["potential"]) < 1e-13 and np.abs(F_r - F_full).max() < 1e-10
        hr.close()
    assert np.abs(Fsum - Fblend).max() < 1e-10
    assert rel(Esum, full["H"][0, 0]) < 1e-13


@pytest.mark.parametrize("evb", [False, True])
def test_blown_up_trajectory_is_an_error_not_a_fault(evb):
    """velocities 60x thermal: atoms collide, coordinates run away.  The reference's (and our) grid index arithmetic only covers atoms
    within one box image; the library must notice and return MSEVB_ERR_NUMERIC instead of indexing outside its buffers"""
    g = golden("evb_n6_d2_s1")
    spec = system.from_golden(g)
    h = api.MsEvbHip(spec)
    h.set_velocities(60.0 * synth.maxwell_boltzmann(spec.mass, 300.0, 9))
    if evb:
        h.ms_evb_calculate_total_force_energy(want_force=False)
    else:
        h.calculate_total_force_energy(want_force=False)
    with pytest.raises(RuntimeError, match="not finite|more than one box|too big|diabat|hydronium|too many"):
        for _ in range(40):
            h.md_integrate_atomic(25, spec.delta_t, use_evb=evb)
    # new coordinates clear the condition
    h.set_positions(spec.xyz)
    h.set_velocities(synth.maxwell_boltzmann(spec.mass, 300.0, 9))
    h.construct_verlet_list()
    if evb:
        e, _ = h.ms_evb_calculate_total_force_energy()
    else:
        e = h.calculate_total_force_energy()[0]["potential"]
    assert np.isfinite(e)
    h.close()


def test_principal_switch_against_reference():
    """src/ms_evb.f90:218-227.  The fixture's reference run switched the principal diabat in its first EVB force call and holds the
    state after it (in the reference's re-packed atom arrays).  From the state before the switch the library must arrive at that
    topology - atom types, charges, molecule records, the transferred proton made whole with its new molecule - and follow the
    reference's trajectory on it.  All 1-based values are the C-ABI's (Fortran numbering)."""
    g = golden("md_evb_n6_d4_switch_s8")
    post = system.from_golden(g)
    pre = pre_switch_spec(g)
    h = api.MsEvbHip(pre)
    h.ms_evb_calculate_total_force_energy(want_force=False)
    st = h.evb_state()
    assert st["principle_diabat"] != 1 and st["new_hydronium_molecule"] == post.hydronium + 1
    hops, calls = h.hop_log()      # the hop is logged with the force call it happened in (src/ms_evb.f90:331-335 writes it with its step)
    assert calls == 1 and hops.tolist() == [[1, pre.hydronium + 1, post.hydronium + 1]] and len(h.hop_log()[0]) == 0
    t = h.topology()
    assert t["hydronium_molecule"] == post.hydronium + 1
    t["mol_atom_index"] = t["mol_atom_index"] - 1
    perm = reference_atom_order(t)
    assert sorted(perm) == list(range(pre.n_atom))
    assert np.array_equal(t["mol_n_atom"], post.mol_natom) and np.array_equal(t["mol_type_index"] - 1, post.mol_type)
    assert np.array_equal(t["atom_type_index"][perm] - 1, post.atype)
    assert np.array_equal(t["charge"][perm], post.charge)
    assert np.abs(h.get_positions()[perm] - post.xyz).max() < 1e-12
    # the force call on the new topology against the reference's Hamiltonian, adiabatic energy and force
    e, F = h.ms_evb_calculate_total_force_energy()
    st = h.evb_state()
    D = int(g["n_diabat"][0])
    assert st["D"] == D and st["principle_diabat"] == 1 and st["new_hydronium_molecule"] == post.hydronium + 1
    assert np.abs(np.triu(st["H"]) - np.triu(g["evb_hamiltonian"])).max() < E_RTOL * np.abs(g["evb_hamiltonian"]).max()
    assert rel(e, g["E_adiabatic"][0]) < E_RTOL
    assert np.abs(F[perm] - g["F_adiabatic"]).max() < F_TOL
    # ... and the steps after it (the reference read its velocities in ITS atom order)
    vel = np.zeros_like(g["md_velocity0"])
    vel[perm] = g["md_velocity0"]
    h.set_velocities(vel)
    n = int(g["md_nsteps"][0])
    en = h.md_integrate_atomic(n, post.delta_t, use_evb=True)
    assert np.abs(en[:, 0] - g["md_energy"][:, 0]).max() < 1e-9 * np.abs(g["md_energy"][:, 0]).max()
    assert np.abs(en[:, 1] - g["md_energy"][:, 1]).max() < 1e-8 * np.abs(g["md_energy"][:, 1]).max()
    assert h.topology()["hydronium_molecule"] == int(g["md_hydronium_molecule"][0])
    assert np.abs(h.get_positions()[perm] - g["md_xyz"]).max() < 1e-10
    assert np.abs(h.get_velocities()[perm] - g["md_velocity"]).max() < 1e-9
    assert np.abs(h.get_forces()[perm] - g["md_force"]).max() < 1e-7
    h.close()


@pytest.mark.parametrize("evb", [False, True])
def test_list_row_overflow_repeats_the_force_call(evb):
    """a rebuild inside a force call does not wait for its row statistics; when a row outgrew the stride the call is repeated with
    a larger one.  Forced here with a stride of 64 entries (rows hold ~240)"""
    g = golden("evb_n6_d4_s2")
    spec = system.from_golden(g)
    h = api.MsEvbHip(spec)
    call = (lambda: h.ms_evb_calculate_total_force_energy()) if evb else (lambda: h.calculate_total_force_energy())
    e0, F0 = call()
    vp0, nl0 = h.get_verlet_list()
    h.set_option("nbr_cap", 64)          # also marks the list stale: the next force call rebuilds it (deferred statistics)
    e1, F1 = call()
    vp1, nl1 = h.get_verlet_list()
    assert np.array_equal(vp0, vp1) and np.array_equal(nl0, nl1)
    assert np.abs(F1 - F0).max() < 1e-10
    assert (rel(e1, e0) if evb else rel(e1["potential"], e0["potential"])) < 1e-13
    if evb:      # ... and inside the MD loop
        h.set_velocities(synth.maxwell_boltzmann(spec.mass, 300.0, 3))
        h.set_option("nbr_cap", 64)
        en1 = h.md_integrate_atomic(5, spec.delta_t, use_evb=True)
        h2 = api.MsEvbHip(spec)
        h2.set_velocities(synth.maxwell_boltzmann(spec.mass, 300.0, 3))
        h2.ms_evb_calculate_total_force_energy(want_force=False)
        en2 = h2.md_integrate_atomic(5, spec.delta_t, use_evb=True)
        assert np.abs(en1 - en2).max() < 1e-9 * np.abs(en2).max()
        h2.close()
    h.close()


@pytest.mark.parametrize("evb,frozen", [(False, False), (True, False), (True, True)])
def test_langevin_step_against_host_restatement(evb, frozen):
    """NVT: both kicks of a step follow langevin_integrator (src/md_integration.f90:186-228) with the library's counter-based
    normal deviates, which reactive_pb_nn_md_amd/langevin.py restates; x1 = x0 + v1 dt, then the centre-of-mass momentum goes"""
    from reactive_pb_nn_md_amd import langevin
    g = golden("evb_n6_d2_s1")
    spec = system.from_golden(g)
    h = api.MsEvbHip(spec)
    free = np.ones(spec.n_atom, dtype=bool)
    if frozen:      # atype_freeze (src/md_integration.f90:473, :512, :151-169): water oxygens neither kicked nor moved nor counted in P/N
        ti = h.topology()["atom_type_index"]
        fz = np.zeros(spec.params.n_atom_type, dtype=np.int32)
        fz[np.bincount(ti[spec.mass > 10.0]).argmax() - 1] = 1      # the most common heavy type: water oxygen
        h.set_frozen_types(fz)
        free = fz[ti - 1] == 0
        assert 0 < (~free).sum() < spec.n_atom
    gam, T, seed, dt = 5.0, 320.0, 12345678901, spec.delta_t
    for k, v in (("ensemble_nvt", 1), ("friction_coeff", gam), ("temperature", T), ("langevin_seed", seed)):
        h.set_option(k, v)
    v0 = synth.maxwell_boltzmann(spec.mass, 300.0, 4, ~free)
    h.set_velocities(v0)
    F0 = (h.ms_evb_calculate_total_force_energy() if evb else h.calculate_total_force_energy())[1]
    x0 = h.get_positions()
    h.md_integrate_atomic(1, dt, use_evb=evb)
    v1 = np.where(free[:, None], langevin.kick(v0, F0, spec.mass, dt, gam, T, langevin.normals(spec.n_atom, 0, seed)), v0)
    x1 = x0 + v1 * dt
    xg = h.get_positions()
    d = xg - x1
    assert np.abs(d - spec.box * np.round(d / spec.box)).max() < 1e-12      # molecules may have been shifted back into the box
    F1 = h.get_forces()
    v2 = np.where(free[:, None], langevin.kick(v1, F1, spec.mass, dt, gam, T, langevin.normals(spec.n_atom, 1, seed)), v1)
    v2[free] -= (spec.mass[free, None] * v2[free]).sum(0) / free.sum() / spec.mass[free, None]      # every free atom gives up P/N_free (src/md_integration.f90:125-177)
    if frozen:
        assert np.abs(d[~free] - spec.box * np.round(d[~free] / spec.box)).max() < 1e-13 and np.all(h.get_velocities()[~free] == 0.0)
    assert np.abs(h.get_velocities() - v2).max() < 1e-10
    # the deviates are standard normal and differ from kick to kick
    z = np.concatenate([langevin.normals(spec.n_atom, k, seed) for k in range(40)])
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1.0) < 0.02 and abs((z ** 4).mean() - 3.0) < 0.15
    h.close()


def test_langevin_kicks_against_reference_at_zero_temperature():
    """langevin_integrator (src/md_integration.f90:186-228) in the REAL reference with temperature 0 (the noise term vanishes): four steps of
    the friction + force kick, leap-frog positions, momentum removal - energies, positions, velocities and forces of the reference's run"""
    g = golden("md_std_n6_nvt0_s16")
    spec = system.from_golden(g)
    h = api.MsEvbHip(spec)
    for k, v in (("ensemble_nvt", 1), ("friction_coeff", 5.0), ("temperature", 0.0)):
        h.set_option(k, v)
    h.set_velocities(g["md_velocity0"])
    h.calculate_total_force_energy(want_force=False)
    assert np.abs(h.get_forces() - g["md_F0"]).max() < F_TOL
    n = int(g["md_nsteps"][0])
    en = h.md_integrate_atomic(n, spec.delta_t, use_evb=False)
    assert np.abs(en[:, 0] - g["md_energy"][:, 0]).max() < 1e-9 * np.abs(g["md_energy"][:, 0]).max()
    assert np.abs(en[:, 1] - g["md_energy"][:, 1]).max() < 1e-8 * np.abs(g["md_energy"][:, 1]).max()
    assert np.abs(h.get_positions() - g["md_xyz"]).max() < 1e-10 and np.abs(h.get_velocities() - g["md_velocity"]).max() < 1e-9
    assert np.abs(h.get_forces() - g["md_force"]).max() < 1e-7
    # the friction is really on: the same steps without it end elsewhere
    h2 = api.MsEvbHip(spec)
    h2.set_velocities(g["md_velocity0"])
    h2.calculate_total_force_energy(want_force=False)
    h2.md_integrate_atomic(n, spec.delta_t, use_evb=False)
    assert np.abs(h2.get_velocities() - g["md_velocity"]).max() > 1e-3
    h.close(); h2.close()


def test_monte_carlo_barostat_against_reference():
    """monte_carlo_barostat (src/md_integration.f90:231-386) of the REAL reference: two MD steps, then six (trial move, MD step) pairs as
    mc_sample orders them (:28-37), at 2 000 bar with a 0.4 % trial step.  The fixture holds the random numbers every move drew; replayed
    through barostat.MCBarostat on the GPU library they must give the reference's decisions (2 accepted, 4 rejected), box lengths,
    coordinates after every move and energies, and the same trajectory in between (msevb_set_box = periodic_box_change, src/pme.f90:634)."""
    from reactive_pb_nn_md_amd import barostat
    g = golden("npt_std_n6_s15")
    spec = system.from_golden(g)
    h = api.MsEvbHip(spec)
    h.set_velocities(g["md_velocity0"])
    h.calculate_total_force_energy(want_force=False)
    en = h.md_integrate_atomic(int(g["md_nsteps"][0]), spec.delta_t, use_evb=False)
    assert np.abs(en[:, 0] - g["md_energy"][:, 0]).max() < 1e-9 * np.abs(g["md_energy"][:, 0]).max()
    mol_atoms = [list(range(f, f + n)) for f, n in zip(spec.mol_first, spec.mol_natom)]
    b = barostat.MCBarostat(float(g["baro_pressure"][0]), float(g["baro_temperature"][0]), baroscale=float(g["baro_scale0"][0]))
    e_now = en[-1, 0]
    assert rel(e_now, g["baro_epot0"][0]) < E_RTOL
    box = float(g["baro_box0"][0])
    n_acc = 0
    for k in range(int(g["baro_n"][0])):
        b.uniforms = iter(g["baro_rand"][k])      # (trial length, Metropolis test); a move that is accepted with w < 0 leaves the second one to the next move
        acc, e_now = b.move(h, e_now, False, mol_atoms, spec.mass, spec.n_mole)
        new_box = float(g["baro_box"][k])
        assert acc == (abs(new_box - box) > 1e-12), k
        n_acc += acc
        box = new_box
        assert abs(float(np.asarray(h.spec.box)[0]) - box) < 1e-12 and rel(e_now, g["baro_epot"][k]) < 1e-10, k
        assert np.abs(h.get_positions() - g["baro_xyz"][k]).max() < 1e-10, k
        assert abs(b.baroscale - g["baro_scale"][k]) < 1e-15
        e_md = h.md_integrate_atomic(1, spec.delta_t, use_evb=False)
        assert rel(e_md[0, 0], g["baro_epot_md"][k]) < 1e-9, k
        e_now = e_md[0, 0]
    assert 0 < n_acc < int(g["baro_n"][0])      # the fixture holds accepted AND rejected moves
    assert np.abs(h.get_positions() - g["baro_end_xyz"]).max() < 1e-9 and np.abs(h.get_velocities() - g["baro_end_velocity"]).max() < 1e-8
    h.close()


def test_langevin_thermostat_reaches_its_temperature():
    """strong friction on an (initially much too hot) lattice of water: the kinetic temperature settles at the thermostat's.
    Water only - the example force field's acid proton carries no repulsion outside the EVB terms and collapses onto a water
    oxygen after a few hundred steps of plain MD, thermostat or not"""
    import dataclasses
    s = synth.make_system(6, n_diabat=1, seed=3)
    a0, a1 = s.mol_first[s.hydronium], s.mol_first[s.hydronium] + s.mol_natom[s.hydronium]
    keep = np.r_[0:a0, a1:s.n_atom]
    natom = np.delete(s.mol_natom, s.hydronium)
    s = dataclasses.replace(s, xyz=s.xyz[keep], atom_names=[s.atom_names[k] for k in keep], mol_names=[m for i, m in enumerate(s.mol_names) if i != s.hydronium],
                            mol_natom=natom, mol_first=np.concatenate([[0], np.cumsum(natom)[:-1]]).astype(np.int32), hydronium=-1)
    p = forcefield.build_params(forcefield.ch3so3h_water_forcefield(), moltype_order=["h2o", "so3h"])
    spec = system.from_synth(s, p, rc=7.0, rv=8.0, pme_grid=16, nslist=16)
    h = api.MsEvbHip(spec)
    for k, v in (("ensemble_nvt", 1), ("friction_coeff", 100.0), ("temperature", 300.0), ("langevin_seed", 7)):
        h.set_option(k, v)
    h.set_velocities(synth.maxwell_boltzmann(spec.mass, 300.0, 4))
    h.calculate_total_force_energy(want_force=False)
    h.md_integrate_atomic(400, spec.delta_t)
    en = h.md_integrate_atomic(2000, spec.delta_t)
    Tk = 2.0 * en[:, 1] / ((3 * spec.n_atom - 3) * 0.008314462)
    assert abs(Tk.mean() - 300.0) < 20.0, Tk.mean()
    h.close()


def _water_box(n_side, seed):
    """the synthetic lattice without its acid molecule (plain water; hydronium = -1)"""
    import dataclasses
    s = synth.make_system(n_side, n_diabat=1, seed=seed)
    a0, a1 = s.mol_first[s.hydronium], s.mol_first[s.hydronium] + s.mol_natom[s.hydronium]
    keep = np.r_[0:a0, a1:s.n_atom]
    natom = np.delete(s.mol_natom, s.hydronium)
    return dataclasses.replace(s, xyz=s.xyz[keep], atom_names=[s.atom_names[k] for k in keep], mol_names=[m for i, m in enumerate(s.mol_names) if i != s.hydronium],
                               mol_natom=natom, mol_first=np.concatenate([[0], np.cumsum(natom)[:-1]]).astype(np.int32), hydronium=-1)


@pytest.mark.parametrize("evb", [False, True])
def test_box_change_equals_a_fresh_context(evb):
    """msevb_set_box = periodic_box_change (src/pme.f90:634): CB array, reciprocal lattice, cells, list and the captured PME graph follow
    the new box - the force call after it equals the one of a context created with that box and those coordinates"""
    from reactive_pb_nn_md_amd import barostat
    import dataclasses
    g = golden("evb_n6_d4_s2")
    spec = system.from_golden(g)
    h = api.MsEvbHip(spec)
    call = (lambda hh: hh.ms_evb_calculate_total_force_energy()) if evb else (lambda hh: (lambda r: (r[0]["potential"], r[1]))(hh.calculate_total_force_energy()))
    e0, F0 = call(h)
    h.set_velocities(synth.maxwell_boltzmann(spec.mass, 300.0, 1))
    h.md_integrate_atomic(3, spec.delta_t, use_evb=evb)      # a replayed PME graph exists now
    scale = 1.03
    t = h.topology()
    mol_atoms = [list(t["mol_atom_index"][m, :t["mol_n_atom"][m]] - 1) for m in range(spec.n_mole)]
    x_new = barostat.MCBarostat.scaled_positions(h.get_positions(), spec.mass, mol_atoms, scale)
    h.set_box(spec.box * scale)
    h.set_positions(x_new)
    e1, F1 = call(h)
    h2 = api.MsEvbHip(dataclasses.replace(spec, box=spec.box * scale, xyz=x_new))
    e2, F2 = call(h2)
    assert rel(e1, e2) < 1e-13 and np.abs(F1 - F2).max() < 1e-10 and rel(e1, e0) > 1e-6
    # ... and back
    h.set_box(spec.box)
    h.set_positions(spec.xyz)
    e3, F3 = call(h)
    assert rel(e3, e0) < 1e-13 and np.abs(F3 - F0).max() < 1e-10
    with pytest.raises(RuntimeError, match="verlet_cutoff < L/2"):
        h.set_box(np.full(3, 15.0))
    h.close(); h2.close()


def test_npt_run_through_the_command_line(tmp_path):
    """ensemble NPT in the simulation-parameter file: Langevin steps + a Monte-Carlo volume move every barofreq steps (mc_sample,
    src/md_integration.f90:17-60).  Water box at a pressure that squeezes it: the box shrinks, the run stays finite"""
    import ref_inputs
    from reactive_pb_nn_md_amd import io, run
    ff = forcefield.ch3so3h_water_forcefield()
    s = _water_box(6, 3)
    d = str(tmp_path)
    ref_inputs.write_gro(d + "/sys.gro", s)
    ref_inputs.write_ffpmt(d + "/ff.pmt", ff)
    ref_inputs.write_top(d + "/sys.top", ff)
    ref_inputs.write_simpmt(d + "/sim.pmt", n_step=60, rc=7.0, rv=8.0, nslist=16, pme_grid=16)
    txt = open(d + "/sim.pmt").read().replace("ensemble NVE", "ensemble NPT").replace("n_output 1000000", "n_output 20")
    open(d + "/sim.pmt", "w").write(txt + "pressure 20000.0\nbarofreq 5\nbaroscale 0.004\nfriction_coeff 50.0\n")
    assert run.main([d + "/sys.gro", d + "/ff.pmt", d + "/sys.top", d + "/sim.pmt", d + "/traj.gro", d + "/run.log", "--evb", "no", "--seed", "2"]) == 0
    first, last = io.read_gro(d + "/traj.gro", 0), io.read_gro(d + "/traj.gro", -1)
    assert last.title.split()[1] == "60" and last.box[0] < first.box[0] and last.box[0] > 0.9 * first.box[0]
    log = [ln.split() for ln in open(d + "/run.log") if len(ln.split()) == 4 and ln.split()[0].isdigit() and "E" in ln.split()[2]]
    assert [int(r[0]) for r in log] == [0, 20, 40, 60] and all(np.isfinite(float(r[2])) for r in log)


def test_command_line_run_and_restart(tmp_path):
    """python -m reactive_pb_nn_md_amd.run on the reference's four input files: trajectory / log / checkpoint in the reference's
    formats, the energies of the library call by call, a refused overwrite, and a restart that continues the step numbering"""
    import ref_inputs
    from reactive_pb_nn_md_amd import io, run
    ff = forcefield.ch3so3h_water_forcefield()
    s = synth.make_system(6, n_diabat=2, seed=5)
    d = str(tmp_path)
    ref_inputs.write_gro(d + "/sys.gro", s)
    ref_inputs.write_ffpmt(d + "/ff.pmt", ff)
    ref_inputs.write_top(d + "/sys.top", ff)
    ref_inputs.write_simpmt(d + "/sim.pmt", n_step=8, rc=7.0, rv=8.0, nslist=16, pme_grid=16)
    txt = open(d + "/sim.pmt").read().replace("n_output 1000000", "n_output 4")
    open(d + "/sim.pmt", "w").write(txt)
    args = [d + "/sys.gro", d + "/ff.pmt", d + "/sys.top", d + "/sim.pmt", d + "/traj.gro", d + "/run.log", d + "/hop.out", d + "/a.out",
            "--seed", "11", "--velocity-checkpoint", d + "/vel", "--checkpoint-every", "4"]
    assert run.main(args) == 0
    log = [ln.split() for ln in open(d + "/run.log") if len(ln.split()) == 4 and ln.split()[0].isdigit() and "E" in ln.split()[2]]
    assert [int(r[0]) for r in log] == [0, 4, 8]
    # the same run through the API
    spec = io.load_inputs(d + "/sys.gro", d + "/ff.pmt", d + "/sys.top", d + "/sim.pmt")[0]
    h = api.MsEvbHip(spec)
    e0, _ = h.ms_evb_calculate_total_force_energy(want_force=False)
    h.set_velocities(synth.maxwell_boltzmann(spec.mass, 300.0, 11))
    en = h.md_integrate_atomic(8, spec.delta_t, use_evb=True)
    assert rel(float(log[0][2]), e0) < 1e-5 and rel(float(log[1][2]), en[3, 0]) < 1e-5 and rel(float(log[2][2]), en[7, 0]) < 1e-5      # E16.6 as the reference prints it: six significant digits
    assert rel(float(log[2][3]), en[7, 1]) < 1e-6
    # the diabat table of print_evb_trajectory_data: one block per printed step, coefficients of the last one = c_d^2 of the API run
    txt_log = open(d + "/run.log").read()
    assert txt_log.count("number of diabat states") == 3
    tab = [ln.split() for ln in txt_log.split("number of diabat states")[-1].split("\n")[2:] if len(ln.split()) == 4]
    st = h.evb_state()
    assert len(tab) == st["D"] and all(abs(float(r[2]) - st["eigvec"][k] ** 2) < 2e-6 for k, r in enumerate(tab))
    assert int(tab[0][1]) == h.topology()["hydronium_molecule"] and int(tab[0][3]) == 0 and int(tab[1][3]) >= 1
    last = io.read_gro(d + "/traj.gro", -1)
    assert last.title.split()[1] == "8" and np.abs(last.xyz - h.get_positions()).max() < 6e-3      # F8.3 nm
    st, v = io.read_velocity_checkpoint(d + "/vel", spec.n_atom)
    assert st == 8 and np.abs(v - h.get_velocities()).max() < 1e-6
    h.close()
    # outputs exist and the checkpoint too -> restart: four more steps appended, numbered 12
    open(d + "/sim.pmt", "w").write(txt.replace("n_step 8", "n_step 12"))
    assert run.main(args) == 0
    log = [ln.split() for ln in open(d + "/run.log") if len(ln.split()) == 4 and ln.split()[0].isdigit() and "E" in ln.split()[2]]
    assert [int(r[0]) for r in log] == [0, 4, 8, 12]
    assert io.read_gro(d + "/traj.gro", -1).title.split()[1] == "12"
    # without the checkpoint the existing outputs are not overwritten
    os.remove(d + "/vel")
    with pytest.raises(SystemExit, match="output file exists"):
        run.main(args)


@pytest.mark.parametrize("evb", [False, True])
def test_rolling_pruned_pair_list_is_exact(evb):
    """the pair loop runs on rows pruned to rc + delta and rewrites them whenever the two largest displacements since the last pruning
    add up to delta: the set of pairs inside rc is always complete, so trajectories with and without pruning agree to rounding - also
    with a tiny delta (re-pruned almost every step) and a delta beyond the list radius (never pruned)"""
    g = golden("evb_n6_d4_s2")
    spec = system.from_golden(g)
    out = []
    for prune, delta in ((0, 0.6), (1, 0.6), (1, 0.05), (1, 5.0)):
        h = api.MsEvbHip(spec)
        h.set_option("pair_prune", prune)
        h.set_option("prune_delta", delta)
        h.set_velocities(synth.maxwell_boltzmann(spec.mass, 300.0, 21))
        if evb:
            h.ms_evb_calculate_total_force_energy(want_force=False)
        else:
            h.calculate_total_force_energy(want_force=False)
        en = h.md_integrate_atomic(60, spec.delta_t, use_evb=evb)
        out.append((en, h.get_positions(), h.get_forces()))
        h.close()
    for en, x, F in out[1:]:
        assert np.abs(en - out[0][0]).max() < 1e-8 * np.abs(out[0][0]).max()
        assert np.abs(x - out[0][1]).max() < 1e-9 and np.abs(F - out[0][2]).max() < 1e-6


def test_orthorhombic_box_against_oracle(oracle_lib):
    """box lengths 1 : 1.12 : 1.31 (molecules kept rigid, centres of mass scaled): different cell counts, tile shapes and reciprocal
    vectors per axis.  No reference run for this shape (the fixtures are cubic) - GPU against the CPU oracle, EVB force call + MD steps"""
    import dataclasses
    s = synth.make_system(8, n_diabat=4, seed=4)
    scale = np.array([1.0, 1.12, 1.31])
    xyz = s.xyz.copy()
    for f, n in zip(s.mol_first, s.mol_natom):
        com = xyz[f:f + n].mean(0)
        xyz[f:f + n] += com * (scale - 1.0)
    s = dataclasses.replace(s, xyz=xyz, box=s.box * scale)
    p = forcefield.build_params(forcefield.ch3so3h_water_forcefield(), moltype_order=["h2o", "so3h"])
    spec = system.from_synth(s, p, rc=7.0, rv=8.0, pme_grid=24)
    o = oracle_lib.Oracle(spec, threads=8)
    o.verlet_build()
    r = o.evb()
    h = api.MsEvbHip(spec)
    e, F = h.ms_evb_calculate_total_force_energy()
    st = h.evb_state()
    assert st["D"] == r["D"] and r["D"] >= 2
    assert np.abs(st["H"] - r["H"]).max() < E_RTOL * np.abs(r["H"]).max()
    assert rel(e, r["E"]) < E_RTOL and np.abs(F - r["F"]).max() < 1e-8
    vel = synth.maxwell_boltzmann(spec.mass, 300.0, 6)
    h.set_velocities(vel)
    en = h.md_integrate_atomic(5, spec.delta_t, use_evb=True)
    v, Fo = vel.copy(), r["F"].copy()
    o.apply_principal()
    for k in range(5):
        pe = o.md_step(v, Fo, spec.delta_t, True)
        assert abs(pe - en[k, 0]) < 1e-8 * abs(pe)
    assert np.abs(h.get_positions() - o.get_xyz()).max() < 1e-9
    h.close()


def test_device_side_hamiltonian_solve():
    """evb_device_solve (default 1): H' assembly, Jacobi with the reference's rotations in round-robin order and the blend coefficients in one
    kernel on the main stream, the blend phase behind it without a host turn; 0 = fetch, the reference's row-cyclic Jacobi on the host, launch.
    Hamiltonian, energy and ground-state vector agree to rounding, same trajectory"""
    g = golden("evb_n6_d8_s1")
    spec = system.from_golden(g)
    out = []
    for dev in (0, 1):
        h = api.MsEvbHip(spec)
        h.set_option("evb_device_solve", dev)
        e, F = h.ms_evb_calculate_total_force_energy()
        st = h.evb_state()
        h.set_velocities(synth.maxwell_boltzmann(spec.mass, 300.0, 8))
        en = h.md_integrate_atomic(10, spec.delta_t, use_evb=True)
        out.append((e, F, st["H"], st["eigvec"], en))
        h.close()
    # (two runs of the same call differ in the last bits anyway: the candidate list is compacted with an atomic counter, so the order
    # of the energy partial sums is not fixed)
    dv = min(np.abs(out[0][3] - out[1][3]).max(), np.abs(out[0][3] + out[1][3]).max())      # (the sign of an eigenvector means nothing: the blend uses products c_i c_j)
    assert rel(out[0][0], out[1][0]) < 1e-14 and np.abs(out[0][2] - out[1][2]).max() < 1e-13 * np.abs(out[0][2]).max() and dv < 1e-13
    assert np.abs(out[0][1] - out[1][1]).max() < 1e-10 and np.abs(out[0][4] - out[1][4]).max() < 1e-9 * np.abs(out[0][4]).max()


def test_large_pme_grid_graph_replay_with_batched_diabat_ffts():
    """128^3 grid: rocFFT needs work buffers.  The single-grid front end is replayed from a captured hipGraph (its work buffer address
    is baked in) while the diabat grids use batched plans with a larger work buffer of their own - sharing one buffer freed the
    captured one when the batched plans grew it (SURVEY config C faulted on its second force call)"""
    spec = _synthetic(10, 4, 3, 128, rc=7.0, rv=8.0)
    out = []
    for graphs in (1, 0):
        h = api.MsEvbHip(spec)
        h.set_option("graphs", graphs)
        h.set_velocities(synth.maxwell_boltzmann(spec.mass, 300.0, 2))
        e, F = h.ms_evb_calculate_total_force_energy()
        en = h.md_integrate_atomic(6, spec.delta_t, use_evb=True)
        out.append((e, F, en, h.get_positions()))
        h.close()
    assert rel(out[0][0], out[1][0]) < 1e-13 and np.abs(out[0][1] - out[1][1]).max() < 1e-10
    assert np.abs(out[0][2] - out[1][2]).max() < 1e-9 * np.abs(out[1][2]).max() and np.abs(out[0][3] - out[1][3]).max() < 1e-10


def test_rccl_communicator_of_one_rank():
    """the real collective path (ncclAllGather of the Hamiltonian entries on the side stream, in-place ncclAllReduce of the forces on
    the main stream) on a communicator of ONE rank: same force call, same trajectory as without a communicator"""
    g = golden("evb_n6_d8_s1")
    spec = system.from_golden(g)
    vel = synth.maxwell_boltzmann(spec.mass, 300.0, 5)
    out = []
    for comm in (False, True):
        h = api.MsEvbHip(spec)
        if comm:
            h.comm_init(api.MsEvbHip.comm_unique_id(h.L), 0, 1)
        e, F = h.ms_evb_calculate_total_force_energy()
        st = h.evb_state()
        h.set_velocities(vel)
        en = h.md_integrate_atomic(20, spec.delta_t, use_evb=True)
        out.append((e, F, st["H"], en, h.get_positions()))
        h.close()
    (e0, F0, H0, en0, x0), (e1, F1, H1, en1, x1) = out
    assert rel(e1, e0) < 1e-14 and np.abs(H1 - H0).max() < 1e-12 * np.abs(H0).max()
    assert np.abs(F1 - F0).max() < 1e-10
    assert np.abs(en1 - en0).max() < 1e-9 * np.abs(en0).max() and np.abs(x1 - x0).max() < 1e-10


def test_comm_wait_gives_up_on_a_missing_peer():
    """The host's wait for the per-step all-gather has a wall-clock limit (option comm_timeout_ms, default 30 s): a peer that died or never entered
    the collective must end the force call with MSEVB_ERR_COMM, not hang every rank (round 3's loop spun on hipErrorNotReady for ever).
    One rank, real communicator; the missing peer is played by a kernel that holds the collective's stream for 1.5 s (option comm_stall_ms,
    one-shot) against a limit of 200 ms.  The stream drains by itself, and the context works again afterwards."""
    import time
    g = golden("evb_n6_d4_s2")
    spec = system.from_golden(g)
    h = api.MsEvbHip(spec)
    h.comm_init(api.MsEvbHip.comm_unique_id(h.L), 0, 1)
    e0, F0 = h.ms_evb_calculate_total_force_energy()
    h.set_option("comm_timeout_ms", 200)
    h.set_option("comm_stall_ms", 1500)
    t0 = time.time()
    with pytest.raises(api.MsevbError) as ei:
        h.ms_evb_calculate_total_force_energy()
    dt = time.time() - t0
    assert "peer rank is missing" in str(ei.value) and 0.15 < dt < 1.4, (str(ei.value), dt)
    h.synchronize()
    h.set_option("comm_timeout_ms", 30000)
    e1, F1 = h.ms_evb_calculate_total_force_energy()
    assert rel(e1, e0) < 1e-13 and np.abs(F1 - F0).max() < 1e-10
    h.close()


def _synthetic(n_side, n_diabat, seed, K, rc=9.0, rv=10.5):
    s = synth.make_system(n_side, n_diabat=n_diabat, seed=seed)
    p = forcefield.build_params(forcefield.ch3so3h_water_forcefield(), moltype_order=["h2o", "so3h"])
    return system.from_synth(s, p, rc=rc, rv=rv, pme_grid=K)


def test_config_A_against_oracle(oracle_lib):
    """BASELINE.json configs[1]: ~10k atoms, 2 diabats, 32^3 grid - GPU vs CPU oracle (no reference run at this size)"""
    spec = _synthetic(15, 2, 7, 32)
    o = oracle_lib.Oracle(spec, threads=8)
    o.verlet_build()
    r = o.evb()
    h = api.MsEvbHip(spec)
    e, F = h.ms_evb_calculate_total_force_energy()
    st = h.evb_state()
    assert st["D"] == r["D"] == 2
    assert np.abs(st["H"] - r["H"]).max() < E_RTOL * np.abs(r["H"]).max()
    assert rel(e, r["E"]) < E_RTOL and np.abs(F - r["F"]).max() < 1e-8
    h.close()


def test_full_size_properties_config_B():
    """BASELINE.json configs[2] (~100k atoms, 8 diabats, 64^3): size-independent properties"""
    spec = _synthetic(32, 8, 11, 64)
    h = api.MsEvbHip(spec)
    h.construct_verlet_list()
    vp1, nl1 = h.get_verlet_list()                    # LDS-staged build (workgroup per cell; only used when there are >= 64 cells)
    h.set_option("nbr_tiles", 0)
    h.construct_verlet_list()
    vp0, nl0 = h.get_verlet_list()                    # wave-per-atom build (the one the small fixtures pin against the reference)
    assert np.array_equal(vp1, vp0) and np.array_equal(nl1, nl0)
    h.set_option("nbr_tiles", 1)
    h.construct_verlet_list()
    h.zero_forces()
    h.real_space_energy_force()
    F = h.get_forces()
    assert np.abs(F.sum(0)).max() < 1e-6 * np.abs(F).sum() / len(F) * 1e3      # Newton's third law: sum of pair forces vanishes
    e_fast, F_fast = h.ms_evb_calculate_total_force_energy()      # default: blended-potential path
    h.set_option("evb_element_stores", 1)
    e, F = h.ms_evb_calculate_total_force_energy()
    assert rel(e_fast, e) < 1e-13 and np.abs(F_fast - F).max() < 1e-9
    st = h.evb_state()
    assert st["D"] == 8 and abs(np.linalg.norm(st["eigvec"]) - 1.0) < 1e-12
    w = np.linalg.eigvalsh(np.triu(st["H"]) + np.triu(st["H"], 1).T)
    assert rel(e, w[0]) < 1e-12                                              # Jacobi ground state == LAPACK ground state
    # Hellmann-Feynman blend == explicit combination of the stored element forces
    c = st["eigvec"]
    Fb = sum(c[d] ** 2 * h.evb_element_force(d + 1, d + 1) for d in range(8))
    Fb = Fb + sum(2 * c[st["coupling_matrix"][d] - 1] * c[d] * h.evb_element_force(st["coupling_matrix"][d], d + 1) for d in range(1, 8))
    assert np.abs(Fb - F).max() < 1e-9
    # idempotence: same positions -> same answer
    e2, F2 = h.ms_evb_calculate_total_force_energy()
    assert rel(e2, e) < 1e-13 and np.abs(F2 - F).max() < 1e-9
    # the three-stream schedule is only a schedule: everything in call order on one stream gives the same numbers
    h.set_option("overlap", 0)
    e3, F3 = h.ms_evb_calculate_total_force_energy()
    assert rel(e3, e) < 1e-13 and np.abs(F3 - F).max() < 1e-9
    h.set_option("overlap", 1)
    # ... also along a trajectory (list rebuilds, self-cleaning delta store, asynchronous kinetic energies): 40 steps from the same
    # state with and without overlap give the same energies
    x0 = h.get_positions()
    v0 = synth.maxwell_boltzmann(spec.mass, 300.0, 4321)
    h.set_velocities(v0)
    tr1 = h.md_integrate_atomic(40, spec.delta_t, use_evb=True)       # starts from the forces of the call above (same positions)
    h.set_positions(x0)
    h.set_velocities(v0)
    h.set_option("overlap", 0)
    h.ms_evb_calculate_total_force_energy(want_force=False)           # velocity Verlet starts from the forces at the current positions
    tr2 = h.md_integrate_atomic(40, spec.delta_t, use_evb=True)
    assert np.abs(tr1 - tr2).max() < 1e-6 * np.abs(tr1).max()
    h.close()


@pytest.mark.parametrize("stores", [0, 1])
def test_config_4_workload_eight_virtual_ranks(stores):
    """BASELINE.json configs[3] at its own size: 98 220 atoms, 8 diabats, 64^3, sharded one diabat per rank over EIGHT (virtual) ranks
    with the principal diabat's pair loop / per-molecule terms / reciprocal forces split as msevb_comm_init does: the ranks' partial
    blends add up to the single-rank force, the partial H11 to H11, every Hamiltonian entry is produced by exactly one rank."""
    spec = _synthetic(32, 8, 11, 64)
    h = api.MsEvbHip(spec)
    _e, Fblend = h.ms_evb_calculate_total_force_energy()
    full = h.evb_state()
    assert full["D"] == 8
    h.close()
    R = 8
    Fsum, Esum, Hsum = np.zeros_like(Fblend), 0.0, np.zeros((8, 8))
    for r in range(R):
        hr = api.MsEvbHip(spec)
        hr.set_option("evb_element_stores", stores)
        hr.set_option("shard_principal", 1)
        hr.comm_set_virtual(r, R)
        e_part, _ = hr.ms_evb_calculate_total_force_energy(want_force=False)
        st = hr.evb_state()
        for d in range(2, 9):
            p = full["coupling_matrix"][d - 1]
            if (d - 2) % R == r:
                Hsum[p - 1, d - 1] += st["H"][p - 1, d - 1]
            else:
                assert st["H"][p - 1, d - 1] == 0.0 and st["H"][d - 1, d - 1] == 0.0      # produced by exactly one rank
        Esum += e_part
        Fsum += hr.evb_partial_blend(full["eigvec"])
        hr.close()
    assert rel(Esum, full["H"][0, 0]) < 1e-13
    for d in range(2, 9):
        p = full["coupling_matrix"][d - 1]
        assert rel(Hsum[p - 1, d - 1], full["H"][p - 1, d - 1]) < 1e-11, d
    assert np.abs(Fsum - Fblend).max() < F_TOL


def test_repulsion_of_the_principal_diabat_against_reference():
    """MS-EVB3 repulsion terms on their own (fixture evb_n6_d4_rep_s25: ms_evb_three_atom_repulsion + ms_evb_born_mayer called stand-alone by the
    reference driver): H_11 - E(principal force field) - reference energy of the hydronium is exactly their sum."""
    g = golden("evb_n6_d4_rep_s25")
    spec = system.from_golden(g)
    h = api.MsEvbHip(spec)
    h.ms_evb_calculate_total_force_energy()
    H00 = h.evb_state()["H"][0, 0]
    en, _ = h.calculate_total_force_energy()
    e_hyd = g["evb_reference_energy"][int(g["mol_type_index"][int(g["hydronium_molecule"][0]) - 1]) - 1]
    rep = H00 - en["potential"] - e_hyd
    assert abs(rep - (g["rep3_E"][0] + g["bm_E"][0])) < 1e-9, (rep, g["rep3_E"][0], g["bm_E"][0])
    h.close()


def test_config_B_evb_against_oracle(oracle_lib):
    """BASELINE.json configs[2] at its own size: the WHOLE MS-EVB force call (98 220 atoms, 8 diabats, 64^3) against the CPU oracle, which evaluates
    every diabat from scratch in its own topology - Hamiltonian, adiabatic energy, Hellmann-Feynman force, diabat enumeration."""
    s = synth.make_system(32, n_diabat=8, seed=0, max_chain=3)
    p = forcefield.build_params(forcefield.ch3so3h_water_forcefield(), moltype_order=["h2o", "so3h"])
    spec = system.from_synth(s, p, rc=9.0, rv=10.5, alpha=0.3, pme_grid=64, max_chain=3)
    h = api.MsEvbHip(spec)
    e, F = h.ms_evb_calculate_total_force_energy()
    st = h.evb_state()
    o = oracle_lib.Oracle(spec, threads=min(32, os.cpu_count() or 8))
    o.verlet_build()
    r = o.evb()
    assert st["D"] == r["D"] == 8
    plog_o = r["proton_log"].copy(); plog_o[plog_o >= 0] += 1      # the oracle counts from 0, the library reports the reference's 1-based log
    assert np.array_equal(st["proton_log"], plog_o) and np.array_equal(st["coupling_matrix"], np.where(r["parent"] >= 0, r["parent"] + 1, -1))
    Hs = np.triu(st["H"]); Ho = np.triu(r["H"])
    assert np.abs(Hs - Ho).max() < 1e-11 * np.abs(Ho).max(), np.abs(Hs - Ho).max()
    assert rel(e, r["E"]) < 1e-11 and np.abs(F - r["F"]).max() < 1e-8, (e, r["E"], np.abs(F - r["F"]).max())
    h.close()


def test_config_C_against_oracle(oracle_lib):
    """BASELINE.json configs[4] on one GPU: 985 362 atoms, 128^3 grid, 16 planted diabats (chains of 4 hops).  Principal-diabat force call
    against the CPU oracle at full size (one call), schedule independence, Jacobi ground state == LAPACK, third law."""
    s = synth.make_system(69, n_diabat=16, seed=0, max_chain=4)
    p = forcefield.build_params(forcefield.ch3so3h_water_forcefield(), moltype_order=["h2o", "so3h"])
    spec = system.from_synth(s, p, rc=9.0, rv=10.5, alpha=0.3, pme_grid=128, max_chain=4)
    assert spec.n_atom > 980000
    h = api.MsEvbHip(spec)
    en, F = h.calculate_total_force_energy()
    o = oracle_lib.Oracle(spec, threads=min(64, os.cpu_count() or 8))
    o.verlet_build()
    e_o, E_o, F_o = o.total()
    assert rel(en["potential"], e_o) < E_RTOL and np.abs(F - F_o).max() < 1e-8
    e1, F1 = h.ms_evb_calculate_total_force_energy()
    st = h.evb_state()
    assert st["D"] == 16 and abs(np.linalg.norm(st["eigvec"]) - 1.0) < 1e-12
    # the WHOLE MS-EVB call against the oracle at this size (16 diabats, each from scratch in its own topology on the CPU: about a minute)
    r = o.evb()
    del o
    plog_o = r["proton_log"].copy(); plog_o[plog_o >= 0] += 1
    assert r["D"] == 16 and np.array_equal(st["proton_log"], plog_o) and np.array_equal(st["coupling_matrix"], np.where(r["parent"] >= 0, r["parent"] + 1, -1))
    assert np.abs(np.triu(st["H"]) - np.triu(r["H"])).max() < 1e-11 * np.abs(r["H"]).max()
    assert rel(e1, r["E"]) < 1e-11 and np.abs(F1 - r["F"]).max() < 1e-8, (e1, r["E"], np.abs(F1 - r["F"]).max())
    del r
    w = np.linalg.eigvalsh(np.triu(st["H"]) + np.triu(st["H"], 1).T)
    assert rel(e1, w[0]) < 1e-12
    assert rel(st["H"][0, 0], en["potential"] + (st["H"][0, 0] - en["potential"])) < 1e-15
    h.set_option("overlap", 0); h.set_option("graphs", 0); h.set_option("pair_prune", 0)
    e0, F0 = h.ms_evb_calculate_total_force_energy()
    assert rel(e0, e1) < 1e-13 and np.abs(F0 - F1).max() < F_TOL
    h.set_option("overlap", 1); h.set_option("graphs", 1); h.set_option("pair_prune", 1)
    h.set_velocities(synth.maxwell_boltzmann(spec.mass, 300.0, 1))
    tr = h.md_integrate_atomic(10, spec.delta_t, use_evb=True)
    assert np.all(np.isfinite(tr)) and abs(tr[-1].sum() - tr[0].sum()) < 2e-3 * abs(tr[0].sum())
    h.close()


def test_verlet_rebuild_flag_and_error_paths():
    g = golden("std_n6_s3")
    spec = system.from_golden(g)
    h = api.MsEvbHip(spec)
    h.construct_verlet_list()
    assert h.update_verlet_displacements() == 0
    x = spec.xyz.copy()
    x[0] += 2.0
    x[1] += 2.0          # two displacements of 3.46 A > 1.2 * skin
    h.set_positions(x)
    assert h.update_verlet_displacements() == 1
    with pytest.raises(api.MsevbError):
        h.ms_evb_calculate_total_force_energy()       # EVB tables were not set for a non-EVB fixture
    h.close()


@pytest.mark.gpu
def test_rccl_two_ranks_when_two_gpus_are_visible():
    """tools/rccl_two_rank_check.py under torchrun with 2 ranks (one GPU each): the sharded MS-EVB call over a real 2-rank RCCL communicator equals the
    single-rank call.  Skipped on a one-GPU box (the driver's multi-GPU node runs it)."""
    import subprocess, json, socket, sys
    import torch
    from conftest import ROOT
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "tools", "rccl_two_rank_check.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["ranks_seen"] == 2 and out["ok"], out


@pytest.mark.gpu
def test_host_atom_order_permutation():
    """msevb_set_host_order: the caller's arrays in ANOTHER atom order than the library's (what the reference's host arrays are after a proton hop,
    src/ms_evb.f90:2677-2836) - positions in, forces / positions / velocities out are the same numbers, permuted; nothing else is re-uploaded."""
    import time
    g = golden("evb_n6_d4_s2")
    spec = system.from_golden(g)
    h = api.MsEvbHip(spec)
    e0, F0 = h.ms_evb_calculate_total_force_energy()
    rng = np.random.default_rng(5)
    p = rng.permutation(spec.n_atom)              # caller's atom i = library atom p[i]
    t0 = time.time(); h.set_host_order(p); dt = time.time() - t0
    h.set_positions(spec.xyz[p])
    e1, F1 = h.ms_evb_calculate_total_force_energy()
    assert abs(e1 - e0) < 1e-10 * abs(e0) and np.abs(F1 - F0[p]).max() < 1e-9
    assert np.array_equal(h.get_positions(), spec.xyz[p])
    v = rng.normal(size=(spec.n_atom, 3)); h.set_velocities(v)
    assert np.array_equal(h.get_velocities(), v)
    h.set_host_order(None)
    assert np.array_equal(h.get_positions(), spec.xyz) and np.array_equal(h.get_velocities()[p], v)
    with pytest.raises(RuntimeError):
        h.set_host_order(np.zeros(spec.n_atom, dtype=int))      # not a permutation
    assert dt < 0.05
    h.close()
