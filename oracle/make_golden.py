#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY (oracle/): generates tests/golden/*.npz by running the REAL reference
(oracle/_ref/ref_dump_{std,evb}: /root/reference/src compiled by oracle/build_ref.sh, FFTs by
Intel MKL) on seeded synthetic inputs.  Only runs in the dev container; the fixtures it writes are
data (inputs + reference outputs) and are committed.

    python oracle/make_golden.py            # regenerate everything
"""
from __future__ import annotations

import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_inputs  # noqa: E402
from reactive_pb_nn_md_amd import forcefield, synth  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
ONLY = set(sys.argv[1:])       # optional: names of the cases to (re)generate

BIG_TABLES = ("B6_spline", "B5_spline", "erfc_table", "ewaldscale_table", "Tang_Toennies_table",
              "dTang_Toennies_table", "CB")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def pairs_hash(verlet_point, neighbor_list):
    """Order-independent digest of the half Verlet list: per atom i the sorted j's."""
    vp = verlet_point.astype(np.int64) - 1
    h = hashlib.sha256()
    for i in range(len(vp) - 1):
        js = np.sort(neighbor_list[vp[i]:vp[i + 1]]).astype("<i4")
        h.update(np.int32(i + 1).tobytes())
        h.update(js.tobytes())
    return h.hexdigest()


def compact(d, keep_store=True):
    out = {}
    for k, v in d.items():
        if k in BIG_TABLES:
            out[k + "_sha256"] = np.array(sha(v))
            flat = v.ravel()
            out[k + "_sample_idx"] = np.arange(0, flat.size, 997, dtype=np.int64)
            out[k + "_sample"] = flat[::997].copy()
            out[k + "_shape"] = np.array(v.shape)
        elif k in ("Q_grid", "theta_conv_Q") and v.size > 100000:      # a 64^3 grid: every 61st point (61 is coprime to 64: all planes, rows and columns are visited)
            out[k + "_sample"] = v.ravel()[::61].copy()
            out[k + "_sample_stride"] = np.array(61)
            out[k + "_sum"] = np.array(v.sum())
        elif k == "neighbor_list":
            out["neighbor_pairs_sha256"] = np.array(pairs_hash(d["verlet_point"], v))
            out["neighbor_count"] = np.array(v.size)
        elif k == "Q_grid_diabats":
            D = v.shape[0]
            base = v[0]
            out["Qd_base_is_Q_grid"] = np.array(np.array_equal(base, d["Q_grid"]))
            for i in range(1, D):
                diff = (v[i] - base).ravel()
                nz = np.nonzero(v[i].ravel() != base.ravel())[0]
                out["Qd%d_idx" % (i + 1)] = nz.astype(np.int64)
                out["Qd%d_val" % (i + 1)] = v[i].ravel()[nz]
        elif k == "evb_forces_store" and not keep_store:
            continue
        elif k in ("evb_hamiltonian", "evb_forces_lookup_index") and "n_diabat" in d:
            D = int(d["n_diabat"][0])
            out[k] = v.T[:D, :D].copy()          # -> [i,j] = reference (i,j)
        elif k == "evb_diabat_proton_log" and "n_diabat" in d:
            D = int(d["n_diabat"][0])
            out[k] = v.transpose(2, 1, 0)[:D].copy()   # [diabat, hop, 5]
        elif k == "evb_diabat_coupling_matrix" and "n_diabat" in d:
            out[k] = v[:int(d["n_diabat"][0])].copy()
        else:
            out[k] = v
    return out


def case(name, n_side, n_diabat, seed, evb, keep_store=True, md_steps=0, sapt=False, proton_shift=0.0, so3h_eref=None, standard=False, rb=False, cf2=False, freeze=(), md_temp=300.0, nbaro=0, box_scale=None, bifurcate=(), **simkw):
    if ONLY and name not in ONLY:
        return
    ff = forcefield.ch3so3h_water_forcefield()
    if sapt:
        ff = forcefield.with_test_sapt(ff)
    if rb:
        ff = forcefield.with_rb_dihedral(ff)
    if cf2:
        ff = forcefield.with_gaussian_coupling(ff)
    if freeze:
        import dataclasses
        ff = dataclasses.replace(ff, freeze=list(freeze))
    if standard:      # Lorentz-Berthelot combination rule (lj_comb_rule standard): cross terms are (epsilon, sigma)
        ff = forcefield.with_standard_cross_terms(ff)
        simkw["lj_comb_rule"] = "standard"
    if so3h_eref is not None:
        import dataclasses
        ff = dataclasses.replace(ff, reference_energy=dict(ff.reference_energy, so3h=so3h_eref))
    s = synth.make_system(n_side, n_diabat=n_diabat, seed=seed, proton_shift=proton_shift, bifurcate=bifurcate)
    if box_scale is not None:      # orthorhombic box: the same atoms, the cell stretched (vacuum at the faces)
        s.box = s.box * np.asarray(box_scale, dtype=float)
    vel = None
    if md_steps:
        p = forcefield.build_params(ff, moltype_order=["h2o", "so3h"])
        idx = {n: i for i, n in enumerate(p.atype_names)}
        mass = np.array([p.atype_mass[idx[a]] for a in s.atom_names])
        vel = synth.maxwell_boltzmann(mass, md_temp, seed + 100)
    d = ref_inputs.run_reference(s, ff, evb=evb, velocities=vel, nsteps=md_steps, nbaro=nbaro, **simkw)
    out = compact(d, keep_store=keep_store)
    out["gen_n_side"] = np.array(n_side)
    out["gen_n_diabat"] = np.array(n_diabat)
    out["gen_seed"] = np.array(seed)
    out["gen_proton_shift"] = np.array(proton_shift)
    out["gen_so3h_eref"] = np.array(np.nan if so3h_eref is None else so3h_eref)
    out["gen_xyz"] = s.xyz
    out["gen_hydronium"] = np.array(s.hydronium)
    out["gen_bifurcate"] = np.asarray(bifurcate, dtype=np.int64)
    out["gen_box_scale"] = np.asarray([1.0, 1.0, 1.0] if box_scale is None else box_scale, dtype=float)
    if vel is not None:
        out["md_velocity0"] = vel
    os.makedirs(GOLD, exist_ok=True)
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-22s N=%5d  %s  %.0f kB" % (name, s.n_atom, "D=%d" % int(d["n_diabat"][0]) if evb else "std", os.path.getsize(path) / 1e3))


def main():
    small = dict(rc=7.0, rv=8.0, nslist=16, pme_grid=16)
    case("std_n8_s0", 8, 1, 0, evb=False)
    case("std_n6_s3", 6, 1, 3, evb=False, **small)
    case("evb_n6_d1_s1", 6, 1, 1, evb=True, **small)
    case("evb_n6_d2_s1", 6, 2, 1, evb=True, **small)
    case("evb_n6_d4_s2", 6, 4, 2, evb=True, **small)
    case("evb_n6_d8_s1", 6, 8, 1, evb=True, **small)
    case("evb_n8_d8_s1", 8, 8, 1, evb=True, keep_store=False)
    case("evb_n6_d4_cf2_s11", 6, 4, 11, evb=True, cf2=True, **small)
    case("std_n6_sapt_s6", 6, 1, 6, evb=False, sapt=True, **small)
    case("evb_n6_d4_sapt_s7", 6, 4, 7, evb=True, sapt=True, **small)
    case("std_n6_lb_s9", 6, 1, 9, evb=False, standard=True, **small)
    case("std_n6_rb_s10", 6, 1, 10, evb=False, rb=True, **small)
    case("std_n6_k20_s13", 6, 1, 13, evb=False, rc=7.0, rv=8.0, nslist=16, pme_grid=20, alpha=0.33)      # grid size that is no power of two, another Ewald parameter
    # 64^3 grid (the size of BASELINE.json configs[2], where the library runs its own LDS transform instead of rocFFT): grids kept as samples
    case("std_n6_k64_s17", 6, 1, 17, evb=False, rc=7.0, rv=8.0, nslist=16, pme_grid=64)
    case("evb_n6_d4_k64_s18", 6, 4, 18, evb=True, keep_store=False, rc=7.0, rv=8.0, nslist=16, pme_grid=64)
    case("std_n6_ex3_s12", 6, 1, 12, evb=False, n_exclusions=3, **small)      # 1-4 pairs excluded instead of taking the pairtypes parameters
    case("md_std_n6_s4", 6, 1, 4, evb=False, md_steps=4, **small)
    case("md_evb_n6_d2_s5", 6, 2, 5, evb=True, md_steps=4, keep_store=False, **small)
    case("md_std_n6_frz_s14", 6, 1, 14, evb=False, md_steps=4, freeze=("OW", "S_a"), **small)      # frozen atom types (atype_freeze)
    # trajectories that cross Verlet-list rebuilds (update_verlet_displacements, src/general_routines.f90:1259-1337: the flag raised in
    # one force call makes the NEXT one rebuild, src/total_energy_forces.f90:30-41) and changes of the diabat count: 60 steps at 600 K.
    # md_verlet_flag / md_n_diabat hold flag_verlet_list and diabat_index after every step.
    case("md_std_n6_rebuild_s21", 6, 1, 21, evb=False, md_steps=60, md_temp=600.0, **small)
    case("md_evb_n6_d4_rebuild_s24", 6, 4, 24, evb=True, md_steps=60, md_temp=600.0, keep_store=False, **small)      # 7 rebuilds, D = 4 -> 5 -> 6
    case("md_evb_n6_d4_drop_s22", 6, 4, 22, evb=True, md_steps=60, md_temp=600.0, keep_store=False, **small)         # 6 rebuilds, D = 4 -> 1
    # Langevin leap-frog kicks (langevin_integrator, src/md_integration.f90:186-228) at temperature 0: the noise term vanishes, what is left
    # (friction + force kick) is deterministic
    case("md_std_n6_nvt0_s16", 6, 1, 16, evb=False, md_steps=4, ensemble="NVT", temperature=0.0, friction_coeff=5.0, **small)
    # Monte-Carlo barostat (monte_carlo_barostat, :231-386): 2 MD steps, then 6 x (move, MD step) at 2 000 bar with a 0.4 % trial step;
    # baro_rand holds the two random numbers each move drew (trial length, Metropolis test)
    case("npt_std_n6_s15", 6, 1, 15, evb=False, md_steps=2, nbaro=6, pressure=2000.0, baroscale=0.004, **small)
    # ORTHORHOMBIC (non-cubic) boxes: the reference's CB array takes V = box(1,1)^3 whatever the other edges are (src/pme.f90:646), its reciprocal lattice the
    # true volume (src/general_routines.f90:473) - both reproduced by tables.cpp and pinned here
    case("std_n6_ortho_s19", 6, 1, 19, evb=False, box_scale=(1.0, 1.07, 1.15), **small)
    case("evb_n6_d4_ortho_s20", 6, 4, 20, evb=True, keep_store=False, box_scale=(1.0, 1.07, 1.15), **small)
    # unit vectors of the MS-EVB3 repulsion procedures (rep3_*, bm_*, rep_switch_table: ref_driver.f90 calls them on their own)
    case("evb_n6_d4_rep_s25", 6, 4, 25, evb=True, keep_store=False, **small)
    # Boxes with >= 4 neighbour cells per axis (12^3 lattice sites, L = 37.2 A, r_v = 8 A): the library's PRODUCTION kernels (half list on LDS tiles,
    # flat-list pair kernel, tiled list build, tiled charge spreading / force interpolation) run against the reference's own numbers
    big = dict(rc=7.0, rv=8.0, nslist=16)
    case("std_n12_k32_s26", 12, 1, 26, evb=False, pme_grid=32, **big)
    case("evb_n12_d8_k64_s27", 12, 8, 27, evb=True, keep_store=False, pme_grid=64, **big)
    case("md_evb_n12_d4_s29", 12, 4, 29, evb=True, md_steps=4, keep_store=False, pme_grid=32, **big)
    # 16 diabats with the reference's compile-time evb_max_chain = 3 (src/glob_v.f90:65): the acid proton and one third-shell hydrogen
    # have TWO acceptors each (synth.py: bifurcate) - 1 + 2 + 4 + 9 states
    case("evb_n12_d16_bif_s31", 12, 16, 31, evb=True, keep_store=False, pme_grid=32, bifurcate=(0, 5), **big)
    # live Buckingham / Tang-Toennies terms on boxes with >= 4 neighbour cells per axis: the SAPT branch of k_pair_flat (the production pair kernel) against the
    # reference's pairwise_real_space_sapt (src/pair_int_real_space.f90:651-690) itself (round 5; the n6 SAPT fixtures run k_pair)
    case("std_n12_sapt_s32", 12, 1, 32, evb=False, sapt=True, pme_grid=32, **big)
    case("evb_n12_d4_sapt_s33", 12, 4, 33, evb=True, keep_store=False, sapt=True, pme_grid=32, **big)
    # principal-diabat switch (src/ms_evb.f90:218-227).  Acid proton started next to its acceptor and the acid's reference energy
    # raised by 850 kJ/mol (the unrelaxed lattice otherwise keeps the acid diabat ~800 kJ/mol below): the reference's FIRST EVB force
    # call (during its initialisation) hands the proton to the water, so everything this fixture holds - 'xyz', topology, Hamiltonian,
    # the MD steps - is the state AFTER the switch, in the reference's re-ordered atom arrays; gen_xyz is the state before it.
    case("md_evb_n6_d4_switch_s8", 6, 4, 8, evb=True, md_steps=6, keep_store=False, proton_shift=0.55, so3h_eref=-643.65 + 850.0, **small)


if __name__ == "__main__":
    main()
