"""The CPU oracle (oracle/msevb_oracle*.c) against golden vectors produced by the REAL reference
(oracle/make_golden.py).  This is what pins the oracle; the GPU tests then compare against it."""
import hashlib

import os

import numpy as np
import pytest

from conftest import all_golden, golden, grid_diff, pre_switch_spec, reference_atom_order
from reactive_pb_nn_md_amd import system

TABLES = (("B6", "B6_spline"), ("B5", "B5_spline"), ("erfc", "erfc_table"), ("ewaldscale", "ewaldscale_table"),
          ("tt", "Tang_Toennies_table"), ("dtt", "dTang_Toennies_table"), ("CB", "CB"))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("name", ["std_n8_s0", "std_n6_s3"])
def test_tables_bit_exact(oracle_lib, name):
    g = golden(name)
    o = oracle_lib.Oracle(system.from_golden(g))
    for nm, key in TABLES:
        assert sha(o.table(nm)) == str(g[key + "_sha256"]), nm
    assert o.ewald_self() == g["Ewald_self"][0]


@pytest.mark.parametrize("name", ["std_n8_s0", "std_n6_s3", "std_n12_k32_s26"])
def test_verlet_list_identical(oracle_lib, name):
    g = golden(name)
    spec = system.from_golden(g)
    o = oracle_lib.Oracle(spec)
    n = o.verlet_build()
    vp, nl = o.verlet_get()
    assert n == int(g["neighbor_count"])
    assert np.array_equal(vp + 1, g["verlet_point"])
    h = hashlib.sha256()
    for i in range(spec.n_atom):
        h.update(np.int32(i + 1).tobytes())
        h.update(np.sort(nl[vp[i]:vp[i + 1]] + 1).astype("<i4").tobytes())
    assert h.hexdigest() == str(g["neighbor_pairs_sha256"])


@pytest.mark.parametrize("name", ["std_n8_s0", "std_n6_s3", "std_n6_sapt_s6", "std_n6_lb_s9", "std_n6_rb_s10", "std_n6_ex3_s12", "std_n6_k20_s13", "std_n6_k64_s17", "std_n6_ortho_s19", "std_n12_k32_s26", "std_n12_sapt_s32"])
def test_stages(oracle_lib, name):
    g = golden(name)
    o = oracle_lib.Oracle(system.from_golden(g))
    o.verlet_build()
    E, F = o.pair()
    assert np.array_equal(F, g["F_pair"])                       # same operation order -> bit-identical
    assert E[0] == g["E_elec_pair"][0] and E[1] == g["E_vdw_pair"][0]
    E, F = o.intra_nonbonded()
    assert np.array_equal(F, g["F_intra_nb"]) and E[0] == g["E_elec_intra_nb"][0]
    E, F = o.pme()
    Q, th, u, kk = o.grid()
    assert np.array_equal(u, g["xyz_scale"]) and grid_diff(g, "Q_grid", Q) == 0.0      # PME grid indices bit-exact
    assert grid_diff(g, "theta_conv_Q", th) < 1e-14                                  # FFT: own radix-2 vs MKL
    assert np.abs(F - g["F_recip"]).max() < 1e-11 and abs(E[0] - g["E_recip"][0]) < 1e-10
    E, F = o.bonded()
    if "_rb_" in name:      # Ryckaert-Bellemans: cosine**n is a libm/compiler power in the reference build, a product here (1 ulp)
        assert np.abs(F - g["F_bonded"]).max() < 1e-12
    else:
        assert np.array_equal(F, g["F_bonded"])
    assert (E[2], E[3], E[4]) == (g["E_bond"][0], g["E_angle"][0], g["E_dihedral"][0])
    e, E, F = o.total()
    assert np.abs(F - g["F_total"]).max() < 1e-11 and abs(e - g["E_total"][0]) < 1e-9 * abs(e)


@pytest.mark.parametrize("name", all_golden("evb_"))
def test_evb_from_scratch_equals_reference_delta(oracle_lib, name):
    """delta-vs-full KAT: the oracle evaluates every diabat from scratch, the reference by delta updates."""
    g = golden(name)
    o = oracle_lib.Oracle(system.from_golden(g))
    o.verlet_build()
    r = o.evb()
    D = int(g["n_diabat"][0])
    assert r["D"] == D
    mine = r["proton_log"].copy()
    mine[mine >= 0] += 1
    assert np.array_equal(mine[:, :g["evb_diabat_proton_log"].shape[1], :], g["evb_diabat_proton_log"])
    assert np.array_equal(np.where(r["parent"] >= 0, r["parent"] + 1, -1), g["evb_diabat_coupling_matrix"])
    Href = g["evb_hamiltonian"]
    assert np.abs(r["H"] - Href).max() < 1e-12 * np.abs(Href).max()
    assert abs(r["E"] - g["E_adiabatic"][0]) < 1e-12 * abs(r["E"])
    assert np.abs(r["F"] - g["F_adiabatic"]).max() < 1e-9
    assert r["principal"] + 1 == int(g["principle_diabat"][0])
    if "evb_forces_store" in g:
        lk, st = g["evb_forces_lookup_index"], g["evb_forces_store"]
        for i in range(D):
            for j in range(i, D):
                if lk[i, j] > 0:
                    Fm = r["F_diag"][i] if i == j else r["F_off"][j]
                    assert np.abs(Fm - st[lk[i, j] - 1]).max() < 1e-9, (i, j)
    # invariants: symmetric tree Hamiltonian, normalised ground state
    assert abs(np.linalg.norm(r["eigvec"]) - 1.0) < 1e-12


@pytest.mark.parametrize("name", all_golden("evb_"))
def test_evb_delta_update_formulation(oracle_lib, name):
    """The oracle's SECOND formulation of the call - the reference's own delta update (ms_evb_diabat_force_energy, src/ms_evb.f90:1421-1560: donor / acceptor
    molecules out in the principal topology and in in the diabat's, all-atom scans, modify_Q_grid on a copy of the principal grid, one reciprocal-space pass per
    diabat :1962-2090), which is what bench.py times as the CPU baseline - against the real reference's Hamiltonian, adiabatic energy / force and per-element force
    stores, and against the from-scratch formulation."""
    g = golden(name)
    o = oracle_lib.Oracle(system.from_golden(g))
    o.verlet_build()
    r = o.evb(delta=True)
    Href = g["evb_hamiltonian"]
    assert r["D"] == int(g["n_diabat"][0])
    assert np.abs(r["H"] - Href).max() < 1e-12 * np.abs(Href).max()
    assert abs(r["E"] - g["E_adiabatic"][0]) < 1e-12 * abs(r["E"]) and np.abs(r["F"] - g["F_adiabatic"]).max() < 1e-9
    if "evb_forces_store" in g:
        lk, st = g["evb_forces_lookup_index"], g["evb_forces_store"]
        for i in range(r["D"]):
            if lk[i, i] > 0:
                assert np.abs(r["F_diag"][i] - st[lk[i, i] - 1]).max() < 1e-9, i
    f = o.evb()
    assert np.abs(f["H"] - r["H"]).max() < 1e-12 * np.abs(Href).max() and np.abs(f["F"] - r["F"]).max() < 1e-10 and np.abs(f["F_diag"] - r["F_diag"]).max() < 1e-10


@pytest.mark.parametrize("name,evb", [("md_std_n6_s4", False), ("md_evb_n6_d2_s5", True), ("md_std_n6_frz_s14", False),
                                      ("md_std_n6_rebuild_s21", False), ("md_evb_n6_d4_rebuild_s24", True), ("md_evb_n6_d4_drop_s22", True),
                                      ("md_evb_n12_d4_s29", True)])
def test_md_trajectory(oracle_lib, name, evb):
    """the *_rebuild_* / *_drop_* fixtures are 60 reference steps at 600 K that cross 6-7 Verlet-list rebuilds
    (update_verlet_displacements, src/general_routines.f90:1259-1337) and changes of the diabat count (4 -> 5 -> 6, 4 -> 1)"""
    g = golden(name)
    spec = system.from_golden(g)
    o = oracle_lib.Oracle(spec)
    o.verlet_build()
    vel = g["md_velocity0"].copy()
    if evb:
        F = o.evb()["F"].copy()
    else:
        F = o.total()[2].copy()
    assert np.abs(F - g["md_F0"]).max() < 1e-9
    n = int(g["md_nsteps"][0])
    for s in range(n):
        pe = o.md_step(vel, F, spec.delta_t, evb)
        ke = o.kinetic_energy(vel)
        assert abs(pe - g["md_energy"][s, 0]) < 1e-8 * abs(pe)
        assert abs(ke - g["md_energy"][s, 1]) < 1e-8 * abs(ke)
        if "md_verlet_flag" in g:
            assert o.verlet_flag() == g["md_verlet_flag"][s], s
            assert (o.n_diabat() if evb else 1) == g["md_n_diabat"][s], s
    if "rebuild" in name or "drop" in name:
        assert g["md_verlet_flag"].sum() >= 6 and (not evb or len(set(g["md_n_diabat"])) > 1)      # the fixture does cross the events
    assert np.abs(o.get_xyz() - g["md_xyz"]).max() < 1e-10
    assert np.abs(vel - g["md_velocity"]).max() < 1e-9
    assert np.abs(F - g["md_force"]).max() < 1e-7


def test_principal_switch(oracle_lib):
    """src/ms_evb.f90:218-227.  The fixture's reference run switched the principal diabat in its first EVB force call; starting
    from the state before it, the oracle must arrive at the reference's topology (atom types, charges, molecule records, the
    transferred proton made whole with its new molecule) and then follow its trajectory on the new topology."""
    g = golden("md_evb_n6_d4_switch_s8")
    post = system.from_golden(g)
    pre = pre_switch_spec(g)
    assert pre.hydronium != post.hydronium
    o = oracle_lib.Oracle(pre)
    o.verlet_build()
    r = o.evb()
    assert r["new_hydronium"] == post.hydronium and r["principal"] != 0
    o.apply_principal()
    t = o.topology()
    perm = reference_atom_order(t)
    assert sorted(perm) == list(range(pre.n_atom))
    assert np.array_equal(t["mol_n_atom"], post.mol_natom) and np.array_equal(t["mol_type_index"], post.mol_type)
    assert np.array_equal(t["atom_type_index"][perm], post.atype)
    assert np.array_equal(t["charge"][perm], post.charge)
    assert np.abs(o.get_xyz()[perm] - post.xyz).max() < 1e-12
    # the force call on the new topology: the reference's Hamiltonian, adiabatic energy and force
    r = o.evb()
    D = int(g["n_diabat"][0])
    assert r["D"] == D and r["new_hydronium"] == post.hydronium and r["principal"] == 0
    assert np.abs(r["H"] - np.triu(g["evb_hamiltonian"])).max() < 1e-9 * np.abs(g["evb_hamiltonian"]).max()
    assert abs(r["E"] - g["E_adiabatic"][0]) < 1e-10 * abs(r["E"])
    assert np.abs(r["F"][perm] - g["F_adiabatic"]).max() < 1e-8
    # ... and the steps after it (the reference read its velocities in ITS atom order)
    vel = np.zeros_like(g["md_velocity0"])
    vel[perm] = g["md_velocity0"]
    F = r["F"].copy()
    assert np.abs(F[perm] - g["md_F0"]).max() < 1e-8
    for s in range(int(g["md_nsteps"][0])):
        pe = o.md_step(vel, F, post.delta_t, True)
        ke = o.kinetic_energy(vel)
        assert abs(pe - g["md_energy"][s, 0]) < 1e-8 * abs(pe) and abs(ke - g["md_energy"][s, 1]) < 1e-8 * abs(ke)
    assert int(g["md_hydronium_molecule"][0]) - 1 == post.hydronium      # no further switch inside the fixture
    assert np.abs(o.get_xyz()[perm] - g["md_xyz"]).max() < 1e-10
    assert np.abs(vel[perm] - g["md_velocity"]).max() < 1e-9
    assert np.abs(F[perm] - g["md_force"]).max() < 1e-7


def test_dfti_adapter_against_numpy_fft(tmp_path):
    """oracle/dfti_mkl_adapter.c is the calling-convention layer between the reference's Fortran DFTI interface and the real Intel MKL
    DFTI inside libtorch_cpu.so (it contains no arithmetic).  Driven here exactly as flang-compiled reference code drives it - a Fortran
    pointer descriptor whose first word is the handle, every argument by reference - on random 20^3 (the size of fixture std_n6_k20_s13)
    and non-cubic grids: forward and backward transforms, unnormalised, in place, against numpy.fft."""
    import ctypes as C
    import importlib.util
    import subprocess
    spec = importlib.util.find_spec("torch")
    if spec is None:
        pytest.skip("no torch (MKL DFTI provider)")
    tlib = os.path.join(os.path.dirname(spec.origin), "lib")
    so = str(tmp_path / "libdfti_adapter.so")
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "dfti_mkl_adapter.c")
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", src, "-o", so, "-L" + tlib, "-ltorch_cpu", "-lc10", "-Wl,-rpath," + tlib])
    L = C.CDLL(so)
    rng = np.random.default_rng(5)
    for dims in ((20, 20, 20), (16, 16, 16), (12, 20, 8)):
        desc = (C.c_void_p * 8)()                       # stands for the Fortran pointer descriptor (first word = base address)
        lens = (C.c_int * 3)(*dims)                     # Fortran extents n1 (fastest), n2, n3
        assert L.dfti_create_descriptor_highd_(desc, C.byref(C.c_int(36)), C.byref(C.c_int(32)), C.byref(C.c_int(3)), lens) == 0
        assert L.dfti_set_value_intval_(desc, C.byref(C.c_int(26)), C.byref(C.c_int(1))) == 0      # DFTI_NUMBER_OF_USER_THREADS
        assert L.dfti_commit_descriptor_external_(desc) == 0
        a = rng.standard_normal(dims[::-1]) + 1j * rng.standard_normal(dims[::-1])                 # C view [n3][n2][n1] of X(n1,n2,n3)
        x = a.copy()
        assert L.dfti_compute_forward_z_(desc, x.ctypes.data_as(C.c_void_p)) == 0
        ref = np.fft.fftn(a)
        assert np.abs(x - ref).max() < 1e-12 * np.abs(ref).max()
        assert L.dfti_compute_backward_z_(desc, x.ctypes.data_as(C.c_void_p)) == 0
        assert np.abs(x - a * a.size).max() < 1e-12 * a.size                                     # unnormalised both ways
        assert L.dfti_free_descriptor_external_(desc) == 0 and not desc[0]


def test_repulsion_procedures_against_reference(oracle_lib):
    """ms_evb_three_atom_repulsion, ms_evb_born_mayer, ms_evb_repulsive_switch (src/ms_evb.f90:2295, :2405, :2484) called ON THEIR OWN by the reference
    driver on the principal topology (fixture evb_n6_d4_rep_s25): the oracle's restatement of their sum, and of the switching function."""
    g = golden("evb_n6_d4_rep_s25")
    o = oracle_lib.Oracle(system.from_golden(g))
    e, F = o.evb_repulsion_principal()
    e_ref = g["rep3_E"][0] + g["bm_E"][0]
    assert e_ref > 1.0 and abs(e - e_ref) < 1e-12 * abs(e_ref)
    assert np.abs(F - (g["rep3_F"] + g["bm_F"])).max() < 1e-12 * np.abs(g["rep3_F"]).max()
    tab = g["rep_switch_table"].reshape(61, 2)
    for k in range(61):
        sw, dsw = o.rep_switch(2.0 + 0.02 * k, 2.4, 3.0)
        assert sw == tab[k, 0] and dsw == tab[k, 1], (k, sw, dsw, tab[k])
