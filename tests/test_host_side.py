"""Host-side logic on the CPU: force-field table construction, synthetic-system generator, the C-ABI surface."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, golden
from reactive_pb_nn_md_amd import api, forcefield, synth, system


def test_generator_is_deterministic_and_matches_fixture_inputs():
    for name in ("std_n8_s0", "evb_n6_d8_s1", "evb_n6_d4_s2"):
        g = golden(name)
        s = synth.make_system(int(g["gen_n_side"]), n_diabat=int(g["gen_n_diabat"]), seed=int(g["gen_seed"]))
        assert np.array_equal(s.xyz, g["gen_xyz"])
        assert s.hydronium == int(g["gen_hydronium"])
        assert np.abs(s.xyz - g["xyz"]).max() == 0.0      # the reference did not move anything at load time


def test_forcefield_tables_match_reference_init():
    """build_params mirrors read_param/gen_param/... (src/initialize_routines.f90:281-691); compare with the tables the
    reference built from the same force field."""
    g = golden("evb_n6_d8_s1")
    ref = system.from_golden(g).params
    mine = forcefield.build_params(forcefield.ch3so3h_water_forcefield(), moltype_order=["h2o", "so3h"])
    assert np.array_equal(mine.vdw_type, ref.vdw_type)
    assert np.allclose(mine.vdw_param, ref.vdw_param, rtol=1e-14, atol=0)
    assert np.allclose(mine.vdw_param14, ref.vdw_param14, rtol=1e-14, atol=0)
    assert np.array_equal(mine.bond_type, ref.bond_type) and np.array_equal(mine.bond_param, ref.bond_param)
    assert np.array_equal(mine.atype_chg, ref.atype_chg) and np.array_equal(mine.atype_mass, ref.atype_mass)
    assert set(mine.angle_type) == set(ref.angle_type) and set(mine.dihedral_type) == set(ref.dihedral_type)
    for k in mine.angle_type:
        assert mine.angle_type[k][0] == ref.angle_type[k][0] and np.allclose(mine.angle_type[k][1:], ref.angle_type[k][1:], rtol=1e-15)
    for k in mine.dihedral_type:
        assert mine.dihedral_type[k][0] == ref.dihedral_type[k][0] and np.allclose(mine.dihedral_type[k][1], ref.dihedral_type[k][1], rtol=1e-15)
    for a, b in zip(mine.moltypes, ref.moltypes):
        assert a["n_atom"] == b["n_atom"] and np.array_equal(a["atype"], b["atype"])
        for key in ("bonds", "angles", "dihedrals", "excl", "reactive_protons", "reactive_basic"):
            assert np.array_equal(a[key], b[key]), (a["name"], key)
    for key in ("acid", "base", "conjugate_mol", "proton_type", "heavy_type", "conjugate_atype", "reference_energy", "exch_atomic", "exch_proton"):
        assert np.array_equal(mine.evb[key], ref.evb[key]), key
    for key in ("donor_acceptor", "proton_acceptor"):
        assert len(mine.evb[key]) == len(ref.evb[key])
        for a, b in zip(mine.evb[key], ref.evb[key]):
            assert tuple(a[0]) == tuple(b[0]) and np.array_equal(a[1], b[1])
    for a, b in zip(mine.evb["coupling"], ref.evb["coupling"]):
        assert tuple(a[0]) == tuple(b[0]) and a[1] == b[1] and np.array_equal(a[2], b[2])


def test_lorentz_berthelot_tables_match_reference_init():
    """lj_comb_rule standard (src/initialize_routines.f90:446-528, :603-605): epsilon/sigma combined by Lorentz-Berthelot, explicit
    cross terms read as (epsilon, sigma), everything converted to C12/C6 last - against the tables of a reference run with that rule"""
    g = golden("std_n6_lb_s9")
    ref = system.from_golden(g).params
    ff = forcefield.with_standard_cross_terms(forcefield.ch3so3h_water_forcefield())
    mine = forcefield.build_params(ff, moltype_order=["h2o", "so3h"], lj_comb_rule="standard")
    opls = forcefield.build_params(forcefield.ch3so3h_water_forcefield(), moltype_order=["h2o", "so3h"])
    assert np.array_equal(mine.vdw_type, ref.vdw_type)
    assert np.allclose(mine.vdw_param, ref.vdw_param, rtol=1e-14, atol=0) and np.allclose(mine.vdw_param14, ref.vdw_param14, rtol=1e-14, atol=0)
    assert not np.allclose(mine.vdw_param, opls.vdw_param, rtol=1e-6)      # a different rule, different cross terms
    with pytest.raises(ValueError, match="isn't recognized"):
        forcefield.build_params(ff, moltype_order=["h2o", "so3h"], lj_comb_rule="geometric")
    with pytest.raises(ValueError, match="should be opls"):      # C6/C12-sized cross terms under the standard rule (:399-404)
        forcefield.build_params(forcefield.ch3so3h_water_forcefield(), moltype_order=["h2o", "so3h"], lj_comb_rule="standard")


def test_exclusion_matrices_for_n_exclusions_3():
    """n_exclusions = 3: atoms three bonds apart are excluded outright instead of interacting with the 1-4 parameters
    (src/intra_bonded_interactions.f90:574-660) - against the matrices of a reference run with that setting"""
    g = golden("std_n6_ex3_s12")
    ref = system.from_golden(g).params
    mine3 = forcefield.build_params(forcefield.ch3so3h_water_forcefield(), moltype_order=["h2o", "so3h"], n_exclusions=3)
    mine2 = forcefield.build_params(forcefield.ch3so3h_water_forcefield(), moltype_order=["h2o", "so3h"], n_exclusions=2)
    differs = False
    for a, b, c2 in zip(mine3.moltypes, ref.moltypes, mine2.moltypes):
        assert np.array_equal(a["excl"], b["excl"]), a["name"]
        differs |= not np.array_equal(a["excl"], c2["excl"])
    assert differs


def test_spec_from_synth_equals_spec_from_reference_dump():
    g = golden("evb_n6_d2_s1")
    ref = system.from_golden(g)
    s = synth.make_system(int(g["gen_n_side"]), n_diabat=int(g["gen_n_diabat"]), seed=int(g["gen_seed"]))
    p = forcefield.build_params(forcefield.ch3so3h_water_forcefield(), moltype_order=["h2o", "so3h"])
    mine = system.from_synth(s, p, rc=ref.rc, rv=ref.rv, pme_grid=ref.pme_grid, nslist=ref.nslist)
    for key in ("atype", "charge", "mass", "mol_first", "mol_natom", "mol_type", "xyz", "box"):
        assert np.array_equal(getattr(mine, key), getattr(ref, key)), key
    assert mine.hydronium == ref.hydronium


def test_fortran_bindings_cover_the_header():
    """reactive_pb_nn_md_amd/fortran/msevb_hip_c.f90 binds every entry point include/msevb_hip.h declares, and nothing else"""
    import re
    src = open(os.path.join(os.path.dirname(api.LIB_PATH), "fortran", "msevb_hip_c.f90")).read()
    bound = sorted(set(re.findall(r'bind\(C,\s*name="(msevb_[A-Za-z_0-9]+)"\)', src)))
    assert bound == api.exported_symbols()


def test_library_exports_every_declared_symbol():
    assert os.path.exists(api.LIB_PATH), "libmsevb_hip.so must be built (python -c 'import __graft_entry__ as g; g.build()')"
    lib = ctypes.CDLL(api.LIB_PATH)
    names = api.exported_symbols()
    assert len(names) > 30
    for n in names:
        assert hasattr(lib, n), n


def test_no_cpu_fallback_and_errors_are_reported():
    """without a GPU the create call must fail loudly (no silent CPU path); with one, bad input must be rejected"""
    lib = api.load_library()
    ctx = ctypes.c_void_p()
    rc = lib.msevb_create(ctypes.byref(ctx), 0)
    if rc != 0:
        assert b"no HIP device" in lib.msevb_last_error(ctx) or b"hip" in lib.msevb_last_error(ctx).lower()
    else:
        box = np.asfortranarray(np.diag([20.0, 20.0, 20.0]))
        assert lib.msevb_set_system(ctx, 10, 3, box.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), ctypes.c_double(9.0), ctypes.c_double(8.0),
                                    ctypes.c_double(0.3), 16, 6, 100000, 100000) != 0       # verlet_cutoff <= real_space_cutoff
        assert b"verlet_cutoff" in lib.msevb_last_error(ctx)
        assert lib.msevb_finalize(ctx) != 0
    lib.msevb_destroy(ctx)


def test_header_cites_reference_for_every_hot_path_entry():
    hdr = open(os.path.join(ROOT, "include", "msevb_hip.h")).read()
    for fn in ("msevb_calculate_total_force_energy", "msevb_real_space_energy_force", "msevb_pme_reciprocal_space_energy_force",
               "msevb_intra_molecular_energy_force", "msevb_construct_verlet_list", "msevb_ms_evb_calculate_total_force_energy",
               "msevb_md_integrate_atomic"):
        i = hdr.index("int " + fn + "(")
        assert re.search(r"src/[a-z_]+\.f90:\d+", hdr[max(0, i - 900):i]), fn


# ---- the reference's file formats (reactive_pb_nn_md_amd/io.py).  The files are written by oracle/ref_inputs.py - the writers whose
# ---- output the REAL reference parsed to make every fixture in tests/golden - and read back by the package's own parsers.
def _write_inputs(tmp, s, ff, **simkw):
    import ref_inputs
    p = {k: os.path.join(tmp, k) for k in ("sys.gro", "ff.pmt", "sys.top", "sim.pmt")}
    ref_inputs.write_gro(p["sys.gro"], s)
    ref_inputs.write_ffpmt(p["ff.pmt"], ff)
    ref_inputs.write_top(p["sys.top"], ff)
    ref_inputs.write_simpmt(p["sim.pmt"], **simkw)
    return p


@pytest.mark.parametrize("sapt", [False, True])
def test_input_files_parse_to_the_same_system(tmp_path, sapt):
    import dataclasses
    from reactive_pb_nn_md_amd import io
    ff = forcefield.ch3so3h_water_forcefield()
    if sapt:
        ff = forcefield.with_test_sapt(ff)
    s = synth.make_system(6, n_diabat=4, seed=2)
    p = _write_inputs(str(tmp_path), s, ff, rc=7.0, rv=8.0, nslist=16, pme_grid=16)
    spec, gro, ff2, strings, numbers = io.load_inputs(p["sys.gro"], p["ff.pmt"], p["sys.top"], p["sim.pmt"])
    for f in dataclasses.fields(ff):
        assert getattr(ff, f.name) == getattr(ff2, f.name), f.name
    assert gro.hydronium == s.hydronium and np.array_equal(gro.xyz, s.xyz) and gro.atom_names == list(s.atom_names)
    ref = system.from_synth(s, forcefield.build_params(ff, moltype_order=["h2o", "so3h"]), rc=7.0, rv=8.0, pme_grid=16, nslist=16)
    for f in dataclasses.fields(ref):
        a, b = getattr(ref, f.name), getattr(spec, f.name)
        if isinstance(a, np.ndarray):
            assert np.array_equal(a, b), f.name
        elif f.name not in ("params", "extra"):
            assert a == b, f.name
    assert strings["ensemble"] == "NVE" and numbers["verlet_cutoff"] == 8.0 and numbers["n_exclusions"] == 2
    # ... and the spec equals the one the reference itself built from these files (fixture of the same generator arguments)
    g = golden("evb_n6_d4_s2")
    if not sapt:
        gs = system.from_golden(g)
        assert np.array_equal(gs.atype, spec.atype) and np.array_equal(gs.charge, spec.charge) and np.array_equal(gs.mass, spec.mass)
        assert gs.hydronium == spec.hydronium and np.array_equal(gs.mol_type, spec.mol_type)


def test_parser_rules_of_the_reference(tmp_path):
    from reactive_pb_nn_md_amd import io
    ff = forcefield.ch3so3h_water_forcefield()
    s = synth.make_system(6, n_diabat=1, seed=1)
    p = _write_inputs(str(tmp_path), s, ff, rc=7.0, rv=8.0, nslist=16, pme_grid=16)
    # a six-token atom-type line (the reference's own stale example file) is refused: exactly five (src/initialize_routines.f90:334)
    txt = open(p["ff.pmt"]).read().split("\n")
    txt[3] = txt[3] + " 0"
    open(p["ff.pmt"], "w").write("\n".join(txt))
    with pytest.raises(ValueError, match="exactly 5 tokens"):
        io.read_ffpmt(p["ff.pmt"])
    # comment lines (first token contains ';') are skipped, a blank line ends a section (src/general_routines.f90:329-360)
    top = open(p["sys.top"]).read().replace("[ angletypes ]\n", "[ angletypes ]\n; a comment line\n")
    open(p["sys.top"], "w").write(top)
    assert io.read_top(p["sys.top"])["angletypes"] == ff.angletypes
    # r_c < r_v < L/2 (src/general_routines.f90:1115-1130)
    p2 = _write_inputs(str(tmp_path), s, ff, rc=7.0, rv=30.0, nslist=16, pme_grid=16)
    with pytest.raises(ValueError, match="real_space_cutoff < verlet_cutoff < L/2"):
        io.load_inputs(p2["sys.gro"], p2["ff.pmt"], p2["sys.top"], p2["sim.pmt"])
    # two acid molecules: "can't have more than 1 hydronium" (src/ms_evb.f90:96-120)
    g = io.read_gro(p["sys.gro"])
    g.mol_names[0] = "so3h"
    with pytest.raises(ValueError, match="more than 1 hydronium"):
        io.locate_hydronium(g, ff)


def test_trajectory_frame_and_velocity_checkpoint_round_trip(tmp_path):
    from reactive_pb_nn_md_amd import io
    s = synth.make_system(6, n_diabat=2, seed=3, quantize_gro=True)
    mol_atoms = [list(range(f, f + n)) for f, n in zip(s.mol_first, s.mol_natom)]
    # a proton moved from the acid to the next molecule: membership, names and numbering follow (print_gro_file prints by molecule)
    mol_atoms[s.hydronium + 1].append(mol_atoms[s.hydronium].pop())
    names = list(s.mol_names)
    names[s.hydronium], names[s.hydronium + 1] = "so3", "h3o"
    path = str(tmp_path / "traj.gro")
    with open(path, "w") as f:
        io.write_gro_frame(f, "step  0 time(ps) 0.0", s.mol_names, [list(range(a, a + n)) for a, n in zip(s.mol_first, s.mol_natom)], s.atom_names, s.xyz, s.box)
        io.write_gro_frame(f, "step  10 time(ps) 0.005", names, mol_atoms, s.atom_names, s.xyz, s.box)
    g0, g1 = io.read_gro(path, 0), io.read_gro(path, -1)
    assert np.abs(g0.xyz - s.xyz).max() < 1e-9 and np.array_equal(g0.mol_natom, s.mol_natom) and np.allclose(g0.box, s.box, atol=1e-3)
    assert g1.title.split()[1] == "10" and g1.mol_natom[s.hydronium] == 5 and g1.mol_natom[s.hydronium + 1] == 4
    assert g1.mol_names[s.hydronium + 1] == "h3o" and g1.atom_names[sum(g1.mol_natom[:s.hydronium + 2]) - 1] == "H_a"
    v = np.random.default_rng(0).normal(size=(s.n_atom, 3)) * 30.0
    vp = str(tmp_path / "vel")
    with open(vp, "w") as f:
        io.write_velocity_checkpoint(f, 5, s.mol_names, [list(range(a, a + n)) for a, n in zip(s.mol_first, s.mol_natom)], s.atom_names, v)
        io.write_velocity_checkpoint(f, 10, s.mol_names, [list(range(a, a + n)) for a, n in zip(s.mol_first, s.mol_natom)], s.atom_names, 2 * v)
    st, v2 = io.read_velocity_checkpoint(vp, s.n_atom)
    assert st == 10 and np.abs(v2 - 2 * v).max() < 1e-6
    st, v1 = io.read_velocity_checkpoint(vp, s.n_atom, step=5)
    assert np.abs(v1 - v).max() < 1e-6


@pytest.mark.parametrize("name", ["md_std_n6_s4", "md_evb_n6_d2_s5"])
def test_output_writers_against_files_the_reference_wrote(name):
    """io.py's writers against the trajectory frame, log block and velocity checkpoint the REAL reference wrote for the end state of a
    fixture's MD steps (print_gro_file / print_step / print_velocities_checkpoint, src/general_routines.f90:906, :870, :997; kept by
    oracle/ref_driver.f90).  Formatted records must agree byte for byte; list-directed records (the reference writes the frame title, the
    atom count and the captions with `write(*,*)`, whose spacing is the compiler's) token by token."""
    import io as _io
    from conftest import golden
    from reactive_pb_nn_md_amd import io
    g = golden(name)
    s = synth.make_system(int(g["gen_n_side"]), n_diabat=int(g["gen_n_diabat"]), seed=int(g["gen_seed"]))
    mol_atoms = [list(range(f, f + n)) for f, n in zip(s.mol_first, s.mol_natom)]
    n = int(g["md_nsteps"][0])
    dt = float(g["delta_t"][0])

    def compare(mine, ref, formatted_from=0):
        a, b = mine.splitlines(), ref.splitlines()
        assert len(a) == len(b)
        for k, (x, y) in enumerate(zip(a, b)):
            if k < formatted_from:
                tx, ty = x.split(), y.split()
                assert len(tx) == len(ty) and all(p == q or float(p) == float(q) for p, q in zip(tx, ty)), (k, x, y)
            else:
                assert x == y, (k, x, y)

    f = _io.StringIO()
    io.write_gro_frame(f, "step  %d time(ps) %r" % (n, n * dt), s.mol_names, mol_atoms, s.atom_names, g["md_xyz"], s.box)
    compare(f.getvalue(), bytes(g["file_traj"]).decode(), formatted_from=2)
    f = _io.StringIO()
    io.write_velocity_checkpoint(f, n, s.mol_names, mol_atoms, s.atom_names, g["md_velocity"])
    compare(f.getvalue(), bytes(g["file_vel"]).decode(), formatted_from=1)
    f = _io.StringIO()
    evb = name.startswith("md_evb")
    io.write_log_step(f, n, dt, g["md_energy"][-1, 0], g["md_energy"][-1, 1], None if evb else g["md_E_terms"])
    mine, ref = f.getvalue().splitlines(), bytes(g["file_log"]).decode().splitlines()
    ref = ref[-len(mine):]      # the log also holds what initialisation printed
    for x, y in zip(mine, ref):
        assert x == y if x[:1] == " " and x.strip()[:1] in "0123456789-" or x.strip().startswith("-") and "E" in x else x.split() == y.split(), (x, y)


def test_monte_carlo_barostat_logic():
    """monte_carlo_barostat (src/md_integration.f90:229-386) on a stub handle with a known E(L): acceptance weight, rigid scaling of
    the molecules' centres of mass, forced acceptance when the excess proton moved, step-size adaptation"""
    from reactive_pb_nn_md_amd import barostat

    class Stub:
        def __init__(self, xyz, L, k):
            self.spec = type("S", (), {})()
            self.spec.box = np.full(3, float(L))
            self.x, self.k, self.hyd, self.hop_on_next = xyz.copy(), k, 7, False
        def get_positions(self): return self.x.copy()
        def set_positions(self, x): self.x = np.array(x, dtype=float)
        def set_box(self, b): self.spec.box = np.array(b, dtype=float)
        def topology(self): return {"hydronium_molecule": self.hyd}
        def calculate_total_force_energy(self, want_force=True): return {"potential": self.k * (self.spec.box[0] - 20.0) ** 2}, None
        def ms_evb_calculate_total_force_energy(self, want_force=True):
            if self.hop_on_next:
                self.hyd += 1
                self.hop_on_next = False
            return self.k * (self.spec.box[0] - 20.0) ** 2, None

    rng = np.random.default_rng(1)
    mol_atoms = [[0, 1, 2], [3, 4, 5], [6, 7, 8, 9]]
    mass = np.array([16.0, 1.0, 1.0] * 2 + [16.0, 1.0, 1.0, 1.0])
    xyz = rng.uniform(0, 20, size=(10, 3))
    xs = barostat.MCBarostat.scaled_positions(xyz, mass, mol_atoms, 1.05)
    for a in mol_atoms:
        com0 = (xyz[a] * mass[a, None]).sum(0) / mass[a].sum()
        com1 = (xs[a] * mass[a, None]).sum(0) / mass[a].sum()
        assert np.allclose(com1, 1.05 * com0) and np.allclose(xs[a] - com1, xyz[a] - com0)      # rigid molecules, scaled centres
    # a steep well around L = 20: moves away from it are rejected and undone, moves towards it accepted (w < 0)
    h = Stub(xyz, 21.0, 1.0e4)
    b = barostat.MCBarostat(1.0, 300.0, baroscale=0.01, seed=3)
    e = h.calculate_total_force_energy()[0]["potential"]
    n_acc = 0
    for _ in range(40):
        L0, x0 = h.spec.box[0], h.get_positions()
        acc, e1 = b.move(h, e, False, mol_atoms, mass, 3)
        kT = barostat.BOLTZMANN * 300.0
        w = 1.0e4 * (b.last["new_len"] - 20.0) ** 2 - e + barostat.BAR_TO_KJMOL_A3 * 1.0 * (b.last["new_len"] ** 3 - L0 ** 3) - 3 * kT * 3.0 * np.log(b.last["new_len"] / L0)
        assert abs(w - b.last["w"]) < 1e-9 * max(1.0, abs(w))
        if acc:
            assert h.spec.box[0] == b.last["new_len"] and abs(e1 - 1.0e4 * (h.spec.box[0] - 20.0) ** 2) < 1e-9
            n_acc += 1
        else:
            assert h.spec.box[0] == L0 and np.array_equal(h.get_positions(), x0) and e1 == e
        e = e1
    assert 0 < n_acc < 40 and abs(h.spec.box[0] - 20.0) < abs(21.0 - 20.0)
    # at the bottom of a much steeper well nearly every move is rejected: after more than 10 trials below 25 % the step shrinks by 1.1
    h3 = Stub(xyz, 20.0, 1.0e9)
    b3 = barostat.MCBarostat(1.0, 300.0, baroscale=0.01, seed=5)
    e3 = 0.0
    for _ in range(11):
        _acc, e3 = b3.move(h3, e3, False, mol_atoms, mass, 3)
    assert abs(b3.baroscale - 0.01 / 1.1) < 1e-15 and b3.n_trials == 0
    # a trial call that moves the excess proton is accepted whatever its weight
    h2 = Stub(xyz, 20.0, 1.0e6)
    h2.hop_on_next = True
    b2 = barostat.MCBarostat(1.0, 300.0, baroscale=0.05, seed=4)
    acc, _ = b2.move(h2, 0.0, True, mol_atoms, mass, 3)
    assert acc and b2.last["w"] == -1.0
    # non-cubic boxes are refused, as in the reference
    h2.spec.box = np.array([20.0, 20.0, 21.0])
    with pytest.raises(ValueError, match="non-cubic"):
        b2.move(h2, 0.0, True, mol_atoms, mass, 3)


def test_velocity_sampling_with_frozen_atoms():
    """src/md_integration.f90:89-114, :151-169: frozen atom types get no velocity, carry no momentum, and the temperature count is the free atoms"""
    from reactive_pb_nn_md_amd import synth
    from reactive_pb_nn_md_amd.forcefield import BOLTZMANN, CONV_KJMOL_ANG2PS2GMOL
    rng = np.random.default_rng(0)
    mass = rng.uniform(1.0, 32.0, 300)
    frozen = np.zeros(300, dtype=bool)
    frozen[::3] = True
    v = synth.maxwell_boltzmann(mass, 300.0, 7, frozen)
    assert np.all(v[frozen] == 0.0)
    assert np.abs((mass[:, None] * v).sum(0)).max() < 1e-10
    ke = 0.5 * (mass[:, None] * v * v).sum() / CONV_KJMOL_ANG2PS2GMOL
    assert abs(ke - 1.5 * BOLTZMANN * 300.0 * (~frozen).sum()) < 1e-9 * ke
    v0 = synth.maxwell_boltzmann(mass, 300.0, 7)
    assert np.all(v0 != 0.0) and np.abs((mass[:, None] * v0).sum(0)).max() < 1e-10
