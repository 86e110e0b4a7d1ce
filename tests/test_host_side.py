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


def test_spec_from_synth_equals_spec_from_reference_dump():
    g = golden("evb_n6_d2_s1")
    ref = system.from_golden(g)
    s = synth.make_system(int(g["gen_n_side"]), n_diabat=int(g["gen_n_diabat"]), seed=int(g["gen_seed"]))
    p = forcefield.build_params(forcefield.ch3so3h_water_forcefield(), moltype_order=["h2o", "so3h"])
    mine = system.from_synth(s, p, rc=ref.rc, rv=ref.rv, pme_grid=ref.pme_grid, nslist=ref.nslist)
    for key in ("atype", "charge", "mass", "mol_first", "mol_natom", "mol_type", "xyz", "box"):
        assert np.array_equal(getattr(mine, key), getattr(ref, key)), key
    assert mine.hydronium == ref.hydronium


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
