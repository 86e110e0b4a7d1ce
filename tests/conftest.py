import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))   # tests/ may use the oracle (checker only)

GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def grid_diff(g, key, arr):
    """largest deviation of a PME grid from the fixture's: the whole grid, or - 64^3 fixtures - every 61st point of it (+ its sum)"""
    if key in g.files:
        return np.abs(arr - g[key]).max()
    st = int(g[key + "_sample_stride"])
    return max(np.abs(arr.ravel()[::st] - g[key + "_sample"]).max(), abs(arr.sum() - float(g[key + "_sum"])) * 1e-3)


def all_golden(prefix=""):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, prefix + "*.npz")))


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle_py
    oracle_py.build()
    return oracle_py


def pre_switch_spec(g):
    """the state BEFORE the reference's initial principal-diabat switch of a '*_switch_*' fixture: the generating synthetic box
    (gen_* fields) with the force field of oracle/make_golden.py; the fixture itself holds the state after the switch"""
    import dataclasses
    from reactive_pb_nn_md_amd import forcefield, synth, system
    ff = forcefield.ch3so3h_water_forcefield()
    if np.isfinite(float(g["gen_so3h_eref"])):
        ff = dataclasses.replace(ff, reference_energy=dict(ff.reference_energy, so3h=float(g["gen_so3h_eref"])))
    p = forcefield.build_params(ff, moltype_order=["h2o", "so3h"])
    s = synth.make_system(int(g["gen_n_side"]), n_diabat=int(g["gen_n_diabat"]), seed=int(g["gen_seed"]), proton_shift=float(g["gen_proton_shift"]))
    assert np.array_equal(s.xyz, g["gen_xyz"])
    post = system.from_golden(g)
    return system.from_synth(s, p, rc=post.rc, rv=post.rv, alpha=post.alpha, pme_grid=post.pme_grid, nslist=post.nslist, max_chain=post.max_chain)


def reference_atom_order(topo):
    """the reference keeps every molecule's atoms contiguous and re-packs its atom arrays after a proton transfer
    (shift_array_data_donor_acceptor_transfer, src/ms_evb.f90:865); we keep atom indices fixed and edit the molecules' atom lists.
    Returns perm with reference_array[k] <-> our_array[perm[k]]"""
    return np.concatenate([topo["mol_atom_index"][m, :topo["mol_n_atom"][m]] for m in range(len(topo["mol_n_atom"]))])
