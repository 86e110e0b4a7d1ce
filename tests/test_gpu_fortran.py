"""The Fortran host path on the GPU: a flang-built driver (reactive_pb_nn_md_amd/fortran/fortran_smoke.f90) feeds the same
arrays through the iso_c_binding interfaces (msevb_hip_c.f90) and must reproduce the reference's golden energies."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, golden
from reactive_pb_nn_md_amd import system

pytestmark = pytest.mark.gpu


def write_spec(path, spec):
    p = spec.params
    nt, nmt = p.n_atom_type, len(p.moltypes)
    f32 = lambda a: np.asfortranarray(np.asarray(a, dtype=np.int32)).tobytes(order="F")
    f64 = lambda a: np.asfortranarray(np.asarray(a, dtype=np.float64)).tobytes(order="F")
    ai, ap = spec.angle_tables()
    ai = ai.copy(); ai[:, :3] += 1
    di, dp = spec.dihedral_tables()
    di = di.copy(); di[:, :4] += 1
    with open(path, "wb") as f:
        f.write(f32([spec.n_atom, spec.n_mole, nt, nmt, spec.pme_grid]))
        f.write(f64(np.diag(spec.box)) + f64([spec.rc, spec.rv, spec.alpha]))
        f.write(f32(spec.atype + 1) + f64(spec.charge) + f64(spec.mass) + f32(spec.mol_first + 1) + f32(spec.mol_natom) + f32(spec.mol_type + 1) + f64(spec.xyz.T))
        f.write(f32(p.vdw_type) + f64(p.vdw_param) + f64(p.vdw_param14) + f32(p.bond_type) + f64(p.bond_param) + f64(p.atype_chg))
        f.write(f32([len(ai)]) + f32(ai.T) + f64(ap.T))
        f.write(f32([len(di)]) + f32(di.T) + f64(dp.T))
        for m in p.moltypes:
            f.write(f32([m["n_atom"], len(m["bonds"]), len(m["angles"]), len(m["dihedrals"])]))
            f.write(f32(m["atype"] + 1))
            if len(m["bonds"]):
                f.write(f32((m["bonds"] + 1).T))
            if len(m["angles"]):
                f.write(f32((m["angles"] + 1).T))
            if len(m["dihedrals"]):
                f.write(f32((m["dihedrals"] + 1).T))
            f.write(f32(m["excl"]) + f32(m["reactive_protons"]) + f32(m["reactive_basic"]))


def test_fortran_driver_reproduces_reference_energies(tmp_path):
    exe = os.path.join(ROOT, "reactive_pb_nn_md_amd", "fortran", "fortran_smoke")
    if not os.path.exists(exe):
        pytest.skip("fortran_smoke not built (flang missing at build time)")
    g = golden("std_n6_s3")
    spec = system.from_golden(g)
    path = str(tmp_path / "spec.bin")
    write_spec(path, spec)
    r = subprocess.run([exe, path], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout
    en = fs = None
    for line in r.stdout.splitlines():
        if line.startswith("ENERGIES"):
            en = [float(x) for x in line.split()[1:]]
        if line.startswith("FORCESUM"):
            fs = [float(x) for x in line.split()[1:]]
    assert en is not None and fs is not None, r.stdout
    ref = [g["E_total"][0], g["E_elec_total"][0], g["E_vdw_total"][0], g["E_bond"][0], g["E_angle"][0], g["E_dihedral"][0]]
    for a, b in zip(en, ref):
        assert abs(a - b) < 1e-11 * max(abs(b), 1.0), (en, ref)
    assert np.allclose(fs, np.abs(g["F_total"]).sum(0), rtol=1e-12)
