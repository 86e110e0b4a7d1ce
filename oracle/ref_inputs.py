"""TEST INFRASTRUCTURE ONLY (oracle/): writes a synthetic system in the four input formats the
reference's parsers accept, and runs oracle/_ref/ref_dump_{std,evb} on it.

Formats follow the *parsers*, not the (stale) example files:
  .gro          src/general_routines.f90:184-315  '(I5,2A5,I5,3F8.3)' nm
  force field   src/initialize_routines.f90:281-430, :646-691  (5 tokens per atom type)
  topology      src/intra_bonded_interactions.f90:696-1464, src/ms_evb.f90:3170-3623
  sim params    src/read_simulation_parameters.f90:43-146
"""
from __future__ import annotations

import os
import struct
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def write_gro(path, system):
    with open(path, "w") as f:
        f.write("synthetic\n")
        f.write("%d\n" % system.n_atom)
        k = 0
        for im in range(system.n_mole):
            for _ in range(system.mol_natom[im]):
                x = system.xyz[k] / 10.0
                f.write("%5d%-5s%5s%5d%8.3f%8.3f%8.3f\n" % ((im + 1) % 100000, system.mol_names[im],
                                                             system.atom_names[k], (k + 1) % 100000, x[0], x[1], x[2]))
                k += 1
        f.write("%10.5f%10.5f%10.5f\n" % tuple(system.box / 10.0))


def write_ffpmt(path, ff):
    with open(path, "w") as f:
        f.write("solute_species\natom_type_parameters\n%d\n" % len(ff.atypes))
        for a in ff.atypes:
            f.write("%s %r %r %r %d\n" % (a[0], a[1], a[2], a[3], 1 if a[0] in set(getattr(ff, "freeze", [])) else 0))
        if ff.sapt:
            f.write("\ncustom_sapt_parameters\nname Aex Ael Aind Adhf B C6 C8 C10 C12\n")
            for row in ff.sapt:
                f.write(row[0] + " " + " ".join(repr(float(x)) for x in row[1:]) + "\n")
        f.write("\ncross_terms  c6 c12\n%d\n" % len(ff.cross_terms))
        for c in ff.cross_terms:
            f.write("%d %d %r %r 0\n" % c)
        f.write("\npairtypes  C6 C12\n%d\n" % len(ff.pairtypes))
        for p in ff.pairtypes:
            f.write("%s %s %r %r\n" % p)


def write_top(path, ff):
    with open(path, "w") as f:
        f.write("[ bondtypes ]\n")
        for b in ff.bondtypes:
            f.write(" ".join(str(x) for x in b) + "\n")
        f.write("\n[ angletypes ]\n")
        for a in ff.angletypes:
            f.write(" ".join(str(x) for x in a) + "\n")
        f.write("\n[ dihedraltypes ]\n")
        for d in ff.dihedraltypes:
            f.write(" ".join(str(x) for x in d) + "\n")
        f.write("\n[ evb_parameters ]\n[ reference_energy ]\n")
        for k, v in ff.reference_energy.items():
            f.write("%s %r\n" % (k, v))
        f.write("\n[ adiabat_non_bond ]\n[ donor_acceptor ]\n")
        for r in ff.donor_acceptor:
            f.write(" ".join(str(x) for x in r) + "\n")
        f.write("\n[ proton_acceptor ]\n")
        for r in ff.proton_acceptor:
            f.write(" ".join(str(x) for x in r) + "\n")
        f.write("\n[ diabat_coupling ]\n[ geometry_factor ]\n")
        for r in ff.geometry_factor:
            f.write("%s %s %s %d\n" % r[:4])
            f.write(" ".join(repr(x) for x in r[4]) + "\n")
        f.write("\n[ exchange_charge_atomic ]\n")
        for k, v in ff.exchange_charge_atomic.items():
            f.write("%s %r\n" % (k, v))
        f.write("\n[ exchange_charge_proton ]\n")
        for r in ff.exchange_charge_proton:
            f.write("%s %s %r\n" % r)
        f.write("\n")
        for mt in ff.moltypes:
            f.write("[ moleculetype ]\n%s 3\n\n[ atoms ]\n" % mt.name)
            for i, a in enumerate(mt.atoms):
                f.write("%d %s %r\n" % (i + 1, a[0], a[1]))
            f.write("\n[ bonds ]\n")
            for b in mt.bonds:
                f.write("%d %d 1\n" % b)
            f.write("\n[ angles ]\n")
            for a in mt.angles:
                f.write("%d %d %d 1\n" % a)
            f.write("\n[ dihedrals ]\n")
            for d in mt.dihedrals:
                f.write("%d %d %d %d 1\n" % d)
            f.write("\n[ exclusions ]\n")
            for e in mt.exclusions:
                f.write("%d %d\n" % e)
            f.write("\n")
        f.write("[ evb_topology ]\n\n")
        for p in ff.evb_pairs:
            f.write("[ evb_pairs ]\n%s %s %s %s\n\n" % (p.acid, p.base, p.proton_type, p.heavy_type))
            for head, lst in (("acid_reactive_protons", p.acid_reactive_protons), ("base_reactive_protons", p.base_reactive_protons),
                              ("acid_acceptor_atoms", p.acid_acceptor_atoms), ("base_acceptor_atoms", p.base_acceptor_atoms)):
                f.write("[ %s ]\n" % head)
                for i, v in enumerate(lst):
                    f.write("%d %d\n" % (i + 1, v))
                f.write("\n")
            f.write("[ conjugate_atoms ]\n")
            for c in p.conjugate_atoms:
                f.write("%s %s\n" % c)
            f.write("\n")


def write_simpmt(path, *, n_step=1, rc=8.0, rv=9.5, nslist=10, alpha=0.3, pme_grid=32, n_exclusions=2,
                 delta_t=0.0005, temperature=300.0, lj_comb_rule="opls", ensemble="NVE", friction_coeff=None, pressure=None, baroscale=None):
    with open(path, "w") as f:
        f.write("Simulation Methodology\nensemble %s\nlj_comb_rule %s\ngrid_Tang_Toennies yes\n" % (ensemble, lj_comb_rule))
        f.write("Simulation Parameters\n")
        extra = [(k, v) for k, v in (("friction_coeff", friction_coeff), ("pressure", pressure), ("baroscale", baroscale)) if v is not None]
        for k, v in [("n_step", n_step), ("n_output", 1000000), ("n_exclusions", n_exclusions)] + extra + list((
                     ("temperature", temperature), ("delta_t", delta_t), ("real_space_cutoff", rc),
                     ("na_nslist", nslist), ("nb_nslist", nslist), ("nc_nslist", nslist),
                     ("verlet_cutoff", rv), ("alpha_sqrt", alpha), ("pme_grid", pme_grid),
                     ("spline_order", 6), ("n_threads", 1), ("debug", 0))):
            f.write("%s %r\n" % (k, v))


def read_dump(path):
    """Tagged binary stream written by oracle/ref_driver.f90 -> dict of numpy arrays (Fortran order kept:
    an array dumped with dims (3,N) comes back with shape (N,3), i.e. C view of column-major data)."""
    out = {}
    with open(path, "rb") as f:
        data = f.read()
    off = 0
    while off < len(data):
        name = data[off:off + 32].decode().strip()
        off += 32
        tcode, nd = struct.unpack_from("<ii", data, off)
        off += 8
        dims = struct.unpack_from("<4i", data, off)[:nd]
        off += 16
        n = int(np.prod(dims)) if nd > 0 else 1
        if tcode == 1:
            arr = np.frombuffer(data, dtype="<f8", count=n, offset=off).copy()
            off += 8 * n
        else:
            arr = np.frombuffer(data, dtype="<i4", count=n, offset=off).copy()
            off += 4 * n
        out[name] = arr.reshape(tuple(reversed(dims)))
    return out


def run_reference(system, ff, *, evb, workdir=None, velocities=None, nsteps=0, keep=False, nbaro=0, **simkw):
    """Runs the real reference (oracle/_ref/ref_dump_*) on ``system`` and returns its dump."""
    exe = os.path.join(HERE, "_ref", "ref_dump_evb" if evb else "ref_dump_std")
    if not os.path.exists(exe):
        raise FileNotFoundError(exe + " (run oracle/build_ref.sh in the dev container)")
    tmp = workdir or tempfile.mkdtemp(prefix="refrun_")
    os.makedirs(tmp, exist_ok=True)
    write_gro(os.path.join(tmp, "sys.gro"), system)
    write_ffpmt(os.path.join(tmp, "ff.pmt"), ff)
    write_top(os.path.join(tmp, "sys.top"), ff)
    write_simpmt(os.path.join(tmp, "sim.pmt"), **simkw)
    args = [exe, "sys.gro", "ff.pmt", "sys.top", "sim.pmt", "traj.gro", "out.log", "hop.out", "a.out", "dump.bin"]
    for fn in ("hop.out", "traj.gro", "out.log", "a.out", "dump.bin"):
        p = os.path.join(tmp, fn)
        if os.path.exists(p):
            os.remove(p)
    if nsteps > 0:
        np.ascontiguousarray(velocities, dtype="<f8").tofile(os.path.join(tmp, "vel.bin"))
        args += ["vel.bin", str(nsteps)] + ([str(nbaro)] if nbaro else [])
    env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1")
    r = subprocess.run("ulimit -s unlimited; exec " + " ".join(args), shell=True, cwd=tmp, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0 or "ref_dump finished OK" not in r.stdout:
        raise RuntimeError("reference run failed in %s:\n%s" % (tmp, r.stdout[-3000:]))
    d = read_dump(os.path.join(tmp, "dump.bin"))
    if nsteps > 0:      # what the reference's own writers left behind (text, as bytes)
        for key, fn in (("file_traj", "traj.gro"), ("file_log", "out.log"), ("file_vel", "vel_checkpoint.out")):
            with open(os.path.join(tmp, fn), "rb") as fh:
                d[key] = np.frombuffer(fh.read(), dtype=np.uint8).copy()
    if not keep and workdir is None:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
    return d
