"""The Fortran host path on the GPU: a flang-built driver (reactive_pb_nn_md_amd/fortran/fortran_smoke.f90, ours - nothing of the reference is
linked into it) feeds the arrays of the committed fixtures through the iso_c_binding interfaces (msevb_hip_c.f90) and must reproduce the
numbers the real reference computed for them: a plain force call, an MS-EVB force call, the same call with the host arrays in another atom
order (msevb_set_host_order: the host side of a proton hop), and a block of md_integrate_atomic.

The tests further down run executables that contain REFERENCE objects (oracle/_ref/, built by `make link_check` in the dev container).  They
skip wherever /root/reference/src is absent - whether or not oracle/_ref/ happens to be in the tree that was pushed to the box (the driver's push
does not read .gpurunignore) - so nothing derived from the reference is executed on a GPU box; they run where a GPU and the reference are both
present."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, golden      # conftest puts oracle/ on sys.path (test infrastructure)
from reactive_pb_nn_md_amd import system

pytestmark = pytest.mark.gpu

REFERENCE_SRC = "/root/reference/src"


def ref_exe(name):
    """path of an executable under oracle/_ref/ that contains reference objects; skips the test unless the reference tree itself is present"""
    if not os.path.isdir(REFERENCE_SRC):
        pytest.skip("executables with reference objects only run where %s exists (dev container), never on a GPU box" % REFERENCE_SRC)
    exe = os.path.join(ROOT, "oracle", "_ref", name)
    if not os.path.exists(exe):
        pytest.skip(name + " not built (oracle/build_ref.sh / make link_check)")
    return exe

SMOKE = os.path.join(ROOT, "reactive_pb_nn_md_amd", "fortran", "fortran_smoke")


def write_spec(path, spec, evb=False, md=None):
    """md = dict(nsteps, dt, use_evb, vel (N,3), perm (N,) 0-based host_to_library) or None"""
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
        if not evb and md is None:
            return
        f.write(f32([1 if evb else 0]))
        if evb:      # the argument list of msevb_set_evb, in its order (api.py: _set_evb)
            e, ldi = p.evb, 15

            def table(rows, nty, npar):
                ti, tp = np.zeros((ldi, nty), dtype=np.int32), np.zeros((ldi, npar))
                for r, (tys, par) in enumerate(rows):
                    ti[r, :] = np.array(tys) + 1
                    tp[r, :len(par)] = par
                return f32(ti) + f64(tp)
            cp_t = np.zeros(ldi, dtype=np.int32)
            cp_t[:len(e["coupling"])] = [c[1] for c in e["coupling"]]
            f.write(f32([ldi]) + f32(e["acid"]) + f32(e["base"]) + f32(e["conjugate_mol"] + 1) + f32(e["proton_type"] + 1) + f32(e["heavy_type"] + 1)
                    + f32(e["conjugate_atype"] + 1) + f64(e["reference_energy"]) + f64(e["exch_atomic"]) + f64(e["exch_proton"])
                    + table(e["donor_acceptor"], 3, 6) + table(e["proton_acceptor"], 2, 5) + table([(c[0], c[2]) for c in e["coupling"]], 3, 10)
                    + f32(cp_t) + f32([int(spec.hydronium) + 1, int(spec.max_chain)]))
        f.write(f32([1 if md is not None else 0]))
        if md is not None:
            f.write(f32([md["nsteps"]]) + f64([md["dt"]]) + f32([1 if md["use_evb"] else 0]) + f64(np.asarray(md["vel"]).T) + f32(np.asarray(md["perm"]) + 1))


def run_smoke(tmp_path, spec, **kw):
    if not os.path.exists(SMOKE):
        pytest.skip("fortran_smoke not built (flang missing at build time)")
    path, out = str(tmp_path / "spec.bin"), str(tmp_path / "state.bin")
    write_spec(path, spec, **kw)
    r = subprocess.run([SMOKE, path, out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and "fortran_smoke done" in r.stdout, r.stdout[-3000:]
    rec = {}
    for line in r.stdout.splitlines():
        t = line.split()
        if t and t[0] in ("ENERGIES", "FORCESUM", "EVB", "EVBFORCESUM", "HOSTORDER", "HOSTORDERXYZ"):
            rec[t[0]] = [float(x) for x in t[1:]]
        if t and t[0] == "MD":
            rec.setdefault("MD", []).append([float(t[2]), float(t[3])])
    return rec, out


def test_fortran_driver_reproduces_reference_energies(tmp_path):
    g = golden("std_n6_s3")
    spec = system.from_golden(g)
    rec, _ = run_smoke(tmp_path, spec)
    en, fs = rec["ENERGIES"], rec["FORCESUM"]
    ref = [g["E_total"][0], g["E_elec_total"][0], g["E_vdw_total"][0], g["E_bond"][0], g["E_angle"][0], g["E_dihedral"][0]]
    for a, b in zip(en, ref):
        assert abs(a - b) < 1e-11 * max(abs(b), 1.0), (en, ref)
    assert np.allclose(fs, np.abs(g["F_total"]).sum(0), rtol=1e-12)


def test_fortran_driver_evb_call_and_host_atom_order(tmp_path):
    """ms_evb_calculate_total_force_energy through the Fortran bindings on fixture evb_n6_d4_s2 (adiabatic energy, diabat count and the
    Hellmann-Feynman force of the real reference), then the same call after msevb_set_host_order with a random permutation of the host arrays:
    forces come back permuted, positions read back as they were sent, the energy does not move."""
    g = golden("evb_n6_d4_s2")
    spec = system.from_golden(g)
    perm = np.random.default_rng(7).permutation(spec.n_atom)
    rec, out = run_smoke(tmp_path, spec, evb=True, md=dict(nsteps=0, dt=spec.delta_t, use_evb=True, vel=np.zeros((spec.n_atom, 3)), perm=perm))
    e, D = rec["EVB"][0], int(rec["EVB"][1])
    assert D == int(g["n_diabat"][0])
    assert abs(e - g["E_adiabatic"][0]) < 1e-10 * abs(g["E_adiabatic"][0])
    F = np.fromfile(out + ".evb").reshape(spec.n_atom, 3)
    assert np.abs(F - g["F_adiabatic"]).max() < 1e-8
    assert rec["HOSTORDER"][0] < 1e-9 and rec["HOSTORDER"][1] < 1e-10 * abs(e) and rec["HOSTORDERXYZ"][0] == 0.0


@pytest.mark.parametrize("name,evb", [("md_std_n6_s4", False), ("md_evb_n6_d2_s5", True)])
def test_fortran_driver_md_block_against_reference_trajectory(tmp_path, name, evb):
    """msevb_md_integrate_atomic from Fortran: the reference's own trajectory fixtures (energies per step, end positions, velocities, forces)."""
    g = golden(name)
    spec = system.from_golden(g)
    n = int(g["md_nsteps"][0])
    rec, out = run_smoke(tmp_path, spec, evb=evb, md=dict(nsteps=n, dt=spec.delta_t, use_evb=evb, vel=g["md_velocity0"], perm=np.arange(spec.n_atom)))
    en = np.array(rec["MD"])
    assert en.shape == (n, 2)
    assert np.abs(en[:, 0] - g["md_energy"][:, 0]).max() < 1e-9 * np.abs(g["md_energy"][:, 0]).max()
    assert np.abs(en[:, 1] - g["md_energy"][:, 1]).max() < 1e-8 * np.abs(g["md_energy"][:, 1]).max()
    st = np.fromfile(out).reshape(3, spec.n_atom, 3)
    assert np.abs(st[0] - g["md_xyz"]).max() < 1e-10 and np.abs(st[1] - g["md_velocity"]).max() < 1e-9 and np.abs(st[2] - g["md_force"]).max() < 1e-7


def _parse_log(path):
    """print_step blocks (src/general_routines.f90:870-901): {step: (PE, KE, [elec, vdw, bond, angle, dihedral] or None)}"""
    out, lines = {}, open(path).read().splitlines()
    k = 0
    while k < len(lines):
        if lines[k].strip().startswith("i_step"):
            t = lines[k + 1].split()
            step, pe, ke = int(t[0]), float(t[2]), float(t[3])
            comps = None
            if k + 3 < len(lines) and lines[k + 2].strip().startswith("Electrostatic"):
                comps = [float(x) for x in lines[k + 3].split()]
            out[step] = (pe, ke, comps)
        k += 1
    return out


@pytest.mark.parametrize("name,evb", [("std_n6_s3", False), ("evb_n6_d4_s2", True)])
def test_unchanged_reference_main_program_on_the_gpu_library(tmp_path, name, evb):
    """The drop-in claim itself.  oracle/_ref/main_ms_evb_hip_{std,evb} is the reference's UNCHANGED main_ms_evb.f90 + md_integration.f90 +
    initialize_routines.f90 + parsers, linked by `make -C reactive_pb_nn_md_amd/fortran link_check` against the drop-in modules
    total_energy_forces / ms_evb (reactive_pb_nn_md_amd/fortran/) and libmsevb_hip.so + libmsevb_dfti.so - no MKL, no CPU force code on the
    path.  It is run on the input files of a golden fixture, with the reference's own command line, for a few MD steps; its step-0 log
    block (velocity independent) must show the energies the real reference computed for that fixture (print_step writes 6 digits), and the
    steps after it must stay on the energy shell as well as the reference's own integrator does on this box."""
    exe = ref_exe("main_ms_evb_hip_" + ("evb" if evb else "std"))
    import ref_inputs      # oracle/: writes the four input files in the formats the reference's parsers accept
    from reactive_pb_nn_md_amd import forcefield, synth
    g = golden(name)
    s = synth.make_system(int(g["gen_n_side"]), n_diabat=int(g["gen_n_diabat"]), seed=int(g["gen_seed"]))
    assert np.array_equal(s.xyz, g["gen_xyz"])
    ff = forcefield.ch3so3h_water_forcefield()
    ref_inputs.write_gro(str(tmp_path / "sys.gro"), s)
    ref_inputs.write_ffpmt(str(tmp_path / "ff.pmt"), ff)
    ref_inputs.write_top(str(tmp_path / "sys.top"), ff)
    ref_inputs.write_simpmt(str(tmp_path / "sim.pmt"), n_step=6, rc=7.0, rv=8.0, nslist=16, pme_grid=16)
    sim = open(tmp_path / "sim.pmt").read().replace("n_output 1000000", "n_output 1")
    open(tmp_path / "sim.pmt", "w").write(sim)
    r = subprocess.run("ulimit -s unlimited; exec %s sys.gro ff.pmt sys.top sim.pmt traj.gro out.log hop.out a.out" % exe, shell=True, cwd=str(tmp_path),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    log = _parse_log(str(tmp_path / "out.log"))
    assert sorted(log) == list(range(7)), (sorted(log), r.stdout[-2000:])
    pe0, ke0, comps = log[0]
    close = lambda a, b: abs(a - b) <= 1e-5 * max(abs(b), 1.0)      # print_step writes E16.6: six significant digits
    if evb:
        assert close(pe0, g["E_adiabatic"][0]), (pe0, g["E_adiabatic"][0])
    else:
        assert close(pe0, g["E_total"][0]), (pe0, g["E_total"][0])
        for a, b in zip(comps, [g["E_elec_total"][0], g["E_vdw_total"][0], g["E_bond"][0], g["E_angle"][0], g["E_dihedral"][0]]):
            assert close(a, b), (comps, b)
    etot = np.array([log[k][0] + log[k][1] for k in range(7)])
    # NVE, 0.5 fs steps from clock-seeded velocities (not reproducible).  The unrelaxed lattice converts ~4 000 kJ/mol per step; the real
    # reference's own trajectories on these boxes drift by 2 % in the first steps (tests/golden/md_std_n6_s4: 17 665 -> 17 318 in 4 steps)
    assert np.all(np.isfinite(etot)) and np.abs(etot - etot[0]).max() < 0.05 * abs(etot[0]), etot
    frames = open(tmp_path / "traj.gro").read().count("step")
    assert frames >= 7 or os.path.getsize(tmp_path / "traj.gro") > 7 * 30 * s.n_atom


def test_fortran_stage_procedures_against_the_reference_on_the_cpu(tmp_path):
    """oracle/_ref/stages_check (reactive_pb_nn_md_amd/fortran/stages_check.f90, built by `make link_check`): in ONE Fortran process, on the
    data structures the reference's own start-up builds from the input files of fixture std_n6_s3, the reference's stage procedures on the
    CPU against the procedures of the same names and argument lists in module msevb_hip_stages (GPU): real_space_energy_force after
    construct_verlet_list, pme_reciprocal_space_energy_force (forces, E_recip, Q_grid, theta_conv_Q), grid_Q / derivative_grid_Q,
    intra_molecular_energy_force, and one step of md_integrate_atomic."""
    exe = ref_exe("stages_check")
    import ref_inputs
    from reactive_pb_nn_md_amd import forcefield, synth
    g = golden("std_n6_s3")
    s = synth.make_system(int(g["gen_n_side"]), n_diabat=int(g["gen_n_diabat"]), seed=int(g["gen_seed"]))
    ff = forcefield.ch3so3h_water_forcefield()
    ref_inputs.write_gro(str(tmp_path / "sys.gro"), s)
    ref_inputs.write_ffpmt(str(tmp_path / "ff.pmt"), ff)
    ref_inputs.write_top(str(tmp_path / "sys.top"), ff)
    ref_inputs.write_simpmt(str(tmp_path / "sim.pmt"), n_step=1, rc=7.0, rv=8.0, nslist=16, pme_grid=16)
    r = subprocess.run("ulimit -s unlimited; exec %s sys.gro ff.pmt sys.top sim.pmt traj.gro out.log hop.out a.out" % exe, shell=True, cwd=str(tmp_path),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and "stages_check done" in r.stdout, r.stdout[-3000:]
    st = {ln.split()[1]: [float(x) for x in ln.split() if x[0] in "-0123456789."] for ln in r.stdout.splitlines() if ln.startswith("stage ")}
    assert set(st) == {"real_space", "pme", "grid_Q_derivative", "bonded", "md_step"}, r.stdout[-2000:]
    assert st["real_space"][0] < 1e-9 and st["real_space"][1] < 1e-9 * max(abs(st["real_space"][2]), abs(st["real_space"][3]))
    assert st["pme"][0] < 1e-9 and st["pme"][1] < 1e-10 * abs(st["pme"][4]) and st["pme"][2] < 1e-13 and st["pme"][3] < 1e-12
    assert st["grid_Q_derivative"][0] < 1e-9 and st["grid_Q_derivative"][1] < 1e-13
    assert st["bonded"][0] < 1e-9 and st["bonded"][1] < 1e-10 * max(abs(st["bonded"][2]), abs(st["bonded"][3]), 1.0)
    assert st["md_step"][0] < 1e-10 and st["md_step"][1] < 1e-9 and st["md_step"][2] < 1e-8


def test_unchanged_reference_main_program_across_a_proton_hop(tmp_path):
    """The same executable on the box of fixture md_evb_n6_d4_switch_s8, whose FIRST EVB force call makes another diabat the principal one
    (a property of the positions: velocity independent): the reference's own evb_change_diabat_data_structure_topology re-packs the host arrays,
    the drop-in module hands the new atom order to the library (msevb_set_host_order: no context tear-down, no re-upload of tables) and the run
    goes on.  Checked: the hop is logged at step 0 with the molecules the real reference hopped between, the step-0 energy is the reference's
    adiabatic energy, and the steps AFTER the hop stay on the energy shell (a wrong permutation of positions or forces does not)."""
    exe = ref_exe("main_ms_evb_hip_evb")
    import dataclasses
    import ref_inputs
    from reactive_pb_nn_md_amd import forcefield, synth
    g = golden("md_evb_n6_d4_switch_s8")
    s = synth.make_system(int(g["gen_n_side"]), n_diabat=int(g["gen_n_diabat"]), seed=int(g["gen_seed"]), proton_shift=float(g["gen_proton_shift"]))
    assert np.array_equal(s.xyz, g["gen_xyz"])
    ff = forcefield.ch3so3h_water_forcefield()
    ff = dataclasses.replace(ff, reference_energy=dict(ff.reference_energy, so3h=float(g["gen_so3h_eref"])))
    ref_inputs.write_gro(str(tmp_path / "sys.gro"), s)
    ref_inputs.write_ffpmt(str(tmp_path / "ff.pmt"), ff)
    ref_inputs.write_top(str(tmp_path / "sys.top"), ff)
    ref_inputs.write_simpmt(str(tmp_path / "sim.pmt"), n_step=8, rc=7.0, rv=8.0, nslist=16, pme_grid=16)
    sim = open(tmp_path / "sim.pmt").read().replace("n_output 1000000", "n_output 1")
    open(tmp_path / "sim.pmt", "w").write(sim)
    r = subprocess.run("ulimit -s unlimited; exec %s sys.gro ff.pmt sys.top sim.pmt traj.gro out.log hop.out a.out" % exe, shell=True, cwd=str(tmp_path),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "msevb_hip_remap: host atom order handed to the GPU library" in r.stdout, r.stdout[-3000:]
    hop = open(tmp_path / "hop.out").read()      # (written only with print_ms_evb_data = yes, src/ms_evb.f90:329-335)
    if hop.strip():
        old, new = int(g["gen_hydronium"]), int(g["hydronium_molecule"][0])      # 1-based molecule numbers before / after the reference's own switch
        line = hop.split("proton hop from")[1].split("\n")[0]
        assert str(old) in line and str(new) in line, hop[:400]
    ms = float(r.stdout.split("handed to the GPU library in")[1].split("ms")[0])
    assert r.stdout.count("msevb_hip_remap:") >= 1 and ms < 50.0, r.stdout[-2000:]
    log = _parse_log(str(tmp_path / "out.log"))
    assert sorted(log) == list(range(9)), (sorted(log), r.stdout[-2000:])
    assert abs(log[0][0] - g["E_adiabatic"][0]) <= 1e-5 * abs(g["E_adiabatic"][0]), (log[0][0], g["E_adiabatic"][0])
    etot = np.array([log[k][0] + log[k][1] for k in range(9)])
    assert np.all(np.isfinite(etot)) and np.abs(etot - etot[0]).max() < 0.05 * abs(etot[0]), etot


def test_fortran_evb_force_call_against_the_reference_routine(tmp_path):
    """oracle/_ref/stages_check_evb (the EVB build of stages_check.f90): in ONE Fortran process the reference's own ms_evb_calculate_total_force_energy
    (linked beside the drop-in as ms_evb_calculate_total_force_energy_cpu: its diabat enumeration, topology hand-overs, per-diabat real-space and
    reciprocal-space updates, couplings and Jacobi all run on the CPU) against the drop-in's nine-argument procedure of the same name, on fixture
    evb_n6_d4_s2's input files: forces to 1e-9 kJ/mol/A, the adiabatic energy and every Hamiltonian entry to 1e-10 relative."""
    exe = ref_exe("stages_check_evb")
    import ref_inputs
    from reactive_pb_nn_md_amd import forcefield, synth
    g = golden("evb_n6_d4_s2")
    s = synth.make_system(int(g["gen_n_side"]), n_diabat=int(g["gen_n_diabat"]), seed=int(g["gen_seed"]))
    ff = forcefield.ch3so3h_water_forcefield()
    ref_inputs.write_gro(str(tmp_path / "sys.gro"), s)
    ref_inputs.write_ffpmt(str(tmp_path / "ff.pmt"), ff)
    ref_inputs.write_top(str(tmp_path / "sys.top"), ff)
    ref_inputs.write_simpmt(str(tmp_path / "sim.pmt"), n_step=1, rc=7.0, rv=8.0, nslist=16, pme_grid=16)
    r = subprocess.run("ulimit -s unlimited; exec %s sys.gro ff.pmt sys.top sim.pmt traj.gro out.log hop.out a.out" % exe, shell=True, cwd=str(tmp_path),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and "stages_check done" in r.stdout, r.stdout[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("stage evb")]
    assert line, r.stdout[-3000:]
    t = line[0].split()
    dF, dE, dH, d_ref, d_gpu, e = float(t[5]), float(t[6]), float(t[7]), int(t[8]), int(t[9]), float(t[10])
    assert d_ref == d_gpu == int(g["n_diabat"][0])
    assert dF < 1e-9 and dE < 1e-10 * abs(e) and dH < 1e-10 * abs(e), line[0]
