#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY (oracle/): a LONG trajectory of the real reference (oracle/_ref/ref_dump_evb) on a liquid-like 12^3-site box, run until the reference
itself stops ('force on atom too big', src/md_integration.f90:523) or the step budget ends; what it leaves is its per-step history (oracle/ref_driver.f90: md_trace.txt:
potential and kinetic energy, Verlet flag, diabat count, largest force, hydronium molecule).  Committed as tests/golden/long_evb_*.npz: the GPU path has to walk the same
surface step by step (whose collapse is it?).

    python oracle/make_collapse_trace.py [n_steps] [temperature] [seed]
"""
import os, subprocess, sys, tempfile
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
import ref_inputs  # noqa: E402
from reactive_pb_nn_md_amd import forcefield, synth  # noqa: E402

n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
temp = float(sys.argv[2]) if len(sys.argv) > 2 else 300.0
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 40
n_side, n_diabat = 12, 4
ff = forcefield.ch3so3h_water_forcefield()
s = synth.make_system(n_side, n_diabat=n_diabat, seed=seed, liquid_like=True)
p = forcefield.build_params(ff, moltype_order=["h2o", "so3h"])
idx = {n: i for i, n in enumerate(p.atype_names)}
mass = np.array([p.atype_mass[idx[a]] for a in s.atom_names])
vel = synth.maxwell_boltzmann(mass, temp, seed + 100)
simkw = dict(rc=7.0, rv=8.0, nslist=16, pme_grid=32)
tmp = tempfile.mkdtemp(prefix="refrun_long_")
exe = os.path.join(HERE, "_ref", "ref_dump_evb")
ref_inputs.write_gro(os.path.join(tmp, "sys.gro"), s); ref_inputs.write_ffpmt(os.path.join(tmp, "ff.pmt"), ff); ref_inputs.write_top(os.path.join(tmp, "sys.top"), ff)
ref_inputs.write_simpmt(os.path.join(tmp, "sim.pmt"), **simkw)
np.ascontiguousarray(vel, dtype="<f8").tofile(os.path.join(tmp, "vel.bin"))
args = [exe, "sys.gro", "ff.pmt", "sys.top", "sim.pmt", "traj.gro", "out.log", "hop.out", "a.out", "dump.bin", "vel.bin", str(n_steps)]
env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1")
r = subprocess.run("ulimit -s unlimited; exec " + " ".join(args), shell=True, cwd=tmp, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
tr = np.loadtxt(os.path.join(tmp, "md_trace.txt"), ndmin=2)
finished = "ref_dump finished OK" in r.stdout
tail = r.stdout[-600:]
print("reference exit code %d, finished its %d steps: %s, trace lines %d" % (r.returncode, n_steps, finished, len(tr)))
print(tail)
out = dict(gen_n_side=np.array(n_side), gen_n_diabat=np.array(n_diabat), gen_seed=np.array(seed), gen_temperature=np.array(temp), gen_xyz=s.xyz, md_velocity0=vel,
           rc=np.array(simkw["rc"]), rv=np.array(simkw["rv"]), nslist=np.array(simkw["nslist"]), pme_grid=np.array(simkw["pme_grid"]),
           trace_step=tr[:, 0].astype(np.int32), trace_epot=tr[:, 1], trace_ekin=tr[:, 2], trace_verlet_flag=tr[:, 3].astype(np.int32), trace_n_diabat=tr[:, 4].astype(np.int32),
           trace_fmax=tr[:, 5], trace_hydronium=tr[:, 6].astype(np.int32), reference_finished=np.array(finished), reference_steps_asked=np.array(n_steps),
           reference_last_output=np.array(tail))
name = "long_evb_n12_liquid_s%d" % seed
np.savez_compressed(os.path.join(ROOT, "tests", "golden", name + ".npz"), **out)
print(name, "fmax first/last", tr[0, 5], tr[-1, 5], "D", sorted(set(tr[:, 4].astype(int))), "hydronium", sorted(set(tr[:, 6].astype(int))))
