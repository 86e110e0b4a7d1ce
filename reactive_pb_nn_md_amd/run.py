"""Command-line mirror of the reference's main program (src/main_ms_evb.f90:16-125) on the MI355X library:

    python -m reactive_pb_nn_md_amd.run  sys.gro  ff.pmt  sys.top  sim.pmt  traj_out.gro  log_out  [hop_out]  [a_out]
                                         [--evb yes|no] [--seed N] [--velocity-checkpoint FILE --checkpoint-every N] [--device D]

Same eight positional files as the reference (src/general_routines.f90:20-27).  Like the reference it refuses to overwrite
its outputs (they are opened status='new', :49-50) and, when trajectory + log + velocity checkpoint all exist, continues the
old trajectory from the checkpoint's last frame.  Differences, deliberately: `--evb` replaces the reference's compile-time
switch ms_evb_simulation (src/glob_v.f90:45; default: yes when the topology has [ evb_pairs ]); the Maxwell-Boltzmann
velocities and the Langevin noise come from seeded generators (the reference seeds from the clock, src/general_routines.f90:726);
ensembles NVE, NVT (Langevin leap-frog) and NPT (Langevin + the Monte-Carlo barostat every `barofreq` steps) run.
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np

from . import api, io, synth


def _mol_atoms(topo):
    return [list(topo["mol_atom_index"][m, :topo["mol_n_atom"][m]] - 1) for m in range(len(topo["mol_n_atom"]))]


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m reactive_pb_nn_md_amd.run", description=__doc__.split("\n\n")[0])
    for name in ("gro", "ffpmt", "top", "simpmt", "traj_out", "log_out"):
        ap.add_argument(name)
    ap.add_argument("hop_out", nargs="?", default=None)
    ap.add_argument("a_out", nargs="?", default=None)
    ap.add_argument("--evb", choices=("yes", "no"), default=None)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--velocity-checkpoint", default="velocity_checkpoint")
    ap.add_argument("--checkpoint-every", type=int, default=0, help="steps between velocity checkpoints (0 = none; the reference: checkpoint_velocity / n_step_velocity)")
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args(argv)

    # restart logic of check_restart_trajectory (src/general_routines.f90:60-140): trajectory, log and velocity checkpoint all present
    # -> continue from the trajectory's last frame (its molecule membership included: protons may have moved) and the checkpoint
    have = [os.path.exists(p) for p in (a.traj_out, a.log_out, a.velocity_checkpoint)]
    restart = all(have)
    if not restart and (have[0] or have[1]):
        raise SystemExit("output file exists (the reference opens its outputs with status='new'): %s" % (a.traj_out if have[0] else a.log_out))
    spec, gro, ff, strings, numbers = io.load_inputs(a.traj_out if restart else a.gro, a.ffpmt, a.top, a.simpmt, frame=-1 if restart else 0, evb=a.evb != "no")
    if strings["ensemble"] not in ("NVE", "NVT", "NPT"):
        raise SystemExit("ensemble %s: NVE, NVT (Langevin) and NPT (Langevin + Monte-Carlo barostat) are built" % strings["ensemble"])
    use_evb = (a.evb == "yes") if a.evb else bool(ff.evb_pairs)
    n_step, n_output, dt = int(numbers["n_step"]), int(numbers["n_output"]), float(numbers["delta_t"])

    n_old = 0
    vel = None
    if restart:
        n_old = int(gro.title.split()[1])
        n_ck, vel = io.read_velocity_checkpoint(a.velocity_checkpoint, spec.n_atom, step=n_old)

    h = api.MsEvbHip(spec, device=a.device)
    baro = None
    if strings["ensemble"] == "NPT":      # mc_sample (src/md_integration.f90:17-60): Langevin steps + a volume move every barofreq steps
        from . import barostat
        if "pressure" not in numbers:
            raise SystemExit("variable pressure has not been given a value AND ensemble is set to NPT")
        baro = barostat.MCBarostat(numbers["pressure"], numbers["temperature"], numbers.get("baroscale", 0.01), a.seed + 1)
    if strings["ensemble"] in ("NVT", "NPT"):      # Langevin leap-frog kicks (src/md_integration.f90:186-228); friction default of :203-204
        for k, v in (("ensemble_nvt", 1), ("friction_coeff", numbers.get("friction_coeff", 0.1)), ("temperature", numbers["temperature"]), ("langevin_seed", a.seed)):
            h.set_option(k, float(v))
    atom_names = list(gro.atom_names)
    mol_type_names = list(spec.params.moltype_names)

    def names_now():
        t = h.topology()
        return t, [mol_type_names[k - 1] for k in t["mol_type_index"]], [spec.params.atype_names[k - 1] for k in t["atom_type_index"]]

    traj = open(a.traj_out, "a" if restart else "x")
    log = open(a.log_out, "a" if restart else "x")
    hop = open(a.hop_out, "a" if restart else "x") if (a.hop_out and use_evb) else None
    ckpt = open(a.velocity_checkpoint, "a") if a.checkpoint_every > 0 else None

    def force_call():
        if use_evb:
            e, _ = h.ms_evb_calculate_total_force_energy(want_force=False)
            return e, None
        en, _ = h.calculate_total_force_energy(want_force=False)
        return en["potential"], (en["elec"], en["vdw"], en["bond"], en["angle"], en["dihedral"])

    def print_step(i_step, e_pot, e_kin, terms):
        t, mnames, anames = names_now()
        io.write_gro_frame(traj, "step  %d time(ps) %r" % (i_step, i_step * dt), mnames, _mol_atoms(t), anames, h.get_positions(), spec.box)
        io.write_log_step(log, i_step, dt, e_pot, e_kin, terms)
        if use_evb:      # print_evb_trajectory_data (src/ms_evb.f90:3128-3162): the diabats of the last force call
            st = h.evb_state()
            log.write(" number of diabat states :  %d\n diabat state    hydronium molecule   evb coefficient  solvation shell\n" % st["D"])
            for d in range(st["D"]):
                hops = [hp for hp in st["proton_log"][d] if hp[0] >= 0]
                mol = hops[-1][3] if hops else t["hydronium_molecule"]
                log.write("%5d%10d%14.6f%5d\n" % (d + 1, mol, st["eigvec"][d] ** 2, len(hops)))
        traj.flush()
        log.flush()

    e_pot, terms = force_call()
    if not restart:
        log.write(" real_space_cutoff %r verlet_cutoff %r alpha_sqrt %r pme_grid %d delta_t %r n_atom %d\n" % (spec.rc, spec.rv, spec.alpha, spec.pme_grid, dt, spec.n_atom))
        frozen = np.asarray(spec.params.atype_freeze)[h.topology()["atom_type_index"] - 1] == 1
        vel = synth.maxwell_boltzmann(spec.mass, float(numbers["temperature"]), a.seed, frozen)
    h.set_velocities(vel)
    e_kin = h.calculate_kinetic_energy()
    if not restart:
        print_step(0, e_pot, e_kin, terms)
    hyd = h.topology()["hydronium_molecule"] if use_evb else 0
    done = 0
    total = n_step - n_old
    block = max(1, n_output)
    if a.checkpoint_every > 0:
        block = np.gcd(block, a.checkpoint_every)
    barofreq = int(numbers.get("barofreq", 25))
    if baro is not None:
        block = np.gcd(block, barofreq)
    while done < total:
        nb = int(min(block, total - done))
        en = h.md_integrate_atomic(nb, dt, use_evb=use_evb)
        done += nb
        step = n_old + done
        def write_hops(at_block_end):
            # every hop of the block with its OWN step, as the reference writes them (src/ms_evb.f90:331-335): the library logs the number
            # of the EVB force call a switch happened in; read BEFORE the barostat makes force calls of its own, the block's MD calls are the
            # last nb of them (a there-and-back pair shows both); a switch inside a barostat force call belongs to the barostat's step
            hops, calls = h.hop_log()
            if hop is not None:
                for call, h_from, h_to in hops:
                    hop.write(" step  %d\n proton hop from  %d  to  %d\n" % (step - (calls - int(call)) if at_block_end else step, h_from, h_to))
                hop.flush()
        if use_evb:
            write_hops(True)
        if baro is not None and step % barofreq == 0:
            t = h.topology()
            acc, e_now = baro.move(h, en[-1, 0], use_evb, _mol_atoms(t), spec.mass, spec.n_mole)
            en[-1, 0] = e_now
            spec = h.spec
            if use_evb:
                write_hops(False)
        if use_evb:
            hyd = h.topology()["hydronium_molecule"]
        if done % max(1, n_output) == 0:
            t5 = None
            if not use_evb:
                e5 = h.calculate_total_force_energy(want_force=False)[0]
                t5 = (e5["elec"], e5["vdw"], e5["bond"], e5["angle"], e5["dihedral"])
            print_step(step, en[-1, 0], h.calculate_kinetic_energy(), t5)
        if ckpt is not None and done % a.checkpoint_every == 0:
            t, mnames, anames = names_now()
            io.write_velocity_checkpoint(ckpt, step, mnames, _mol_atoms(t), anames, h.get_velocities())
            ckpt.flush()
    for f in (traj, log, hop, ckpt):
        if f is not None:
            f.close()
    h.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
