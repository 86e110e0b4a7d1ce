"""Monte-Carlo barostat of the reference (monte_carlo_barostat + scale_coordinates, src/md_integration.f90:229-421), on the host:
a trial change of the cubic box length with every molecule's centre of mass scaled along, one force call in the new box, Metropolis
acceptance with the reference's weight  w = E_new - E_old + p (V_new - V_old) - n_mole kT 3 ln(L_new / L_old),  forced acceptance when the
trial call moved the excess proton (so that proton jumps are not undone), step size adapted every > 10 trials to keep the acceptance between
25 % and 75 %.  The random numbers come from a seeded generator (the reference seeds from the clock, src/general_routines.f90:726).
Lifted limit: the reference keeps its saved coordinates in 10 000-atom stack arrays (:245-246)."""
from __future__ import annotations

import numpy as np

BOLTZMANN = 0.008314462          # src/glob_v.f90:390
BAR_TO_KJMOL_A3 = 6.022 / 10 ** 5      # `conv` of the reference (:253)


class MCBarostat:
    def __init__(self, pressure_bar, temperature, baroscale=0.01, seed=0):
        self.pressure, self.temperature, self.baroscale = float(pressure_bar), float(temperature), float(baroscale)
        self.rng = np.random.default_rng(seed)
        self.n_trials = self.n_accept = 0
        self.last = None
        self.uniforms = None      # optional iterator of uniform deviates used INSTEAD of the generator (replaying a reference run's draws)

    def _uniform(self):
        return float(next(self.uniforms)) if self.uniforms is not None else self.rng.random()

    @staticmethod
    def scaled_positions(xyz, mass, mol_atoms, scale):
        """scale_coordinates (:388-421): r_com *= scale (cubic box), atoms keep their displacement from the centre of mass"""
        out = xyz.copy()
        for atoms in mol_atoms:
            a = np.asarray(atoms)
            com = (xyz[a] * mass[a, None]).sum(0) / mass[a].sum()
            out[a] = xyz[a] + com * (scale - 1.0)
        return out

    def move(self, h, e_old, use_evb, mol_atoms, mass, n_mole):
        """one trial move on handle h (api.MsEvbHip) whose last force call gave the potential energy e_old; returns (accepted, e_now)"""
        box = np.asarray(h.spec.box, dtype=np.float64)
        if np.abs(box - box[0]).max() >= 0.001:
            raise ValueError("monte carlo barostat cannot be used with non-cubic box")
        self.n_trials += 1
        kT = BOLTZMANN * self.temperature
        old_len, v_old = box[0], box[0] ** 3
        x_old = h.get_positions()
        hyd_old = h.topology()["hydronium_molecule"] if use_evb else 0
        new_len = old_len + old_len * self.baroscale * (self._uniform() * 2.0 - 1.0)
        h.set_box(np.full(3, new_len))
        h.set_positions(self.scaled_positions(x_old, mass, mol_atoms, new_len / old_len))
        e_new = self._energy(h, use_evb)
        w = e_new - e_old + BAR_TO_KJMOL_A3 * self.pressure * (new_len ** 3 - v_old) - n_mole * kT * 3.0 * np.log(new_len / old_len)
        if use_evb and h.topology()["hydronium_molecule"] != hyd_old:
            w = -1.0
        accepted = True
        if w >= 0.0 and self._uniform() > np.exp(-w / kT):
            accepted = False
            h.set_box(np.full(3, old_len))
            h.set_positions(x_old)
            e_new = self._energy(h, use_evb)      # the reference restores its saved energies and forces; the same values, recomputed
        if accepted:
            self.n_accept += 1
        self.last = dict(w=w, new_len=new_len, old_len=old_len, accepted=accepted)
        if self.n_trials > 10:
            if self.n_accept < 0.25 * self.n_trials:
                self.baroscale /= 1.1
                self.n_trials = self.n_accept = 0
            elif self.n_accept > 0.75 * self.n_trials:
                self.baroscale *= 1.1
                self.n_trials = self.n_accept = 0
        return accepted, e_new

    @staticmethod
    def _energy(h, use_evb):
        if use_evb:
            return h.ms_evb_calculate_total_force_energy(want_force=False)[0]
        return h.calculate_total_force_energy(want_force=False)[0]["potential"]
