"""Synthetic CH3SO3H / H2O boxes: lattice water + one methanesulfonic
acid + a planted hydrogen-bond tree of waters that fixes the number of EVB diabats.

The reference ships no coordinate file (example_input_files/ has only .pmt/.top), so every
configuration of BASELINE.json is generated here, deterministically from a seed.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .forcefield import BOLTZMANN, CONV_KJMOL_ANG2PS2GMOL

SPACING = 3.1  # Angstrom, simple-cubic lattice -> 0.1007 atoms / A^3

_WATER = np.array([[0.0, 0.0, 0.0], [0.816, 0.577, 0.0], [-0.816, 0.577, 0.0]])
# C_a, S_a, O_a, O_a, O_ah, H_a  (acidic proton last, reference: src/ms_evb.f90:144-165)
_ACID = np.array([
    [0.000, 0.000, 1.810],
    [0.000, 0.000, 0.000],
    [1.367, 0.000, -0.483],
    [-0.684, 1.184, -0.483],
    [-0.764, -1.323, -0.540],
    [-0.310, -2.170, -0.420],
])


@dataclass
class System:
    box: np.ndarray          # [3] orthorhombic box lengths (Angstrom)
    xyz: np.ndarray          # [N,3]
    atom_names: list         # atom-type name per atom
    mol_names: list          # molecule-type name per molecule
    mol_first: np.ndarray    # [M] first atom (0-based) of each molecule
    mol_natom: np.ndarray    # [M]
    hydronium: int           # 0-based molecule index of the acid
    n_diabat_planted: int

    @property
    def n_atom(self):
        return len(self.xyz)

    @property
    def n_mole(self):
        return len(self.mol_first)


def _unit(v):
    return v / np.linalg.norm(v)


def _plant_tree(acid_xyz, n_water, max_chain, bifurcate=()):
    """Waters H-bonded to the acid proton, breadth first: O placed 1.6 A beyond each donating
    H along the donor O-H axis (dist(H,O_acc)=1.6 < evb_reactive_pair_distance=2.5,
    reference: src/glob_v.f90:55, src/ms_evb.f90:747).
    bifurcate: numbers (breadth-first, 0 = the acid proton) of the donating hydrogens that get TWO acceptors, 2.2 A away at +-50 degrees
    off the O-H axis (both inside the 2.5 A search radius, 3.4 A apart): more than 2^k diabats per hop depth without a longer chain -
    SURVEY.md 8d's way to 16 diabats with the reference's compile-time evb_max_chain = 3."""
    waters = []
    if n_water == 0:
        return waters
    # queue entries: (donor heavy atom pos, donor H pos, perpendicular hint, depth)
    o_ah, h_a = acid_xyz[4], acid_xyz[5]
    a0 = _unit(h_a - o_ah)
    p0 = _unit(np.cross(a0, np.array([0.3, 0.5, 0.81])))
    queue = [(o_ah, h_a, p0, 1)]
    c, s = np.cos(np.deg2rad(54.735)), np.sin(np.deg2rad(54.735))
    link = 0
    while queue and len(waters) < n_water:
        o_d, h_d, p, depth = queue.pop(0)
        a = _unit(h_d - o_d)
        p = _unit(p - np.dot(p, a) * a)
        if link in bifurcate:
            cb, sb = np.cos(np.deg2rad(50.0)), np.sin(np.deg2rad(50.0))
            places = [(h_d + 2.2 * (cb * a + sb * p), _unit(cb * a + sb * p)), (h_d + 2.2 * (cb * a - sb * p), _unit(cb * a - sb * p))]
        else:
            places = [(h_d + 1.6 * a, a)]
        link += 1
        for o_new, ax in places:
            if len(waters) >= n_water:
                break
            q = _unit(np.cross(a, p))
            pp = p if len(places) == 1 else q      # the hydrogens of a bifurcated pair open out of the plane the two acceptors lie in
            pp = _unit(pp - np.dot(pp, ax) * ax)
            h1 = o_new + 1.0 * (c * ax + s * pp)
            h2 = o_new + 1.0 * (c * ax - s * pp)
            waters.append(np.stack([o_new, h1, h2]))
            if depth < max_chain:
                qq = _unit(np.cross(ax, pp))
                queue.append((o_new, h1, qq, depth + 1))
                queue.append((o_new, h2, qq, depth + 1))
    return waters


def _random_rotations(rng, n):
    """n uniformly distributed rotation matrices (normalised Gaussian quaternions)"""
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1)[:, None]
    w, x, y, z = q.T
    return np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], -1),
                     np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], -1),
                     np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1)], 1)


def make_system(n_side: int, n_diabat: int = 1, seed: int = 0, max_chain: int = 3,
                acid_position: str = "middle", quantize_gro: bool = True, liquid_like: bool = False,
                proton_shift: float = 0.0, bifurcate=()) -> System:
    """n_side^3 lattice sites (box = n_side*3.1 A); n_diabat = 1 + number of planted waters
    (1, 2, 4 -> partial tree, 8 -> full depth-3 binary tree, 16 needs max_chain=4)."""
    rng = np.random.default_rng(seed)
    L = n_side * SPACING
    centre = np.full(3, 0.5 * L)
    acid = _ACID + centre - _ACID[1]
    tree = _plant_tree(acid, n_diabat - 1, max_chain, tuple(bifurcate))
    if proton_shift:      # acid proton moved along O-H towards its planted acceptor: > ~0.3 A makes the acceptor's diabat the principal one
        acid = acid.copy()
        acid[5] += proton_shift * _unit(acid[5] - acid[4])
    special = np.concatenate([acid] + tree) if tree else acid

    g = (np.arange(n_side) + 0.5) * SPACING
    sites = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3)
    keep = np.ones(len(sites), dtype=bool)
    # lattice sites within 4.6 A of the acid (any atom within 3.4 A, S within 4.6 A) removed
    d = sites - acid[1]
    d -= L * np.round(d / L)
    keep &= (d ** 2).sum(1) > 4.6 ** 2
    for at in acid:
        d = sites - at
        d -= L * np.round(d / L)
        keep &= (d ** 2).sum(1) > 3.4 ** 2
    # ... and within 3.3 A of any planted atom (keeps stray acceptors > 2.5 A from chain protons
    # even after jitter and water-H offsets)
    for w in tree:
        for at in w:
            d = sites - at
            d -= L * np.round(d / L)
            keep &= (d ** 2).sum(1) > 3.3 ** 2
    sites = sites[keep]
    nl = len(sites)
    if liquid_like:
        # rigid molecules at the force field's equilibrium geometry (r0 = 1.012 A, 113.24 deg), random orientation, rigid jitter:
        # no intramolecular strain and no systematic H...H contacts (the default per-atom jitter starts ~60 kJ/mol per molecule hot)
        half = np.deg2rad(113.24) / 2.0
        tmpl = np.array([[0.0, 0.0, 0.0], [1.012 * np.sin(half), 1.012 * np.cos(half), 0.0], [-1.012 * np.sin(half), 1.012 * np.cos(half), 0.0]])
        R = _random_rotations(rng, nl)
        lat = sites[:, None, :] + np.einsum("mij,aj->mai", R, tmpl) + rng.uniform(-0.15, 0.15, size=(nl, 1, 3))
    else:
        lat = sites[:, None, :] + _WATER[None, :, :] + rng.uniform(-0.15, 0.15, size=(nl, 3, 3))

    # molecule order: lattice waters with the acid (and alternating planted waters) in the middle
    mols = []  # (name, coords)
    if acid_position == "first":
        n_before = 0
    else:
        n_before = nl // 2
    for i in range(n_before):
        mols.append(("h2o", lat[i]))
    for k, w in enumerate(tree):
        if k % 2 == 0:
            mols.append(("h2o", w))
    hyd = len(mols)
    mols.append(("so3h", acid))
    for k, w in enumerate(tree):
        if k % 2 == 1:
            mols.append(("h2o", w))
    for i in range(n_before, nl):
        mols.append(("h2o", lat[i]))

    xyz = np.concatenate([m[1] for m in mols])
    if quantize_gro:
        # what the reference reads back from a .gro (F8.3 nm, *10: src/general_routines.f90:241,264)
        xyz = np.array([float("%.3f" % v) for v in (xyz / 10.0).ravel()]).reshape(-1, 3) * 10.0
    names = []
    for m in mols:
        names += ["OW", "HW", "HW"] if m[0] == "h2o" else ["C_a", "S_a", "O_a", "O_a", "O_ah", "H_a"]
    natom = np.array([len(m[1]) for m in mols], dtype=np.int32)
    first = np.concatenate([[0], np.cumsum(natom)[:-1]]).astype(np.int32)
    return System(box=np.full(3, L), xyz=xyz, atom_names=names, mol_names=[m[0] for m in mols],
                  mol_first=first, mol_natom=natom, hydronium=hyd, n_diabat_planted=n_diabat)


def maxwell_boltzmann(mass: np.ndarray, temperature: float, seed: int, frozen: np.ndarray | None = None) -> np.ndarray:
    """Seeded velocities (A/ps), zero total momentum, rescaled to the target temperature
    (same post-processing as src/md_integration.f90:100-114; the reference seeds its own
    generator from the clock, src/general_routines.f90:726, so the draw itself is ours).
    `frozen` (bool per atom, from atype_freeze): those atoms get no velocity and are left out of the momentum
    removal and of the temperature count (src/md_integration.f90:89, :107, :151, :169)."""
    rng = np.random.default_rng(seed)
    n = len(mass)
    free = np.ones(n, dtype=bool) if frozen is None else ~np.asarray(frozen, dtype=bool)
    nf = int(free.sum())
    sd = np.sqrt(BOLTZMANN * 1000.0 * temperature / (mass / 1000.0)) / 100.0
    v = rng.standard_normal((n, 3)) * sd[:, None]
    v[~free] = 0.0
    p = (mass[:, None] * v).sum(0) / nf
    v[free] -= p[None, :] / mass[free, None]
    ke = 0.5 * (mass[:, None] * v * v).sum() / CONV_KJMOL_ANG2PS2GMOL
    v *= np.sqrt(1.5 * BOLTZMANN * temperature * nf / ke)
    return v
