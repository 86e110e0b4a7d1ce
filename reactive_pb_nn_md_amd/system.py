"""SystemSpec: the static description of one simulation (0-based indices, numpy arrays) that the
host side hands to the C-ABI library - and, in tests, to the CPU oracle.

Two constructors:
* ``from_synth``      - synthetic box (synth.make_system) + force field tables (forcefield.build_params)
* ``from_golden``     - a tests/golden/*.npz fixture, i.e. the reference's OWN tables as dumped by
                        oracle/ref_driver.f90 (1-based Fortran arrays converted here)
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from . import forcefield as ffmod


@dataclass
class SystemSpec:
    box: np.ndarray
    rc: float
    rv: float
    alpha: float
    pme_grid: int
    nslist: int
    xyz: np.ndarray
    atype: np.ndarray
    charge: np.ndarray
    mass: np.ndarray
    mol_first: np.ndarray
    mol_natom: np.ndarray
    mol_type: np.ndarray
    params: ffmod.Params
    hydronium: int = -1
    max_chain: int = 3
    delta_t: float = 0.0005
    extra: dict = field(default_factory=dict)

    @property
    def n_atom(self):
        return len(self.atype)

    @property
    def n_mole(self):
        return len(self.mol_first)

    # flat views used by both ctypes wrappers ---------------------------------------------
    def angle_tables(self):
        keys = sorted(self.params.angle_type)
        idx = np.array([[k[0], k[1], k[2], self.params.angle_type[k][0]] for k in keys], dtype=np.int32).reshape(-1, 4)
        par = np.array([[self.params.angle_type[k][1], self.params.angle_type[k][2]] for k in keys], dtype=np.float64).reshape(-1, 2)
        return idx, par

    def dihedral_tables(self):
        keys = sorted(self.params.dihedral_type)
        idx = np.array([[k[0], k[1], k[2], k[3], self.params.dihedral_type[k][0]] for k in keys], dtype=np.int32).reshape(-1, 5)
        par = np.array([self.params.dihedral_type[k][1] for k in keys], dtype=np.float64).reshape(-1, 6)
        return idx, par


def from_synth(system, params: ffmod.Params, *, rc=9.0, rv=10.5, alpha=0.3, pme_grid=32, nslist=None, max_chain=3):
    idx = {n: i for i, n in enumerate(params.atype_names)}
    midx = {n: i for i, n in enumerate(params.moltype_names)}
    atype = np.array([idx[a] for a in system.atom_names], dtype=np.int32)
    L = float(system.box[0])
    if nslist is None:
        nslist = max(10, min(99, int(L / rv * 2.5)))
    return SystemSpec(
        box=np.asarray(system.box, dtype=np.float64), rc=rc, rv=rv, alpha=alpha, pme_grid=pme_grid, nslist=nslist,
        xyz=np.ascontiguousarray(system.xyz, dtype=np.float64), atype=atype,
        charge=params.atype_chg[atype].copy(), mass=params.atype_mass[atype].copy(),
        mol_first=system.mol_first.astype(np.int32), mol_natom=system.mol_natom.astype(np.int32),
        mol_type=np.array([midx[m] for m in system.mol_names], dtype=np.int32),
        params=params, hydronium=int(system.hydronium), max_chain=max_chain)


def _f(a):
    """dumped Fortran array (read back with reversed dims) -> natural [i,j,k] indexing"""
    return np.ascontiguousarray(np.transpose(a))


def from_golden(g) -> SystemSpec:
    nt = int(g["n_atom_type"][0])
    nmt = int(g["n_molecule_type"][0])
    vdw_type = _f(g["atype_vdw_type"])[:nt, :nt].astype(np.int32)
    vdw = _f(g["atype_vdw_parameter"])[:nt, :nt, :]
    vdw14 = _f(g["atype_vdw_parameter_14"])[:nt, :nt, :]
    btype = _f(g["atype_bond_type"])[:nt, :nt].astype(np.int32)
    bpar = _f(g["atype_bond_parameter"])[:nt, :nt, :]
    ang = {}
    ai, ap = g["angle_types_idx"], g["angle_types_par"]
    for r in range(ai.shape[0]):
        i, j, k, t = (int(x) for x in ai[r])
        ang[(i - 1, j - 1, k - 1)] = (t, float(ap[r, 0]), float(ap[r, 1]))
    dih = {}
    di, dp = g["dihedral_types_idx"], g["dihedral_types_par"]
    for r in range(di.shape[0]):
        i, j, k, l, t = (int(x) for x in di[r])
        dih[(i - 1, j - 1, k - 1, l - 1)] = (t, np.array(dp[r], dtype=np.float64))
    mts = []
    for it in range(1, nmt + 1):
        p = "mt%d_" % it
        na = int(g[p + "n_atom"][0])

        def lst(name, w):
            a = g[p + name]
            return (a.reshape(-1, w).astype(np.int32) - 1) if a.size and a.shape[0] > 0 and a.size >= w else np.zeros((0, w), np.int32)
        mts.append(dict(name="mt%d" % it, n_atom=na, atype=g[p + "atom_type_index"].astype(np.int32) - 1,
                        bonds=lst("bonds", 2), angles=lst("angles", 3), dihedrals=lst("dihedrals", 4),
                        excl=_f(g[p + "pair_exclusions"]).astype(np.int32),
                        reactive_protons=(g[p + "reactive_protons"].astype(np.int32) if p + "reactive_protons" in g else np.zeros(na, np.int32)),
                        reactive_basic=(g[p + "reactive_basic"].astype(np.int32) if p + "reactive_basic" in g else np.zeros(na, np.int32))))
    evb = {}
    hyd = -1
    if "evb_conjugate_pairs" in g:
        def rows(tname, pname, nty, npar):
            tt = _f(g[tname])
            pp = _f(g[pname])
            out = []
            for r in range(tt.shape[0]):
                if tt[r, 0] == 0:
                    break
                out.append((tuple(int(x) - 1 for x in tt[r, :nty]), np.array(pp[r, :npar], dtype=np.float64)))
            return out
        cp = rows("evb_diabat_coupling_interaction", "evb_diabat_coupling_parameters", 3, 10)
        cpt = g["evb_diabat_coupling_type"]
        evb = dict(
            acid=g["evb_acid_molecule"][:nmt].astype(np.int32), base=g["evb_basic_molecule"][:nmt].astype(np.int32),
            conjugate_mol=g["evb_conjugate_pairs"][:nmt].astype(np.int32) - 1,
            proton_type=g["evb_proton_index"][:nmt].astype(np.int32) - 1,
            heavy_type=g["evb_heavy_acid_index"][:nmt].astype(np.int32) - 1,
            conjugate_atype=g["evb_conjugate_atom_index"][:nt].astype(np.int32) - 1,
            reference_energy=g["evb_reference_energy"][:nmt].astype(np.float64),
            exch_atomic=g["evb_exchange_charge_atomic"][:nt].astype(np.float64),
            exch_proton=_f(g["evb_exchange_charge_proton"])[:nmt, :nmt].copy(),
            donor_acceptor=rows("evb_donor_acceptor_interaction", "evb_donor_acceptor_parameters", 3, 6),
            proton_acceptor=rows("evb_proton_acceptor_interaction", "evb_proton_acceptor_parameters", 2, 5),
            coupling=[(c[0], int(cpt[i]), c[1]) for i, c in enumerate(cp)],
        )
        hyd = int(g["hydronium_molecule"][0]) - 1
    params = ffmod.Params(atype_names=["t%d" % i for i in range(nt)], atype_chg=g["atype_chg"][:nt].astype(np.float64),
                          atype_mass=g["atype_mass"][:nt].astype(np.float64), atype_freeze=g["atype_freeze"][:nt].astype(np.int32), vdw_type=vdw_type, vdw_param=vdw,
                          vdw_param14=vdw14, bond_type=btype, bond_param=bpar, angle_type=ang, dihedral_type=dih,
                          moltype_names=[m["name"] for m in mts], moltypes=mts, evb=evb)
    box = _f(g["box"])
    return SystemSpec(
        box=np.array([box[0, 0], box[1, 1], box[2, 2]]), rc=float(g["real_space_cutoff"][0]), rv=float(g["verlet_cutoff"][0]),
        alpha=float(g["alpha_sqrt"][0]), pme_grid=int(g["pme_grid"][0]), nslist=int(g["nslist"][0]),
        xyz=np.ascontiguousarray(g["xyz"], dtype=np.float64), atype=g["atom_type_index"].astype(np.int32) - 1,
        charge=g["charge"].astype(np.float64), mass=g["mass"].astype(np.float64),
        mol_first=g["mol_first_atom"].astype(np.int32) - 1, mol_natom=g["mol_n_atom"].astype(np.int32),
        mol_type=g["mol_type_index"].astype(np.int32) - 1, params=params, hydronium=hyd,
        max_chain=int(g["evb_max_chain"][0]) if "evb_max_chain" in g else 3, delta_t=float(g["delta_t"][0]))
