"""Force-field description and host-side parameter-table construction.

Host-side mirror of the reference's *initialisation* of the tables the hot path
reads (cold path, SURVEY.md section 8 "init-time, feeds hot path"):

* atom-type table + LJ/SAPT classification and combination rules
  (reference: src/initialize_routines.f90:281-430 ``read_param``, :448-558
  ``gen_param``, :569-634 combination rules, :646-691 1-4 ``pairtypes``),
* bonded parameter tensors (src/intra_bonded_interactions.f90:785-1054),
* intra-molecular exclusion matrices (src/intra_bonded_interactions.f90:574-660),
* MS-EVB parameter tables (src/ms_evb.f90:3170-3623, src/glob_v.f90:54-120).

Everything is expressed with 0-based type indices; the numeric *values* are the
ones shipped in the reference's ``example_input_files`` (data, not code).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

# truncated/rounded constants of the reference (src/glob_v.f90:386-390)
REF_PI = 3.141592654
REF_PI_SQRT = 1.772453851
CONV_E2A_KJMOL = float(np.float32(1389.35465))  # fp32 literal -> 1389.3546142578125
CONV_KJMOL_ANG2PS2GMOL = 100.0
BOLTZMANN = 0.008314462

MAX_VDW_PAR = 6
EXP_INIT = 3.0      # src/initialize_routines.f90:289
SMALL = 1e-6


@dataclass
class MoleculeType:
    name: str
    atoms: list          # [(atype_name, mass)]
    bonds: list          # [(i,j)] 1-based local indices as in the .top format
    angles: list         # [(i,j,k)]
    dihedrals: list      # [(i,j,k,l)]
    exclusions: list = field(default_factory=list)


@dataclass
class EvbPair:
    acid: str
    base: str
    proton_type: str
    heavy_type: str
    acid_reactive_protons: list
    base_reactive_protons: list
    acid_acceptor_atoms: list
    base_acceptor_atoms: list
    conjugate_atoms: list    # [(acid_atype, base_atype)] in file order (later lines overwrite)


@dataclass
class ForceField:
    atypes: list                      # [(name, charge, epsilon, sigma)]
    cross_terms: list                 # [(i,j,C6,C12)] 1-based atom-type indices (opls)
    pairtypes: list                   # [(name1,name2,C6,C12)]
    bondtypes: list                   # [(t1,t2,func,p...)]
    angletypes: list                  # [(t1,t2,t3,func,th0_deg,cth)]
    dihedraltypes: list               # [(t1,t2,t3,t4,func,p...)]
    moltypes: list                    # [MoleculeType] in .top order
    reference_energy: dict
    donor_acceptor: list              # [(acc_heavy, don_heavy, H, B,b,d0,b',rs,rc)]
    proton_acceptor: list             # [(acc_heavy, H, C,c,d0,rs,rc)]
    geometry_factor: list             # [(t1,t2,t3,functype,[params])]
    exchange_charge_atomic: dict
    exchange_charge_proton: list      # [(mol1,mol2,q)]
    evb_pairs: list                   # [EvbPair]
    freeze: list = field(default_factory=list)  # names of frozen atom types (5th column of the force-field file = 1, src/md_integration.f90:473)
    sapt: list = field(default_factory=list)   # optional [(name, A_ex, A_el, A_ind, A_dhf, B, C6, C8, C10, C12)] 'custom_sapt_parameters'


def with_standard_cross_terms(ff: ForceField) -> ForceField:
    """the force field for `lj_comb_rule standard` runs: the explicit cross terms become (epsilon, sigma) pairs (test values); the atomic
    parameters already are epsilon / sigma"""
    import dataclasses
    return dataclasses.replace(ff, cross_terms=[(i, j, 0.55 + 0.1 * k, 2.9 + 0.2 * k) for k, (i, j, _a, _b) in enumerate(ff.cross_terms)])


def with_rb_dihedral(ff: ForceField) -> ForceField:
    """the acid's proper dihedral C-S-O-H as a Ryckaert-Bellemans polynomial (function type 3, test coefficients): exercises the third
    dihedral form of quartet_dihedral_energy_force (src/intra_bonded_interactions.f90:529)"""
    import dataclasses
    dt = [(("C_a", "S_a", "O_ah", "H_a", 3, 9.28, 12.16, -13.12, -3.06, 26.24, -31.5) if d[:4] == ("C_a", "S_a", "O_ah", "H_a") else d) for d in ff.dihedraltypes]
    return dataclasses.replace(ff, dihedraltypes=dt)


def with_gaussian_coupling(ff: ForceField) -> ForceField:
    """the acid-water coupling with the reference's second geometric form, A = exp(-gamma q^2) exp(-k (R_OO - D)^2)
    (evb_diabatic_coupling_function case 2, src/ms_evb.f90:1241-1250; test parameters {Vconst, gamma, k, D})"""
    import dataclasses
    gf = [((r[0], r[1], r[2], 2, [-92.5, 1.75, 9.1, 2.55]) if r[:3] == ("OW", "O_ah", "H_a") else r) for r in ff.geometry_factor]
    return dataclasses.replace(ff, geometry_factor=gf)


def with_test_sapt(ff: ForceField) -> ForceField:
    """adds a synthetic 'custom_sapt_parameters' block (test values, not a fitted force field) so that the SAPT/Tang-Toennies
    branch of the pair kernel is exercised with non-zero coefficients: every pair involving a zero-epsilon (hydrogen) type
    becomes a live Buckingham + damped-dispersion interaction."""
    rows = []
    for (name, q, eps, sig) in ff.atypes:
        if eps == 0.0:
            rows.append((name, 2000.0, 50.0, 10.0, 5.0, 3.5, 50.0, 100.0, 20.0, 5.0))
        else:
            rows.append((name, 100000.0, 500.0, 100.0, 50.0, 3.6, 1500.0, 8000.0, 40000.0, 100000.0))
    ff.sapt = rows
    return ff


def ch3so3h_water_forcefield() -> ForceField:
    """Parameter VALUES of the reference's example_input_files/CH3SO3H.pmt and
    CH3SO3H_H2O.top (MS-EVB3 water + methanesulfonic acid)."""
    atypes = [
        ("C_a", 0.0, 0.86715, 3.74792), ("S_a", 1.2, 1.905868, 3.307694),
        ("O_a", -0.53, 5.024532, 2.62585), ("O_ah", -0.53, 5.024532, 2.62585),
        ("H_a", 0.39, 0.0, 0.0), ("C_b", -0.47, 0.86715, 3.74792),
        ("S_b", 1.54, 1.905868, 3.307694), ("O_b", -0.69, 5.024532, 2.62585),
        ("OW", -0.82, 0.6502995, 3.16549), ("HW", 0.41, 0.0, 0.0),
        ("O_h3o", -0.5, 0.0, 0.0), ("H_h3o", 0.5, 0.0, 0.0),
    ]
    cross_terms = [(11, 9, 1993.468, 1917990.0), (12, 9, 0.660773, 10.38771)]
    pairtypes = [("H_a", "O_a", 434.1, 66466.2), ("H_a", "C_a", 663.7, 373998.7)]
    bondtypes = [
        ("S_b", "O_b", 1, 1.48, 4201.0), ("C_b", "S_b", 2, 1.81, 562.0),
        ("S_a", "O_a", 1, 1.45, 5430.0), ("S_a", "O_ah", 1, 1.62, 3675.7),
        ("C_a", "S_a", 2, 1.81, 562.0), ("O_ah", "H_a", 3, 372.2086, 2.1, 0.97),
        ("OW", "HW", 1, 1.012, 4431.5338), ("O_h3o", "H_h3o", 3, 372.2086, 2.1, 1.0),
    ]
    angletypes = [
        ("C_b", "S_b", "O_b", 2, 103.0, 520.0), ("O_b", "S_b", "O_b", 2, 114.0, 780.0),
        ("C_a", "S_a", "O_a", 2, 110.0, 520.0), ("C_a", "S_a", "O_ah", 2, 100.0, 520.0),
        ("O_a", "S_a", "O_a", 2, 122.0, 780.0), ("O_a", "S_a", "O_ah", 2, 107.0, 780.0),
        ("S_a", "O_ah", "H_a", 1, 106.0, 378.0), ("HW", "OW", "HW", 1, 113.24, 317.5656),
        ("H_h3o", "O_h3o", "H_h3o", 1, 111.7269, 324.20477),
    ]
    dihedraltypes = [
        ("C_a", "S_a", "O_ah", "H_a", 1, 180.0, 2.92, 3.0),
        ("S_b", "C_b", "O_b", "O_b", 2, 35.26439, 334.84617),
        ("S_b", "O_b", "O_b", "O_b", 2, 35.26439, 334.84617),
        ("S_a", "C_a", "O_a", "O_a", 2, 35.26439, 334.84617),
        ("S_a", "C_a", "O_a", "O_ah", 2, 35.26439, 334.84617),
        ("S_a", "O_a", "O_a", "O_ah", 2, 35.26439, 334.84617),
        ("S_a", "O_ah", "O_a", "O_a", 2, 35.26439, 334.84617),
        ("S_a", "O_a", "O_ah", "O_a", 2, 35.26439, 334.84617),
    ]
    moltypes = [
        MoleculeType("so3h", [("C_a", 15.035), ("S_a", 32.06), ("O_a", 16.0), ("O_a", 16.0), ("O_ah", 16.0), ("H_a", 1.008)],
                     [(1, 2), (2, 3), (2, 4), (2, 5), (5, 6)],
                     [(1, 2, 3), (1, 2, 4), (1, 2, 5), (2, 5, 6)],
                     [(1, 2, 5, 6), (2, 3, 4, 5), (2, 4, 5, 3), (2, 5, 3, 4)]),
        MoleculeType("so3", [("C_b", 15.035), ("S_b", 32.06), ("O_b", 16.0), ("O_b", 16.0), ("O_b", 16.0)],
                     [(1, 2), (2, 3), (2, 4), (2, 5)],
                     [(1, 2, 3), (1, 2, 4), (1, 2, 5)],
                     [(2, 3, 4, 5), (2, 4, 5, 3), (2, 5, 3, 4)]),
        MoleculeType("h2o", [("OW", 16.0), ("HW", 1.008), ("HW", 1.008)],
                     [(1, 2), (1, 3)], [(2, 1, 3)], []),
        MoleculeType("h3o", [("O_h3o", 16.0), ("H_h3o", 1.008), ("H_h3o", 1.008), ("H_h3o", 1.008)],
                     [(1, 2), (1, 3), (1, 4)], [(2, 1, 3), (2, 1, 4), (3, 1, 4)], []),
    ]
    gf = [-97.0151921, 1.8302895, 0.232726, 9.562153, 2.94, 6.0179066, 3.1, 10.8831327, 10.0380922, 1.8136426]
    return ForceField(
        atypes=atypes, cross_terms=cross_terms, pairtypes=pairtypes, bondtypes=bondtypes,
        angletypes=angletypes, dihedraltypes=dihedraltypes, moltypes=moltypes,
        reference_energy={"so3h": -643.65, "h3o": 0.0},
        donor_acceptor=[("OW", "O_h3o", "H_h3o", 47.1118977, 1.1, 2.4, 2.12, 2.85, 3.05),
                        ("OW", "O_ah", "H_a", 47.1118977, 1.1, 2.4, 2.12, 2.85, 3.05),
                        ("O_b", "O_h3o", "H_h3o", 47.1118977, 1.1, 2.4, 2.12, 2.85, 3.05)],
        proton_acceptor=[("OW", "H_h3o", 19.1274639, 2.1, 1.0, 2.3, 2.5),
                         ("O_b", "H_h3o", 19.1274639, 2.1, 1.0, 2.3, 2.5),
                         ("OW", "H_a", 19.1274639, 2.1, 1.0, 2.3, 2.5)],
        geometry_factor=[("OW", "O_h3o", "H_h3o", 1, gf), ("O_b", "O_h3o", "H_h3o", 1, gf), ("OW", "O_ah", "H_a", 1, gf)],
        exchange_charge_atomic={"OW": -0.0895456, "O_h3o": -0.0895456, "HW": 0.0252683, "H_h3o": 0.0252683,
                                "C_b": 0.0, "C_a": 0.0, "S_b": 0.229278, "S_a": 0.229278,
                                "O_b": -0.0895456, "O_a": -0.0895456, "O_ah": -0.0895456},
        exchange_charge_proton=[("h3o", "h2o", 0.078018), ("h3o", "so3", 0.078018), ("so3h", "h2o", 0.078018)],
        evb_pairs=[
            EvbPair("so3h", "so3", "H_a", "O_ah", [0, 0, 0, 0, 0, 1], [0, 0, 0, 0, 0], [0, 0, 1, 1, 1, 0], [0, 0, 1, 1, 1],
                    [("C_a", "C_b"), ("S_a", "S_b"), ("O_ah", "O_b"), ("O_a", "O_b")]),
            EvbPair("h3o", "h2o", "H_h3o", "O_h3o", [0, 1, 1, 1], [0, 1, 1], [1, 0, 0, 0], [1, 0, 0],
                    [("O_h3o", "OW"), ("H_h3o", "HW")]),
        ],
    )


# ----------------------------------------------------------------------------------------------
@dataclass
class Params:
    """Processed tables, the host-side input of the C-ABI (include/msevb_hip.h, msevb_upload_*)."""
    atype_names: list
    atype_chg: np.ndarray            # [nt]
    atype_mass: np.ndarray           # [nt]
    vdw_type: np.ndarray             # [nt,nt] int32: 0 LJ, 1 SAPT, -1 none
    vdw_param: np.ndarray            # [nt,nt,6]  LJ: (C12,C6,...)  SAPT: (A,B,C6,C8,C10,C12)
    vdw_param14: np.ndarray          # [nt,nt,6]
    bond_type: np.ndarray            # [nt,nt] int32
    bond_param: np.ndarray           # [nt,nt,3]
    angle_type: dict                 # {(i,j,k): (type, th0_rad, cth)}
    dihedral_type: dict              # {(i,j,k,l): (type, p[6])}
    moltype_names: list
    moltypes: list                   # per molecule type: dict(n_atom, atype[], bonds[nb,2], angles[na,3], dihedrals[nd,4], excl[n,n])
    evb: dict
    atype_freeze: np.ndarray = None  # [nt] int32: 1 = atoms of this type are not integrated (atype_freeze, src/glob_v.f90)

    @property
    def n_atom_type(self):
        return len(self.atype_names)


def _gen_exclusions(n_atom, bonds, explicit, n_exclusions):
    """pair_exclusions of one molecule type: 1 = excluded, 2 = 1-4 pair, 0 = normal
    (reference: src/intra_bonded_interactions.f90:574-660, path search without doubling back)."""
    excl = np.zeros((n_atom, n_atom), dtype=np.int32)
    for (i, j) in explicit:
        excl[i - 1, j - 1] = 1
        excl[j - 1, i - 1] = 1
    adj = [[] for _ in range(n_atom)]
    for (i, j) in bonds:
        adj[i - 1].append(j - 1)
        adj[j - 1].append(i - 1)
    for a in adj:
        a.sort()
    max_search = max(n_exclusions, 3)

    def rec(i, j, n_in, traj):
        for loc in range(n_atom):
            if loc in adj[j] and loc not in traj[:n_in]:
                if n_in == 3 and n_exclusions < 3 and excl[i, loc] != 1:
                    excl[i, loc] = 2
                else:
                    excl[i, loc] = 1
                n_out = n_in + 1
                if n_out <= max_search:
                    rec(i, loc, n_out, traj[:n_in] + [loc])

    for i in range(n_atom):
        excl[i, i] = 1
        rec(i, i, 1, [i])
    return excl


def build_params(ff: ForceField, moltype_order=None, lj_comb_rule="opls", n_exclusions=2) -> Params:
    """moltype_order: molecule-type names in the reference's index order (first appearance in the
    coordinate file, then the remaining ``[ moleculetype ]`` blocks in file order,
    src/initialize_routines.f90:699, src/intra_bonded_interactions.f90:1109-1121)."""
    if lj_comb_rule not in ("opls", "standard"):      # src/initialize_routines.f90:611
        raise ValueError("lj_comb_rule parameter isn't recognized: 'standard' (Lorentz-Berthelot) or 'opls'")
    opls = lj_comb_rule == "opls"
    names = [a[0] for a in ff.atypes]
    nt = len(names)
    idx = {n: i for i, n in enumerate(names)}
    chg = np.array([a[1] for a in ff.atypes], dtype=np.float64)
    vdw = np.zeros((nt, nt, 6))
    tmp = np.zeros((nt, nt, 9))
    tmp[:, :, 4] = EXP_INIT
    for row in ff.sapt:          # src/initialize_routines.f90:359-382
        i = idx[row[0]]
        tmp[i, i, :9] = row[1:10]
    vtype = np.zeros((nt, nt), dtype=np.int32)
    gen_cross = np.zeros((nt, nt), dtype=np.int32)
    for i, a in enumerate(ff.atypes):
        vdw[i, i, 0] = a[2]
        vdw[i, i, 1] = a[3]
    for (i, j, p1, p2) in ff.cross_terms:      # opls: (C6, C12), stored C12 first; standard: (epsilon, sigma) as read (:389-405)
        i -= 1
        j -= 1
        if not opls and (p1 > 1000.0 or p2 > 1000.0):
            raise ValueError("looks like combination rule should be opls: cross term parameters look like C6 and C12")
        vdw[i, j, 0] = vdw[j, i, 0] = p2 if opls else p1
        vdw[i, j, 1] = vdw[j, i, 1] = p1 if opls else p2
        gen_cross[i, j] = gen_cross[j, i] = 1

    def to_c12_c6(i, j):      # gen_C12_C6_epsilon_sigma (:620-634)
        eps, sig = vdw[i, j, 0], vdw[i, j, 1]
        vdw[i, j, 0] = 4.0 * eps * sig ** 12
        vdw[i, j, 1] = 4.0 * eps * sig ** 6

    if opls:      # atomic C12 / C6 first, geometric means of those afterwards (:446-451)
        for i in range(nt):
            to_c12_c6(i, i)

    def combine(i, j):
        if vtype[i, j] == 1:
            t = tmp
            a_ex = math.sqrt(t[i, i, 0] * t[j, j, 0])
            a_el = math.sqrt(t[i, i, 1] * t[j, j, 1])
            a_ind = math.sqrt(t[i, i, 2] * t[j, j, 2])
            a_dhf = math.sqrt(t[i, i, 3] * t[j, j, 3])
            vdw[i, j, 0] = a_ex - a_el - a_ind - a_dhf
            vdw[i, j, 1] = (t[i, i, 4] + t[j, j, 4]) * t[i, i, 4] * t[j, j, 4] / (t[i, i, 4] ** 2 + t[j, j, 4] ** 2)
            for k in range(4):
                vdw[i, j, 2 + k] = math.sqrt(t[i, i, 5 + k] * t[j, j, 5 + k])
        elif opls:
            vdw[i, j, 0] = math.sqrt(vdw[i, i, 0] * vdw[j, j, 0])
            vdw[i, j, 1] = math.sqrt(vdw[i, i, 1] * vdw[j, j, 1])
        else:      # Lorentz-Berthelot on (epsilon, sigma) (:603-605)
            e = math.sqrt(vdw[i, i, 0] * vdw[j, j, 0])
            sg = (vdw[i, i, 1] + vdw[j, j, 1]) / 2.0
            vdw[i, j, 0], vdw[i, j, 1] = e, sg

    for i in range(nt):
        for j in range(nt):
            if i == j:
                continue
            if gen_cross[i, j] == 0:
                if vdw[i, i, 0] > SMALL and vdw[j, j, 0] > SMALL:
                    vtype[i, j] = 0
                elif tmp[i, i, 4] > SMALL and tmp[j, j, 4] > SMALL:
                    vtype[i, j] = 1
                else:
                    vtype[i, j] = -1
                combine(i, j)
            else:
                vtype[i, j] = 0
    for i in range(nt):
        if gen_cross[i, i] == 0:
            if vdw[i, i, 0] > SMALL:
                vtype[i, i] = 0
            elif tmp[i, i, 0] > SMALL:
                vtype[i, i] = 1
            else:
                vtype[i, i] = -1
            combine(i, i)
        else:
            vtype[i, i] = 0
    if not opls:      # standard: every Lennard-Jones entry (explicit cross terms included) goes from (epsilon, sigma) to (C12, C6) last (:518-528)
        for i in range(nt):
            for j in range(nt):
                if vtype[i, j] == 0:
                    to_c12_c6(i, j)
    vdw14 = vdw.copy()
    for (n1, n2, c6, c12) in ff.pairtypes:
        i, j = idx[n1], idx[n2]
        vdw14[i, j, 0] = vdw14[j, i, 0] = c12
        vdw14[i, j, 1] = vdw14[j, i, 1] = c6

    # bonded tables
    btype = np.zeros((nt, nt), dtype=np.int32)
    bpar = np.zeros((nt, nt, 3))
    for bt in ff.bondtypes:
        i, j, f = idx[bt[0]], idx[bt[1]], bt[2]
        btype[i, j] = btype[j, i] = f
        p = list(bt[3:])
        for k, v in enumerate(p):
            bpar[i, j, k] = bpar[j, i, k] = v
    atype_ang = {}
    for at in ff.angletypes:
        i, j, k = idx[at[0]], idx[at[1]], idx[at[2]]
        th0 = at[4] * REF_PI / 180.0
        atype_ang[(i, j, k)] = (at[3], th0, at[5])
        atype_ang[(k, j, i)] = (at[3], th0, at[5])
    atype_dih = {}
    for dt in ff.dihedraltypes:
        i, j, k, l = (idx[x] for x in dt[:4])
        f = dt[4]
        p = np.zeros(6)
        if f == 3:
            p[:6] = dt[5:11]
        else:
            p[0] = dt[5] * REF_PI / 180.0
            p[1] = dt[6]
            if f == 1:
                p[2] = dt[7]
        atype_dih[(i, j, k, l)] = (f, p)
        atype_dih[(l, k, j, i)] = (f, p.copy())

    order = list(moltype_order) if moltype_order else []
    for mt in ff.moltypes:
        if mt.name not in order:
            order.append(mt.name)
    by_name = {mt.name: mt for mt in ff.moltypes}
    mass = -np.ones(nt)
    mts = []
    for name in order:
        mt = by_name[name]
        na = len(mt.atoms)
        for (an, m) in mt.atoms:
            mass[idx[an]] = m
        mts.append(dict(
            name=name, n_atom=na,
            atype=np.array([idx[a[0]] for a in mt.atoms], dtype=np.int32),
            bonds=np.array(mt.bonds, dtype=np.int32).reshape(-1, 2) - 1,
            angles=np.array(mt.angles, dtype=np.int32).reshape(-1, 3) - 1,
            dihedrals=np.array(mt.dihedrals, dtype=np.int32).reshape(-1, 4) - 1,
            excl=_gen_exclusions(na, mt.bonds, mt.exclusions, n_exclusions),
            reactive_protons=np.zeros(na, dtype=np.int32),
            reactive_basic=np.zeros(na, dtype=np.int32),
        ))
    midx = {n: i for i, n in enumerate(order)}

    # EVB tables
    nm = len(order)
    evb = dict(
        acid=np.zeros(nm, dtype=np.int32), base=np.zeros(nm, dtype=np.int32),
        conjugate_mol=-np.ones(nm, dtype=np.int32), proton_type=-np.ones(nm, dtype=np.int32),
        heavy_type=-np.ones(nm, dtype=np.int32), conjugate_atype=-np.ones(nt, dtype=np.int32),
        reference_energy=np.zeros(nm), exch_atomic=np.zeros(nt), exch_proton=np.zeros((nm, nm)),
    )
    for p in ff.evb_pairs:
        a, b = midx[p.acid], midx[p.base]
        evb["acid"][a] = 1
        evb["base"][b] = 1
        evb["conjugate_mol"][a] = b
        evb["conjugate_mol"][b] = a
        evb["proton_type"][a] = idx[p.proton_type]
        evb["heavy_type"][a] = idx[p.heavy_type]
        mts[a]["reactive_protons"] = np.array(p.acid_reactive_protons, dtype=np.int32)
        mts[b]["reactive_protons"] = np.array(p.base_reactive_protons, dtype=np.int32)
        mts[a]["reactive_basic"] = np.array(p.acid_acceptor_atoms, dtype=np.int32)
        mts[b]["reactive_basic"] = np.array(p.base_acceptor_atoms, dtype=np.int32)
        for (t1, t2) in p.conjugate_atoms:
            evb["conjugate_atype"][idx[t1]] = idx[t2]
            evb["conjugate_atype"][idx[t2]] = idx[t1]
    for k, v in ff.reference_energy.items():
        evb["reference_energy"][midx[k]] = v
    for k, v in ff.exchange_charge_atomic.items():
        evb["exch_atomic"][idx[k]] = v
    for (m1, m2, q) in ff.exchange_charge_proton:
        evb["exch_proton"][midx[m1], midx[m2]] = q
        evb["exch_proton"][midx[m2], midx[m1]] = q
    evb["donor_acceptor"] = [((idx[r[0]], idx[r[1]], idx[r[2]]), np.array(r[3:9], dtype=np.float64)) for r in ff.donor_acceptor]
    evb["proton_acceptor"] = [((idx[r[0]], idx[r[1]]), np.array(r[2:7], dtype=np.float64)) for r in ff.proton_acceptor]
    evb["coupling"] = [((idx[r[0]], idx[r[1]], idx[r[2]]), int(r[3]), np.array(list(r[4]) + [0.0] * (10 - len(r[4])))) for r in ff.geometry_factor]
    return Params(atype_names=names, atype_chg=chg, atype_mass=mass, vdw_type=vtype, vdw_param=vdw,
                  vdw_param14=vdw14, bond_type=btype, bond_param=bpar, angle_type=atype_ang,
                  dihedral_type=atype_dih, moltype_names=order, moltypes=mts, evb=evb,
                  atype_freeze=np.array([1 if n in set(ff.freeze) else 0 for n in names], dtype=np.int32))
