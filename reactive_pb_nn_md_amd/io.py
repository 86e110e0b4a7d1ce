"""Readers and writers for the reference's own file formats, as its *parsers* define them:

    .gro                     src/general_routines.f90:184-315   '(I5,2A5,I5,3F8.3)', nm; box line 3 or 9 numbers
    force field .pmt         src/initialize_routines.f90:281-430, :646-691
    topology .top + EVB      src/intra_bonded_interactions.f90:696-1464, src/ms_evb.f90:3170-3623
    simulation parameters    src/read_simulation_parameters.f90:43-146
    trajectory frames        src/general_routines.f90:906-945   (print_gro_file)
    step log                 src/general_routines.f90:870-901   (print_step)
    velocity checkpoint      src/general_routines.f90:997-1026, :147-178

Tokeniser rules follow src/general_routines.f90:2246 / :329-360: whitespace separated tokens, a line whose first token
contains ';' is a comment, a blank line ends a section.  Two limits of the fixed-width formats are lifted on input: molecule
and atom numbers that wrapped at 100 000 are accepted (molecules are told apart by a CHANGE of the number or the name), and
coordinates may carry more than three decimals when the fields are blank separated.
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import numpy as np

from . import forcefield as ffmod
from .forcefield import EvbPair, ForceField, MoleculeType


# ------------------------------------------------------------------------------------------------------------------ .gro
@dataclass
class GroSystem:
    """what synth.System holds, read from a file"""
    box: np.ndarray          # (3,) Angstrom
    xyz: np.ndarray          # (N,3) Angstrom
    atom_names: list
    mol_names: list          # per molecule
    mol_first: np.ndarray
    mol_natom: np.ndarray
    title: str = ""
    hydronium: int = -1      # filled by locate_hydronium()

    @property
    def n_atom(self):
        return len(self.atom_names)

    @property
    def n_mole(self):
        return len(self.mol_names)


def read_gro(path, frame=0) -> GroSystem:
    """frame: which frame of a multi-frame file (a trajectory written by print_gro_file); -1 = the last one"""
    with open(path) as fh:
        lines = fh.readlines()
    starts, k = [], 0
    while k + 1 < len(lines) and lines[k + 1].split():
        starts.append(k)
        k += int(lines[k + 1].split()[0]) + 3
    if not starts:
        raise ValueError("%s: no frame" % path)
    it = iter(lines[starts[frame]:])

    class _F:      # the frame as a file-like object
        @staticmethod
        def readline():
            return next(it, "")
    f = _F
    if True:
        title = f.readline().rstrip("\n")
        n = int(f.readline().split()[0])
        mol_id, mol_name, atom_names = [], [], []
        xyz = np.zeros((n, 3))
        for k in range(n):
            line = f.readline()
            if len(line) < 44:
                raise ValueError("%s: atom line %d too short" % (path, k + 1))
            mol_id.append(int(line[0:5]))
            mol_name.append(line[5:10].strip())
            atom_names.append(line[10:15].strip())
            rest = line[20:]
            tok = rest.split()
            if len(tok) >= 3 and all(_is_float(t) for t in tok[:3]) and len(rest.rstrip("\n")) != 24:
                xyz[k] = [float(t) for t in tok[:3]]              # blank separated, any precision
            else:
                xyz[k] = [float(rest[0:8]), float(rest[8:16]), float(rest[16:24])]
        b = [float(t) for t in f.readline().split()]
    if len(b) == 3:
        box = np.array(b)
    elif len(b) == 9:      # v1(x) v2(y) v3(z) v1(y) v1(z) v2(x) v2(z) v3(x) v3(y); the program stops unless orthorhombic (src/main_ms_evb.f90:62-68)
        if any(abs(v) > 10e-6 for v in b[3:]):
            raise ValueError("%s: only orthorhombic boxes are supported (as in the reference)" % path)
        box = np.array(b[:3])
    else:
        raise ValueError("%s: box line needs 3 or 9 numbers" % path)
    first, names = [], []
    for k in range(n):
        if k == 0 or mol_id[k] != mol_id[k - 1] or mol_name[k] != mol_name[k - 1]:
            first.append(k)
            names.append(mol_name[k])
    first = np.array(first, dtype=np.int32)
    natom = np.diff(np.append(first, n)).astype(np.int32)
    return GroSystem(box=box * 10.0, xyz=xyz * 10.0, atom_names=atom_names, mol_names=names, mol_first=first, mol_natom=natom, title=title)


def _is_float(t):
    try:
        float(t)
        return True
    except ValueError:
        return False


def write_gro_frame(f, title, mol_names, mol_atoms, atom_names, xyz, box):
    """one frame in print_gro_file's layout.  mol_atoms: per molecule the list of (0-based) atom indices in the molecule's
    own order - after a proton transfer the reference prints the NEW molecule membership and renumbers the atoms"""
    n = sum(len(a) for a in mol_atoms)
    f.write(" %s\n" % title)
    f.write(" %d\n" % n)
    count = 1
    for im, atoms in enumerate(mol_atoms):
        for a in atoms:
            x = xyz[a] / 10.0
            f.write("%5d%5s%5s%5d%8.3f%8.3f%8.3f\n" % ((im + 1) % 100000, mol_names[im][:5], atom_names[a][:5], count % 100000, x[0], x[1], x[2]))      # both names right-justified: the reference keeps them that way (trim_end, src/general_routines.f90:767)
            count += 1
    b = np.asarray(box) / 10.0
    f.write("%7.4f%7.4f%7.4f%7.4f%7.4f%7.4f%7.4f%7.4f%7.4f\n" % (b[0], b[1], b[2], 0, 0, 0, 0, 0, 0))


def fortran_e(x, w=16, d=6):
    """Fortran Ew.d editing as the reference's compilers print it: 0.dddddd E+ee (mantissa in [0.1, 1)), right-justified in w columns"""
    from decimal import Decimal, ROUND_HALF_EVEN
    x = float(x)
    if x != x or x in (float("inf"), float("-inf")):
        return ("NaN" if x != x else ("Infinity" if x > 0 else "-Infinity")).rjust(w)
    if x == 0.0:
        body = "0." + "0" * d + "E+00"
    else:
        dx = Decimal(abs(x))
        e = dx.adjusted() + 1                                     # |x| = m 10^e with m in [0.1, 1)
        m = (dx.scaleb(-e)).quantize(Decimal(1).scaleb(-d), rounding=ROUND_HALF_EVEN)
        if m >= 1:                                                # 0.9999996 rounds up to 1.000000
            m = (m / 10).quantize(Decimal(1).scaleb(-d)); e += 1
        body = "0." + str(m).split(".")[1] + ("E%+03d" % e if abs(e) < 100 else "%+04d" % e)
    return (("-" if x < 0 else "") + body).rjust(w)


def write_log_step(f, i_step, delta_t, e_pot, e_kin, terms=None):
    """print_step (src/general_routines.f90:870): terms = (E_elec, E_vdw, E_bond, E_angle, E_dihedral) outside EVB runs"""
    f.write(" i_step , time(ps), potential energy (kJ/mol), kinetic energy (kJ/mol)\n")
    f.write("%9d%10.3f%s%s\n" % (i_step, i_step * delta_t, fortran_e(e_pot), fortran_e(e_kin)))      # '(I9,F10.3,2E16.6)'
    if terms is not None:
        f.write(" Electrostatic ,   VDWs ,   Bond   ,   Angle  ,  Dihedral\n")
        f.write("".join(fortran_e(t) for t in terms) + "\n")                                         # '(5E16.6)'
    f.write(" ------------------------------\n")


def write_velocity_checkpoint(f, i_step, mol_names, mol_atoms, atom_names, vel):
    f.write(" step  %d\n" % i_step)
    for im, atoms in enumerate(mol_atoms):
        for k, a in enumerate(atoms):
            f.write("%5d%5s%5s%5d%14.6f%14.6f%14.6f\n" % ((im + 1) % 100000, mol_names[im][:5], atom_names[a][:5], k + 1, vel[a, 0], vel[a, 1], vel[a, 2]))


def read_velocity_checkpoint(path, n_atom, step=None):
    """velocities (file order = molecule order) of frame `step` (default: the last one) -> (step, (N,3) array)"""
    frames = {}
    with open(path) as f:
        lines = f.readlines()
    k = 0
    while k < len(lines):
        tok = lines[k].split()
        if len(tok) >= 2 and tok[0] == "step":
            s = int(tok[1])
            v = np.zeros((n_atom, 3))
            for a in range(n_atom):
                ln = lines[k + 1 + a]
                v[a] = [float(ln[20:34]), float(ln[34:48]), float(ln[48:62])]
            frames[s] = v
            k += 1 + n_atom
        else:
            k += 1
    if not frames:
        raise ValueError("%s: no velocity frame" % path)
    s = max(frames) if step is None else step
    if s not in frames:
        raise ValueError("%s: no velocity frame for step %d" % (path, s))
    return s, frames[s]


# ----------------------------------------------------------------------------------------------------- tokenised sections
def _lines(path):
    with open(path) as f:
        return [ln.rstrip("\n") for ln in f]


def _tokens(line):
    return line.replace("\t", " ").split()


def _is_comment(line):
    t = _tokens(line)
    return bool(t) and ";" in t[0]


def _section_body(lines, k):
    """lines of the section whose heading is lines[k]: up to the first blank line, comments skipped -> (rows, next index)"""
    rows = []
    k += 1
    while k < len(lines) and _tokens(lines[k]):
        if not _is_comment(lines[k]):
            rows.append(_tokens(lines[k]))
        k += 1
    return rows, k


def _heading(line):
    t = "".join(_tokens(line))
    return t[1:-1] if t.startswith("[") and t.endswith("]") else None


def _num(tok):
    return float(tok.replace("d", "e").replace("D", "e"))


# ------------------------------------------------------------------------------------------------------- force field .pmt
def read_ffpmt(path):
    """-> dict(atypes, freeze, sapt, cross_terms, pairtypes) (src/initialize_routines.f90:281-430, :646-691)"""
    lines = _lines(path)
    k = next((i for i, ln in enumerate(lines) if "solute_species" in ln), None)
    if k is None:
        raise ValueError("%s: no 'solute_species' heading" % path)
    n = int(_tokens(lines[k + 2])[0])
    atypes, freeze = [], []
    for i in range(n):
        t = _tokens(lines[k + 3 + i])
        if len(t) != 5:
            raise ValueError("%s: atom type line needs exactly 5 tokens (name charge epsilon sigma freeze): %r" % (path, lines[k + 3 + i]))
        atypes.append((t[0], _num(t[1]), _num(t[2]), _num(t[3])))
        freeze.append(int(t[4]))
    out = dict(atypes=atypes, freeze=freeze, sapt=[], cross_terms=[], pairtypes=[])
    for i, ln in enumerate(lines):
        t = _tokens(ln)
        if not t:
            continue
        if t[0] == "custom_sapt_parameters":
            j = i + 2
            while j < len(lines) and len(_tokens(lines[j])) == 10:
                r = _tokens(lines[j])
                out["sapt"].append((r[0],) + tuple(_num(x) for x in r[1:]))
                j += 1
        elif t[0] == "cross_terms":
            m = int(_tokens(lines[i + 1])[0])
            for j in range(m):
                r = _tokens(lines[i + 2 + j])
                out["cross_terms"].append((int(r[0]), int(r[1]), _num(r[2]), _num(r[3])))
        elif t[0] == "pairtypes":
            m = int(_tokens(lines[i + 1])[0])
            for j in range(m):
                r = _tokens(lines[i + 2 + j])
                out["pairtypes"].append((r[0], r[1], _num(r[2]), _num(r[3])))
    return out


# ------------------------------------------------------------------------------------------------------------ topology .top
def _typed_rows(rows, n_names):
    out = []
    for r in rows:
        out.append(tuple(r[:n_names]) + (int(r[n_names]),) + tuple(_num(x) for x in r[n_names + 1:]))
    return out


def read_top(path):
    """-> dict with the bonded type tables, molecule types and all EVB sections"""
    lines = _lines(path)
    out = dict(bondtypes=[], angletypes=[], dihedraltypes=[], moltypes=[], reference_energy={}, donor_acceptor=[], proton_acceptor=[],
               geometry_factor=[], exchange_charge_atomic={}, exchange_charge_proton=[], evb_pairs=[])
    k = 0
    cur_mol = None
    cur_pair = None
    while k < len(lines):
        h = _heading(lines[k]) if _tokens(lines[k]) else None
        if h is None:
            k += 1
            continue
        if h in ("bondtypes", "angletypes", "dihedraltypes"):
            rows, k = _section_body(lines, k)
            out[h] = _typed_rows(rows, {"bondtypes": 2, "angletypes": 3, "dihedraltypes": 4}[h])
        elif h == "moleculetype":
            rows, k = _section_body(lines, k)
            cur_mol = MoleculeType(rows[0][0], [], [], [], [], [])
            out["moltypes"].append(cur_mol)
        elif h == "atoms":
            rows, k = _section_body(lines, k)
            for r in rows:
                if len(r) != 3:
                    raise ValueError("%s: [ atoms ] lines need exactly 3 tokens (index type mass)" % path)
                cur_mol.atoms.append((r[1], _num(r[2])))
        elif h in ("bonds", "angles", "dihedrals", "exclusions"):
            rows, k = _section_body(lines, k)
            na = {"bonds": 2, "angles": 3, "dihedrals": 4, "exclusions": 2}[h]
            getattr(cur_mol, h).extend(tuple(int(x) for x in r[:na]) for r in rows)
        elif h == "reference_energy":
            rows, k = _section_body(lines, k)
            for r in rows:
                out["reference_energy"][r[0]] = _num(r[1])
        elif h == "donor_acceptor":
            rows, k = _section_body(lines, k)
            out["donor_acceptor"] = [tuple(r[:3]) + tuple(_num(x) for x in r[3:9]) for r in rows]
        elif h == "proton_acceptor":
            rows, k = _section_body(lines, k)
            out["proton_acceptor"] = [tuple(r[:2]) + tuple(_num(x) for x in r[2:7]) for r in rows]
        elif h == "geometry_factor":
            rows, k = _section_body(lines, k)      # two lines per entry: 't1 t2 t3 functype', then the parameters
            for a, b in zip(rows[0::2], rows[1::2]):
                out["geometry_factor"].append((a[0], a[1], a[2], int(a[3]), [_num(x) for x in b]))
        elif h == "exchange_charge_atomic":
            rows, k = _section_body(lines, k)
            for r in rows:
                out["exchange_charge_atomic"][r[0]] = _num(r[1])
        elif h == "exchange_charge_proton":
            rows, k = _section_body(lines, k)
            out["exchange_charge_proton"] = [(r[0], r[1], _num(r[2])) for r in rows]
        elif h == "evb_pairs":
            rows, k = _section_body(lines, k)
            r = rows[0]
            cur_pair = EvbPair(r[0], r[1], r[2], r[3], [], [], [], [], [])
            out["evb_pairs"].append(cur_pair)
        elif h in ("acid_reactive_protons", "base_reactive_protons", "acid_acceptor_atoms", "base_acceptor_atoms"):
            rows, k = _section_body(lines, k)
            getattr(cur_pair, h).extend(int(r[1]) for r in rows)
        elif h == "conjugate_atoms":
            rows, k = _section_body(lines, k)
            cur_pair.conjugate_atoms.extend((r[0], r[1]) for r in rows)
        else:      # container headings ([ evb_parameters ], [ adiabat_non_bond ], [ diabat_coupling ], [ evb_topology ]) and unknown ones
            k += 1
    return out


def read_forcefield(ffpmt_path, top_path) -> ForceField:
    p, t = read_ffpmt(ffpmt_path), read_top(top_path)
    return ForceField(atypes=p["atypes"], cross_terms=p["cross_terms"], pairtypes=p["pairtypes"], bondtypes=t["bondtypes"],
                      angletypes=t["angletypes"], dihedraltypes=t["dihedraltypes"], moltypes=t["moltypes"],
                      reference_energy=t["reference_energy"], donor_acceptor=t["donor_acceptor"], proton_acceptor=t["proton_acceptor"],
                      geometry_factor=t["geometry_factor"], exchange_charge_atomic=t["exchange_charge_atomic"],
                      exchange_charge_proton=t["exchange_charge_proton"], evb_pairs=t["evb_pairs"], sapt=p["sapt"],
                      freeze=[a[0] for a, f in zip(p["atypes"], p["freeze"]) if f == 1])


# --------------------------------------------------------------------------------------------------- simulation parameters
_REQUIRED = ("n_step", "n_output", "temperature", "n_exclusions", "real_space_cutoff", "n_threads", "delta_t", "verlet_cutoff",
             "na_nslist", "nb_nslist", "nc_nslist")


def read_simpmt(path):
    """-> (strings, numbers) of the two blocks (src/read_simulation_parameters.f90:43-146); defaults of :188-262 applied"""
    lines = _lines(path)
    strings, numbers = {}, {}
    block = 0
    for ln in lines:
        t = _tokens(ln)
        if not t:
            continue
        if block == 0:
            if "Simulation" in ln and "Methodology" in ln:
                block = 1
            continue
        if block == 1 and len(t) >= 2 and "Simulation" in t[0] and "Param" in t[1]:
            block = 2
            continue
        if len(t) < 2:
            continue
        if block == 1:
            strings[t[0]] = t[1]
        else:
            numbers[t[0]] = _num(t[1])
    if block != 2:
        raise ValueError("%s: needs a 'Simulation Methodology' and a 'Simulation Parameters' block" % path)
    for k in ("ensemble", "lj_comb_rule"):
        if k not in strings:
            raise ValueError("%s: '%s' missing" % (path, k))
    for k in _REQUIRED:
        if k not in numbers:
            raise ValueError("%s: '%s' missing" % (path, k))
    numbers.setdefault("alpha_sqrt", 0.3)
    numbers.setdefault("pme_grid", 60)
    numbers.setdefault("spline_order", 6)
    numbers.setdefault("debug", 0)
    return strings, numbers


# ------------------------------------------------------------------------------------------------------------ put together
def locate_hydronium(gro: GroSystem, ff: ForceField) -> int:
    """the one molecule whose type is an acid (update_hydronium_molecule_index, src/ms_evb.f90:96-120); -1 without EVB pairs"""
    acids = {p.acid for p in ff.evb_pairs}
    hits = [m for m, nm in enumerate(gro.mol_names) if nm in acids]
    if not ff.evb_pairs:
        return -1
    if len(hits) < 1:
        raise ValueError("need at least one hydronium molecule!")
    if len(hits) > 1:
        raise ValueError("can't have more than 1 hydronium")
    return hits[0]


def load_inputs(gro_path, ffpmt_path, top_path, simpmt_path, frame=0, evb=True):
    """the four input files of the reference's command line -> (SystemSpec, GroSystem, ForceField, strings, numbers)"""
    from . import system as sysmod
    gro = read_gro(gro_path, frame)
    ff = read_forcefield(ffpmt_path, top_path)
    strings, numbers = read_simpmt(simpmt_path)
    gro.hydronium = locate_hydronium(gro, ff) if evb else -1      # (a build of the reference without MS-EVB never looks for one)
    order = []
    for nm in gro.mol_names:      # molecule types in order of first appearance in the .gro, the .top's extra ones after them
        if nm not in order:
            order.append(nm)
    params = ffmod.build_params(ff, moltype_order=order, lj_comb_rule=strings["lj_comb_rule"], n_exclusions=int(numbers["n_exclusions"]))
    rc, rv = numbers["real_space_cutoff"], numbers["verlet_cutoff"]
    L = float(gro.box.min())
    if not (rc < rv < L / 2.0):      # src/general_routines.f90:1115-1130
        raise ValueError("need real_space_cutoff < verlet_cutoff < L/2")
    if int(numbers["spline_order"]) != 6:
        raise NotImplementedError("spline_order %d: only 6 is self-consistent in the reference (src/pme.f90:247)" % int(numbers["spline_order"]))
    spec = sysmod.from_synth(gro, params, rc=rc, rv=rv, alpha=numbers["alpha_sqrt"], pme_grid=int(numbers["pme_grid"]), nslist=int(numbers["na_nslist"]))
    spec.delta_t = numbers["delta_t"]
    return spec, gro, ff, strings, numbers
