"""TEST INFRASTRUCTURE ONLY (oracle/): ctypes front-end of liboracle.so (the CPU restatement).
May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", HERE, "liboracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_create.restype = C.c_void_p
        for f in ("orc_total", "orc_ewald_self", "orc_md_step", "orc_kinetic_energy", "orc_evb_force_energy", "orc_evb_force_energy_delta", "orc_evb_repulsion_principal"):
            getattr(L, f).restype = C.c_double
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, _p(a, C.c_int)


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, _p(a, C.c_double)


class Oracle:
    def __init__(self, spec, threads=1):
        L = lib()
        self.L, self.spec = L, spec
        self.N, self.M, self.K = spec.n_atom, spec.n_mole, spec.pme_grid
        p = spec.params
        nt, nmt = p.n_atom_type, len(p.moltypes)
        self.nt, self.nmt = nt, nmt
        box, pb = _d(spec.box)
        self.h = C.c_void_p(L.orc_create(self.N, self.M, nt, nmt, self.K, pb, C.c_double(spec.rc), C.c_double(spec.rv), C.c_double(spec.alpha)))
        L.orc_set_threads(self.h, threads)
        keep = [_i(spec.atype), _d(spec.charge), _d(spec.mass), _i(spec.mol_first), _i(spec.mol_natom), _i(spec.mol_type)]
        L.orc_set_atoms(self.h, *[k[1] for k in keep])
        k = [_i(p.vdw_type), _d(p.vdw_param), _d(p.vdw_param14)]
        L.orc_set_vdw(self.h, *[x[1] for x in k])
        ai, ap = spec.angle_tables()
        di, dp = spec.dihedral_tables()
        k = [_i(p.bond_type), _d(p.bond_param), _i(ai), _d(ap), _i(di), _d(dp)]
        L.orc_set_bonded(self.h, k[0][1], k[1][1], len(ai), k[2][1], k[3][1], len(di), k[4][1], k[5][1])
        for it, mt in enumerate(p.moltypes):
            k = [_i(mt["atype"]), _i(mt["bonds"]), _i(mt["angles"]), _i(mt["dihedrals"]), _i(mt["excl"]), _i(mt["reactive_protons"]), _i(mt["reactive_basic"])]
            L.orc_set_moltype(self.h, it, mt["n_atom"], k[0][1], len(mt["bonds"]), k[1][1], len(mt["angles"]), k[2][1],
                              len(mt["dihedrals"]), k[3][1], k[4][1], k[5][1], k[6][1])
        c, pc = _d(p.atype_chg)
        L.orc_set_atype_chg(self.h, pc)
        fz = getattr(p, "atype_freeze", None)
        if fz is not None:
            k = _i(np.asarray(fz)[:nt])
            L.orc_set_atype_freeze(self.h, k[1])
        if p.evb:
            e = p.evb
            da_t = np.array([r[0] for r in e["donor_acceptor"]], dtype=np.int32).reshape(-1, 3)
            da_p = np.array([r[1] for r in e["donor_acceptor"]], dtype=np.float64).reshape(-1, 6)
            pa_t = np.array([r[0] for r in e["proton_acceptor"]], dtype=np.int32).reshape(-1, 2)
            pa_p = np.array([r[1] for r in e["proton_acceptor"]], dtype=np.float64).reshape(-1, 5)
            cp_t = np.array([r[0] for r in e["coupling"]], dtype=np.int32).reshape(-1, 3)
            cp_f = np.array([r[1] for r in e["coupling"]], dtype=np.int32)
            cp_p = np.array([r[2] for r in e["coupling"]], dtype=np.float64).reshape(-1, 10)
            k = [_i(e["acid"]), _i(e["base"]), _i(e["conjugate_mol"]), _i(e["proton_type"]), _i(e["heavy_type"]), _i(e["conjugate_atype"]),
                 _d(e["reference_energy"]), _d(e["exch_atomic"]), _d(e["exch_proton"]), _i(da_t), _d(da_p), _i(pa_t), _d(pa_p), _i(cp_t), _i(cp_f), _d(cp_p)]
            L.orc_set_evb(self.h, *[x[1] for x in k[:9]], len(da_t), k[9][1], k[10][1], len(pa_t), k[11][1], k[12][1],
                          len(cp_t), k[13][1], k[14][1], k[15][1], int(spec.hydronium), int(spec.max_chain))
        self.set_xyz(spec.xyz)
        L.orc_build_tables(self.h)

    def __del__(self):
        try:
            self.L.orc_destroy(self.h)
        except Exception:
            pass

    def set_xyz(self, xyz):
        a, p = _d(xyz)
        self.L.orc_set_xyz(self.h, p)

    def get_xyz(self):
        out = np.zeros((self.N, 3))
        self.L.orc_get_xyz(self.h, _p(out, C.c_double))
        return out

    def table(self, name):
        n = self.L.orc_get_table(self.h, name.encode(), None, 0)
        out = np.zeros(n)
        self.L.orc_get_table(self.h, name.encode(), _p(out, C.c_double), n)
        return out

    def ewald_self(self):
        return self.L.orc_ewald_self(self.h)

    def verlet_build(self, nslist=None):
        n = self.L.orc_verlet_build(self.h, int(nslist or self.spec.nslist))
        self.L.orc_verlet_check(self.h, 1)
        return n

    def verlet_flag(self):
        return int(self.L.orc_verlet_flag(self.h))

    def n_diabat(self):
        return int(self.L.orc_evb_n_diabat(self.h))

    def verlet_get(self):
        vp = np.zeros(self.N + 1, dtype=np.int32)
        self.L.orc_verlet_get(self.h, _p(vp, C.c_int), None)
        nl = np.zeros(max(int(vp[-1]), 1), dtype=np.int32)
        self.L.orc_verlet_get(self.h, _p(vp, C.c_int), _p(nl, C.c_int))
        return vp, nl[:vp[-1]]

    def _stage(self, fn):
        E = np.zeros(5)
        F = np.zeros((self.N, 3))
        getattr(self.L, fn)(self.h, _p(E, C.c_double), _p(F, C.c_double))
        return E, F

    def pair(self):
        return self._stage("orc_pair_real_space")

    def intra_nonbonded(self):
        return self._stage("orc_intra_nonbonded")

    def pme(self):
        return self._stage("orc_pme_recip")

    def bonded(self):
        return self._stage("orc_bonded")

    def total(self):
        E = np.zeros(5)
        F = np.zeros((self.N, 3))
        e = self.L.orc_total(self.h, _p(E, C.c_double), _p(F, C.c_double))
        return e, E, F

    def grid(self):
        K = self.K
        Q, th, u, kk = np.zeros((K, K, K)), np.zeros((K, K, K)), np.zeros((self.N, 3)), np.zeros((3, 3))
        self.L.orc_get_grid(self.h, _p(Q, C.c_double), _p(th, C.c_double), _p(u, C.c_double), _p(kk, C.c_double))
        return Q, th, u, kk

    def force_recip(self):
        F = np.zeros((self.N, 3))
        self.L.orc_get_force_recip(self.h, _p(F, C.c_double))
        return F

    def md_step(self, vel, F, dt, evb=False):
        return self.L.orc_md_step(self.h, _p(vel, C.c_double), _p(F, C.c_double), C.c_double(dt), int(evb))

    def kinetic_energy(self, vel):
        return self.L.orc_kinetic_energy(self.h, _p(vel, C.c_double))

    def apply_principal(self):
        """src/ms_evb.f90:218-227: adopt the topology of the new principal diabat found by the last evb() call"""
        self.L.orc_evb_apply_principal(self.h)

    def topology(self):
        M = len(self.spec.mol_natom)
        at, ch = np.zeros(self.N, dtype=np.int32), np.zeros(self.N)
        mn, mt, ma = np.zeros(M, dtype=np.int32), np.zeros(M, dtype=np.int32), np.zeros((M, 8), dtype=np.int32)
        self.L.orc_get_topology(self.h, _p(at, C.c_int), _p(ch, C.c_double), _p(mn, C.c_int), _p(mt, C.c_int), _p(ma, C.c_int))
        return dict(atom_type_index=at, charge=ch, mol_n_atom=mn, mol_type_index=mt, mol_atom_index=ma)

    def evb_repulsion_principal(self):
        """ms_evb_intermolecular_repulsion (three-atom + Born-Mayer terms) of the principal topology: (energy, forces)"""
        F = np.zeros((self.N, 3))
        e = self.L.orc_evb_repulsion_principal(self.h, _p(F, C.c_double))
        return float(e), F

    def rep_switch(self, r, rs, rc):
        sw, dsw = C.c_double(), C.c_double()
        self.L.orc_evb_rep_switch(C.c_double(r), C.c_double(rs), C.c_double(rc), C.byref(sw), C.byref(dsw))
        return sw.value, dsw.value

    def evb(self, delta=False):
        """delta=True: the reference's delta-update formulation (orc_evb_force_energy_delta) instead of every diabat from scratch"""
        F = np.zeros((self.N, 3))
        e = (self.L.orc_evb_force_energy_delta if delta else self.L.orc_evb_force_energy)(self.h, _p(F, C.c_double))
        D = self.L.orc_evb_n_diabat(self.h)
        mc = self.spec.max_chain
        plog = np.zeros((D, mc, 5), dtype=np.int32)
        parent = np.zeros(D, dtype=np.int32)
        H = np.zeros((D, D))
        eig = np.zeros(D)
        Fd = np.zeros((D, self.N, 3))
        Fo = np.zeros((D, self.N, 3))
        self.L.orc_evb_get(self.h, _p(plog, C.c_int), _p(parent, C.c_int), _p(H, C.c_double), _p(eig, C.c_double), _p(Fd, C.c_double), _p(Fo, C.c_double))
        nh = C.c_int(0)
        pr = self.L.orc_evb_principal(self.h, C.byref(nh))
        return dict(E=e, F=F, D=D, proton_log=plog, parent=parent, H=H, eigvec=eig, F_diag=Fd, F_off=Fo, principal=pr, new_hydronium=nh.value)
