"""Python host side of libmsevb_hip.so: a thin ctypes binding that mirrors the reference's
procedure surface for the hot path (same names, same argument meaning, errors raised instead of
``stop``):

    calculate_total_force_energy            src/total_energy_forces.f90:19
    real_space_energy_force                 src/pair_int_real_space.f90:60
    pme_reciprocal_space_energy_force       src/pme.f90:28
    intra_molecular_energy_force            src/intra_bonded_interactions.f90:17
    construct_verlet_list / update_verlet_displacements   src/general_routines.f90:1339 / :1259
    ms_evb_calculate_total_force_energy     src/ms_evb.f90:181
    md_integrate_atomic                     src/md_integration.f90:438

There is NO CPU fallback: the library must load and a HIP device must be visible, otherwise
``MsevbError`` is raised.  Arrays cross the boundary in the reference's own layout (1-based indices,
Fortran order), see include/msevb_hip.h.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MSEVB_LIB") or os.path.join(_HERE, "libmsevb_hip.so")      # MSEVB_LIB: a variant build (tools/variant.sh) for A/B runs on one box
_LIB = None

N_ENERGY = 8
E_NAMES = ("potential", "elec", "vdw", "bond", "angle", "dihedral", "recip", "_")


class MsevbError(RuntimeError):
    pass


def load_library():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise MsevbError("libmsevb_hip.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'`); "
                             "there is no CPU fallback for the hot path")
        L = C.CDLL(LIB_PATH)
        L.msevb_last_error.restype = C.c_char_p
        L.msevb_device_forces.restype = C.c_void_p
        L.msevb_device_positions.restype = C.c_void_p
        _LIB = L
    return _LIB


def exported_symbols():
    """names declared in include/msevb_hip.h"""
    import re
    hdr = open(os.path.join(os.path.dirname(_HERE), "include", "msevb_hip.h")).read()
    return sorted(set(re.findall(r"\b(msevb_[A-Za-z_0-9]+)\s*\(", hdr)))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _fi(a):  # int32, Fortran order, kept alive by the caller
    return np.asfortranarray(np.asarray(a, dtype=np.int32))


def _fd(a):
    return np.asfortranarray(np.asarray(a, dtype=np.float64))


class MsEvbHip:
    """One context = one GPU.  Built from a SystemSpec (reactive_pb_nn_md_amd.system)."""

    def __init__(self, spec, device=0, finalize=True):
        self.L = load_library()
        self.spec = spec
        self.N, self.M, self.K = spec.n_atom, spec.n_mole, spec.pme_grid
        self.ctx = C.c_void_p()
        rc = self.L.msevb_create(C.byref(self.ctx), int(device))
        self._chk(rc)
        p = spec.params
        nt, nmt = p.n_atom_type, len(p.moltypes)
        self.nt, self.nmt = nt, nmt
        box = _fd(np.diag(spec.box))
        self._chk(self.L.msevb_set_system(self.ctx, self.N, self.M, _dp(box), C.c_double(spec.rc), C.c_double(spec.rv), C.c_double(spec.alpha),
                                          self.K, 6, 100000, 100000))
        at, ch, ms = _fi(spec.atype + 1), _fd(spec.charge), _fd(spec.mass)
        mf, mn, mt = _fi(spec.mol_first + 1), _fi(spec.mol_natom), _fi(spec.mol_type + 1)
        self._chk(self.L.msevb_set_atoms(self.ctx, _ip(at), _dp(ch), _dp(ms), _ip(mf), _ip(mn), _ip(mt)))
        vt, vp, vp14 = _fi(p.vdw_type), _fd(p.vdw_param), _fd(p.vdw_param14)
        bt, bp, ac = _fi(p.bond_type), _fd(p.bond_param), _fd(p.atype_chg)
        self._chk(self.L.msevb_set_forcefield(self.ctx, nt, nt, _ip(vt), _dp(vp), _dp(vp14), _ip(bt), _dp(bp), _dp(ac)))
        ai, ap = spec.angle_tables()
        ai = ai.copy(); ai[:, :3] += 1
        ai_f, ap_f = _fi(ai.T), _fd(ap.T)
        self._chk(self.L.msevb_set_angle_types(self.ctx, len(ai), _ip(ai_f), _dp(ap_f)))
        di, dpar = spec.dihedral_tables()
        di = di.copy(); di[:, :4] += 1
        di_f, dp_f = _fi(di.T), _fd(dpar.T)
        self._chk(self.L.msevb_set_dihedral_types(self.ctx, len(di), _ip(di_f), _dp(dp_f)))
        for it, m in enumerate(p.moltypes):
            a = _fi(m["atype"] + 1)
            b, an, dh = _fi((m["bonds"] + 1).T), _fi((m["angles"] + 1).T), _fi((m["dihedrals"] + 1).T)
            ex, rp, rb = _fi(m["excl"]), _fi(m["reactive_protons"]), _fi(m["reactive_basic"])
            self._chk(self.L.msevb_set_molecule_type(self.ctx, it + 1, m["n_atom"], _ip(a), len(m["bonds"]), _ip(b), len(m["angles"]), _ip(an),
                                                     len(m["dihedrals"]), _ip(dh), _ip(ex), _ip(rp), _ip(rb)))
        if p.evb:
            self._set_evb(p.evb, nt, nmt, spec)
        if finalize:
            self._chk(self.L.msevb_finalize(self.ctx))
            if getattr(p, "atype_freeze", None) is not None and np.any(p.atype_freeze):
                self.set_frozen_types(p.atype_freeze)
            self.set_positions(spec.xyz)

    def _set_evb(self, e, nt, nmt, spec):
        ldi = 15

        def table(rows, nty, npar):
            ti = np.zeros((ldi, nty), dtype=np.int32)
            tp = np.zeros((ldi, npar))
            for r, (tys, par) in enumerate(rows):
                ti[r, :] = np.array(tys) + 1
                tp[r, :len(par)] = par
            return _fi(ti), _fd(tp)
        da_i, da_p = table(e["donor_acceptor"], 3, 6)
        pa_i, pa_p = table(e["proton_acceptor"], 2, 5)
        cp_i, cp_p = table([(c[0], c[2]) for c in e["coupling"]], 3, 10)
        cp_t = np.zeros(ldi, dtype=np.int32)
        cp_t[:len(e["coupling"])] = [c[1] for c in e["coupling"]]
        k = [_fi(e["acid"]), _fi(e["base"]), _fi(e["conjugate_mol"] + 1), _fi(e["proton_type"] + 1), _fi(e["heavy_type"] + 1),
             _fi(e["conjugate_atype"] + 1), _fd(e["reference_energy"]), _fd(e["exch_atomic"]), _fd(e["exch_proton"])]
        self._chk(self.L.msevb_set_evb(self.ctx, nmt, nmt, ldi, _ip(k[0]), _ip(k[1]), _ip(k[2]), _ip(k[3]), _ip(k[4]), _ip(k[5]), _dp(k[6]), _dp(k[7]),
                                       _dp(k[8]), _ip(da_i), _dp(da_p), _ip(pa_i), _dp(pa_p), _ip(cp_i), _dp(cp_p), _ip(cp_t),
                                       int(spec.hydronium) + 1, int(spec.max_chain)))

    def _chk(self, rc):
        if rc != 0:
            msg = self.L.msevb_last_error(self.ctx)
            raise MsevbError("libmsevb_hip error %d: %s" % (rc, msg.decode() if msg else "?"))

    def close(self):
        if self.ctx:
            self.L.msevb_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- data movement
    def set_positions(self, xyz):
        a = np.ascontiguousarray(xyz, dtype=np.float64)
        self._chk(self.L.msevb_set_positions(self.ctx, _dp(a)))

    def set_host_order(self, host_to_library=None):
        """atom order of the caller's (N,3) arrays: entry i = the library's index (0-based here) of the caller's atom i; None = same order.
        What a Fortran host calls after the reference re-packed its arrays on a proton hop (msevb_set_host_order, include/msevb_hip.h)"""
        if host_to_library is None:
            self._chk(self.L.msevb_set_host_order(self.ctx, None))
        else:
            a = np.ascontiguousarray(np.asarray(host_to_library) + 1, dtype=np.int32)
            self._chk(self.L.msevb_set_host_order(self.ctx, _ip(a)))

    def set_frozen_types(self, atype_freeze):
        """atype_freeze (src/md_integration.f90:473, :512): atoms of these types keep position and velocity"""
        a = np.ascontiguousarray(atype_freeze, dtype=np.int32)
        self._chk(self.L.msevb_set_frozen_types(self.ctx, _ip(a), len(a)))

    def set_box(self, box):
        """periodic_box_change (src/pme.f90:634): new orthorhombic box lengths; CB, reciprocal lattice and Verlet list follow"""
        b = np.ascontiguousarray(box, dtype=np.float64)
        self._chk(self.L.msevb_set_box(self.ctx, _dp(b)))
        import dataclasses
        self.spec = dataclasses.replace(self.spec, box=b.copy())      # the caller's spec object is left alone

    def get_positions(self):
        out = np.zeros((self.N, 3))
        self._chk(self.L.msevb_get_positions(self.ctx, _dp(out)))
        return out

    def set_velocities(self, v):
        a = np.ascontiguousarray(v, dtype=np.float64)
        self._chk(self.L.msevb_set_velocities(self.ctx, _dp(a)))

    def get_velocities(self):
        out = np.zeros((self.N, 3))
        self._chk(self.L.msevb_get_velocities(self.ctx, _dp(out)))
        return out

    def get_forces(self):
        out = np.zeros((self.N, 3))
        self._chk(self.L.msevb_get_forces(self.ctx, _dp(out)))
        return out

    def table(self, name):
        n = self.L.msevb_get_table(self.ctx, name.encode(), None, 0)
        if n < 0:
            raise MsevbError("unknown table " + name)
        out = np.zeros(n)
        self.L.msevb_get_table(self.ctx, name.encode(), _dp(out), n)
        return out

    def ewald_self(self):
        v = C.c_double()
        self._chk(self.L.msevb_get_ewald_self(self.ctx, C.byref(v)))
        return v.value

    # ---- reference-named hot-path calls
    def construct_verlet_list(self):
        self._chk(self.L.msevb_construct_verlet_list(self.ctx))

    def update_verlet_displacements(self):
        f = C.c_int()
        self._chk(self.L.msevb_update_verlet_displacements(self.ctx, C.byref(f)))
        return f.value

    def verlet_flag(self):
        """flag_verlet_list as the last force call left it (1 = the next force call rebuilds the list)"""
        f = C.c_int()
        self._chk(self.L.msevb_get_verlet_flag(self.ctx, C.byref(f)))
        return f.value

    def get_verlet_list(self):
        n = C.c_long()
        self._chk(self.L.msevb_get_verlet_list(self.ctx, None, None, C.byref(n)))
        vp = np.zeros(self.N + 1, dtype=np.int32)
        nl = np.zeros(max(n.value, 1), dtype=np.int32)
        self._chk(self.L.msevb_get_verlet_list(self.ctx, _ip(vp), _ip(nl), C.byref(n)))
        return vp, nl[:n.value]

    def zero_forces(self):
        self._chk(self.L.msevb_zero_forces(self.ctx))

    def real_space_energy_force(self):
        e = np.zeros(4)
        self._chk(self.L.msevb_real_space_energy_force(self.ctx, _dp(e)))
        return e

    def pme_reciprocal_space_energy_force(self):
        e = C.c_double()
        self._chk(self.L.msevb_pme_reciprocal_space_energy_force(self.ctx, C.byref(e)))
        return e.value

    def intra_molecular_energy_force(self):
        e = np.zeros(3)
        self._chk(self.L.msevb_intra_molecular_energy_force(self.ctx, _dp(e)))
        return e

    def calculate_total_force_energy(self, want_force=True):
        e = np.zeros(N_ENERGY)
        f = np.zeros((self.N, 3)) if want_force else None
        self._chk(self.L.msevb_calculate_total_force_energy(self.ctx, _dp(e), _dp(f) if want_force else None))
        return dict(zip(E_NAMES, e)), f

    def pme_state(self):
        K = self.K
        Q, th, u, fr = np.zeros((K, K, K)), np.zeros((K, K, K)), np.zeros((self.N, 3)), np.zeros((self.N, 3))
        self._chk(self.L.msevb_get_pme_state(self.ctx, _dp(Q), _dp(th), _dp(u), _dp(fr)))
        return Q, th, u, fr

    def grid_Q(self, chg, xyz_scale):
        """grid_Q (src/pme.f90:184) at the reference's granularity: scaled coordinates and charges in, Q(K,K,K) out (host arrays)"""
        K = self.K
        Q = np.zeros((K, K, K))
        chg, u = np.ascontiguousarray(chg, dtype=np.float64), np.ascontiguousarray(xyz_scale, dtype=np.float64)
        self._chk(self.L.msevb_grid_Q(self.ctx, _dp(Q), _dp(chg), _dp(u), len(chg)))
        return Q

    def modify_Q_grid(self, Q, chg, xyz_scale, operation):
        """modify_Q_grid (src/pme.f90:275): Q +/-= the stencils of a few atoms, in place"""
        chg, u = np.ascontiguousarray(chg, dtype=np.float64), np.ascontiguousarray(xyz_scale, dtype=np.float64)
        assert Q.flags.c_contiguous and Q.dtype == np.float64
        self._chk(self.L.msevb_modify_Q_grid(self.ctx, _dp(Q), _dp(chg), _dp(u), len(chg), int(operation)))
        return Q

    def derivative_grid_Q(self, FQ, chg, xyz_scale, kk, conv):
        """derivative_grid_Q (src/pme.f90:346) for all the atoms handed in: reciprocal forces (n,3) from theta_conv_Q"""
        chg, u = np.ascontiguousarray(chg, dtype=np.float64), np.ascontiguousarray(xyz_scale, dtype=np.float64)
        FQ, kk = np.ascontiguousarray(FQ, dtype=np.float64), np.ascontiguousarray(kk, dtype=np.float64)      # kk: the memory of the Fortran array kk(3,3)
        F = np.zeros((len(chg), 3))
        self._chk(self.L.msevb_derivative_grid_Q(self.ctx, _dp(F), _dp(FQ), _dp(chg), _dp(u), len(chg), kk.ctypes.data_as(C.POINTER(C.c_double)), C.c_double(conv)))
        return F

    def pme_indices(self):
        near = np.zeros((self.N, 3), dtype=np.int32)
        spl = np.zeros((self.N, 3, 6), dtype=np.int32)
        self._chk(self.L.msevb_get_pme_indices(self.ctx, _ip(near), _ip(spl)))
        return near, spl

    def ms_evb_calculate_total_force_energy(self, want_force=True):
        e = C.c_double()
        f = np.zeros((self.N, 3)) if want_force else None
        self._chk(self.L.msevb_ms_evb_calculate_total_force_energy(self.ctx, C.byref(e), _dp(f) if want_force else None))
        return e.value, f

    def evb_state(self):
        nd = C.c_int()
        self._chk(self.L.msevb_get_evb_state(self.ctx, C.byref(nd), None, None, None, None, None, None))
        D, mc = nd.value, self.spec.max_chain
        plog = np.zeros((D, mc, 5), dtype=np.int32)
        cm = np.zeros(D, dtype=np.int32)
        H = np.zeros((D, D))
        ev = np.zeros(D)
        pr, nh = C.c_int(), C.c_int()
        self._chk(self.L.msevb_get_evb_state(self.ctx, C.byref(nd), _ip(plog), _ip(cm), _dp(H), _dp(ev), C.byref(pr), C.byref(nh)))
        return dict(D=D, proton_log=plog, coupling_matrix=cm, H=H, eigvec=ev, principle_diabat=pr.value, new_hydronium_molecule=nh.value)

    def evb_element_force(self, i, j):
        f = np.zeros((self.N, 3))
        self._chk(self.L.msevb_get_evb_element_force(self.ctx, int(i), int(j), _dp(f)))
        return f

    def evb_partial_blend(self, eigvec, download=True):
        """this rank's share of the Hellmann-Feynman force for a given ground-state vector (sharding tests); download=False leaves it on
        the device (timing of a virtual rank's step, tools/virtual_rank_times.py)"""
        ev = np.ascontiguousarray(eigvec, dtype=np.float64)
        if not download:
            self._chk(self.L.msevb_evb_partial_blend(self.ctx, _dp(ev), None))
            return None
        f = np.zeros((self.N, 3))
        self._chk(self.L.msevb_evb_partial_blend(self.ctx, _dp(ev), _dp(f)))
        return f

    def hop_log(self, clear=True):
        """principal-diabat switches since the last clearing call: (array of {EVB force call number, old, new hydronium molecule}, calls so far)"""
        n, calls = C.c_int(), C.c_long()
        self._chk(self.L.msevb_get_hop_log(self.ctx, C.byref(n), None, 0, C.byref(calls), 0))
        log = np.zeros((max(n.value, 1), 3), dtype=np.int32)
        self._chk(self.L.msevb_get_hop_log(self.ctx, C.byref(n), _ip(log), max(n.value, 1), C.byref(calls), 1 if clear else 0))
        return log[:n.value], calls.value

    def topology(self):
        at, ch = np.zeros(self.N, dtype=np.int32), np.zeros(self.N)
        mn, mt, ma = np.zeros(self.M, dtype=np.int32), np.zeros(self.M, dtype=np.int32), np.zeros((self.M, 8), dtype=np.int32)
        hy = C.c_int()
        self._chk(self.L.msevb_get_topology(self.ctx, _ip(at), _dp(ch), _ip(mn), _ip(mt), _ip(ma), C.byref(hy)))
        return dict(atom_type_index=at, charge=ch, mol_n_atom=mn, mol_type_index=mt, mol_atom_index=ma, hydronium_molecule=hy.value)

    def md_integrate_atomic(self, n_steps, delta_t=None, use_evb=False):
        e = np.zeros((n_steps, 2))
        self._chk(self.L.msevb_md_integrate_atomic(self.ctx, int(n_steps), C.c_double(delta_t if delta_t is not None else self.spec.delta_t),
                                                   int(bool(use_evb)), _dp(e)))
        return e

    def calculate_kinetic_energy(self):
        v = C.c_double()
        self._chk(self.L.msevb_calculate_kinetic_energy(self.ctx, C.byref(v)))
        return v.value

    @staticmethod
    def comm_unique_id(lib=None):
        """128 opaque bytes (ncclUniqueId) made by rank 0 and handed to every rank's comm_init"""
        L = lib or load_library()
        raw = (C.c_ubyte * 128)()
        rc = L.msevb_comm_unique_id(raw)
        if rc != 0:
            raise RuntimeError("msevb_comm_unique_id failed (%d)" % rc)
        return bytes(raw)

    def comm_init(self, unique_id, rank, n_ranks):
        """join the RCCL communicator: diabats (and the principal diabat's pair loop) are sharded over n_ranks processes"""
        raw = (C.c_ubyte * 128)(*unique_id)
        self._chk(self.L.msevb_comm_init(self.ctx, raw, int(rank), int(n_ranks)))

    def comm_set_virtual(self, rank, n_ranks):
        self._chk(self.L.msevb_comm_set_virtual(self.ctx, int(rank), int(n_ranks)))

    def comm_get_payload(self, n_doubles):
        a = np.zeros(int(n_doubles))
        self._chk(self.L.msevb_comm_get_payload(self.ctx, _dp(a), int(n_doubles)))
        return a

    def evb_payload_size(self, n_diabat, n_ranks):
        return int(self.L.msevb_evb_payload_size(int(n_diabat), int(n_ranks)))

    def comm_set_virtual_payload(self, payloads):
        """virtual rank: the n_ranks payloads (evb_payload_pack) a real all-gather would deliver, in rank order; the rank then solves the gathered Hamiltonian on the
        device and blends its share without the caller (None / empty: back to the caller-merged mode)"""
        a = np.ascontiguousarray(payloads if payloads is not None else [], dtype=np.float64).ravel()
        self._chk(self.L.msevb_comm_set_virtual_payload(self.ctx, _dp(a) if a.size else None, int(a.size)))

    def set_option(self, name, value):
        self._chk(self.L.msevb_set_option(self.ctx, name.encode(), C.c_double(value)))

    def info(self, name):
        v = C.c_double()
        self._chk(self.L.msevb_get_info(self.ctx, name.encode(), C.byref(v)))
        return v.value

    def phase_times(self):
        n = C.c_int()
        names = (C.c_char_p * 16)()
        ms = (C.c_double * 16)()
        calls = (C.c_long * 16)()
        self._chk(self.L.msevb_get_phase_times(self.ctx, C.byref(n), names, ms, calls))
        return {names[i].decode(): (ms[i], calls[i]) for i in range(n.value)}

    def synchronize(self):
        self._chk(self.L.msevb_synchronize(self.ctx))
