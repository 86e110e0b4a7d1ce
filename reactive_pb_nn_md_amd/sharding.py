"""Diabat -> rank layout shared by bench.py, the tests and (in C++) csrc/evb.hip / comm.hip:
diabat index d is 0-based, d = 0 is the principal diabat (replicated on every rank); diabat d >= 1 lives on rank
(d-1) mod R in slot (d-1) div R, i.e. in the reference's 1-based numbering diabat D2 on rank (D2-2) mod R.
Per step the ranks exchange the Hamiltonian entries (all-gather, 2 doubles per slot), diagonalise the same matrix, blend THEIR
diabats with the common ground-state vector and sum the partial forces (all-reduce); force stores never travel."""
from __future__ import annotations

import numpy as np


def slots_per_rank(n_diabat, n_ranks):
    return 0 if n_diabat <= 1 else (n_diabat - 1 + n_ranks - 1) // n_ranks


def owner(d, n_ranks):
    return (d - 1) % n_ranks


def local_slot(d, n_ranks):
    return (d - 1) // n_ranks


def diagonalise_and_blend(H_upper, parent, F_diag, F_off):
    """ground state of the tree Hamiltonian + Hellmann-Feynman force (reference: src/ms_evb.f90:242-309)"""
    D = H_upper.shape[0]
    Hs = np.triu(H_upper) + np.triu(H_upper, 1).T
    w, v = np.linalg.eigh(Hs)
    c = v[:, 0]
    F = np.zeros_like(F_diag[0])
    for d in range(D):
        F += c[d] ** 2 * F_diag[d]
        if d > 0:
            F += 2.0 * c[parent[d]] * c[d] * F_off[d]
    return w[0], F, c


def ground_state(H_upper):
    D = H_upper.shape[0]
    Hs = np.triu(H_upper) + np.triu(H_upper, 1).T
    w, v = np.linalg.eigh(Hs)
    return w[0], v[:, 0]


def partial_blend(c, parent, F_diag, F_off, rank, n_ranks):
    """rank's share of the Hellmann-Feynman force (csrc/evb.hip blend_terms): its own diabats, the principal one on rank 0"""
    F = np.zeros_like(F_diag[0])
    for d in range(len(c)):
        if (d == 0 and rank == 0) or (d > 0 and owner(d, n_ranks) == rank):
            F += c[d] ** 2 * F_diag[d]
            if d > 0:
                F += 2.0 * c[parent[d]] * c[d] * F_off[d]
    return F
