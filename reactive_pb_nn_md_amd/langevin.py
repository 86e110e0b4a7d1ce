"""Host restatement of the library's Langevin kick (csrc/integrate.hip; formula of langevin_integrator, src/md_integration.f90:186-228):
the counter-based normal deviates (Philox4x32-10, counter = (atom, kick number), key = seed; Box-Muller) and the velocity update.
Used by the tests to predict a thermostatted step exactly, and by users who want to replay the noise of a run."""
from __future__ import annotations

import numpy as np

BOLTZMANN = 0.008314462      # src/glob_v.f90:390


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & 0xFFFFFFFF for c in (c0, c1, c2, c3))
    k0, k1 = np.uint64(k0 & 0xFFFFFFFF), np.uint64(k1 & 0xFFFFFFFF)
    m = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = np.uint64(0xD2511F53) * c0, np.uint64(0xCD9E8D57) * c2
        n0, n1, n2, n3 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & m, p1 & m, ((p0 >> np.uint64(32)) ^ c3 ^ k1) & m, p0 & m
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0, k1 = (k0 + np.uint64(0x9E3779B9)) & m, (k1 + np.uint64(0xBB67AE85)) & m
    return c0, c1, c2, c3


def normals(n_atom, kick, seed):
    """(n_atom, 3) standard normal deviates of kick number `kick` (two kicks per MD step, counted from the seed's setting)"""
    a = np.arange(n_atom, dtype=np.uint64)
    z = np.zeros(n_atom, dtype=np.uint64)
    r = philox4x32_10(a, z + np.uint64(kick & 0xFFFFFFFF), z + np.uint64(kick >> 32), z, seed & 0xFFFFFFFF, seed >> 32)
    u0, u1, u2, u3 = (r[0].astype(np.float64) + 1.0) / 4294967296.0, r[1].astype(np.float64) / 4294967296.0, (r[2].astype(np.float64) + 1.0) / 4294967296.0, r[3].astype(np.float64) / 4294967296.0
    ra, rb = np.sqrt(-2.0 * np.log(u0)), np.sqrt(-2.0 * np.log(u2))
    tp = 6.283185307179586
    return np.stack([ra * np.cos(tp * u1), ra * np.sin(tp * u1), rb * np.cos(tp * u3)], axis=1)


def kick(v, F, mass, dt, friction, temperature, z):
    """one Langevin half kick: v (A/ps), F (kJ/mol/A), mass (g/mol), dt (ps), friction (1/ps)"""
    a = np.exp(-friction * dt / 2.0)
    m = mass[:, None]
    return (a * v + (1.0 - a) / friction * F / m * 100.0
            + np.sqrt(2.0 * BOLTZMANN * temperature * friction * 1000.0) / np.sqrt(m / 1000.0) / 100.0 * np.sqrt((1.0 - np.exp(-friction * dt)) / (2.0 * friction)) * z)
