// Device functions shared by the pair kernel, the intra-molecular kernel and the EVB delta kernels.
#pragma once
#include "ctx.hpp"

// One pair inside the cutoff: returns fs with f_ij = fs * dr (force on i), electrostatic and vdW energies.
//   Ewald real space: linear interpolation of {erfc, ewaldscale} tables at x = r/dx, row = ceil(x)
//     (src/pair_int_real_space.f90:740-759); E = q_i q_j / r * erfc*, f = q_i q_j / r^3 * scale* * dr (:718-726)
//   LJ: E = C12/r^12 - C6/r^6 ; f = dr/r^2 (12 C12/r^12 - 6 C6/r^6) (:621-645)
//   SAPT: E = A exp(-B r) - sum_n f_n(B r) C_n / r^n, f_n from the Tang-Toennies table row ceil(B r/50*1000),
//     no interpolation (:651-690)
template <bool SAPT = true>
__device__ __forceinline__ void pair_interaction(double r2, double qq, const PairParam& pp, const double2* __restrict__ etab, double inv_dx,
                                                 const double* __restrict__ tt, const double* __restrict__ dtt, double& fs, double& e_el,
                                                 double& e_vdw) {
  const double r = sqrt(r2);
  const double inv_r2 = 1.0 / r2;
  const double inv_r = r * inv_r2;
  const double x1 = r * inv_dx;
  const int idx = (int)ceil(x1);
  const double c2 = (x1 + 1.0) - (double)idx, c1 = 1.0 - c2;
  const double2 t0 = etab[idx - 1], t1 = etab[idx];
  const double ev = c1 * t0.x + c2 * t1.x, sv = c1 * t0.y + c2 * t1.y;
  const double qr = qq * inv_r;
  e_el = qr * ev;
  fs = qr * inv_r2 * sv;
  e_vdw = 0.0;
  if (pp.type == 0) {
    const double r6i = inv_r2 * inv_r2 * inv_r2, r12i = r6i * r6i;
    e_vdw = pp.p[0] * r12i - pp.p[1] * r6i;
    fs += inv_r2 * (12.0 * pp.p[0] * r12i - 6.0 * pp.p[1] * r6i);
  } else if (SAPT && pp.type == 1 && pp.nonzero) {
    const double A = pp.p[0], B = pp.p[1];
    const int ti = (int)ceil(B * r / TT_MAX * (double)TT_GRID);
    const double* t = tt + 4 * (ti - 1);
    const double* dt = dtt + 4 * (ti - 1);
    const double r6i = inv_r2 * inv_r2 * inv_r2, r8i = r6i * inv_r2, r10i = r8i * inv_r2, r12i = r10i * inv_r2;
    const double ex = exp(-B * r);
    e_vdw = A * ex - t[0] * pp.p[2] * r6i - t[1] * pp.p[3] * r8i - t[2] * pp.p[4] * r10i - t[3] * pp.p[5] * r12i;
    const double fac = r * A * B * ex + (r * B * dt[0] - 6.0 * t[0]) * pp.p[2] * r6i + (r * B * dt[1] - 8.0 * t[1]) * pp.p[3] * r8i +
                       (r * B * dt[2] - 10.0 * t[2]) * pp.p[4] * r10i + (r * B * dt[3] - 12.0 * t[3]) * pp.p[5] * r12i;
    fs += inv_r2 * fac;
  }
}

// vdW-only part (intra-molecular non-bonded pairs have no cutoff on the vdW term, src/pair_int_real_space.f90:535-544)
__device__ __forceinline__ void vdw_interaction(double r2, const PairParam& pp, const double* __restrict__ tt, const double* __restrict__ dtt,
                                                double& fs, double& e_vdw) {
  fs = 0.0; e_vdw = 0.0;
  const double inv_r2 = 1.0 / r2;
  if (pp.type == 0) {
    const double r6i = inv_r2 * inv_r2 * inv_r2, r12i = r6i * r6i;
    e_vdw = pp.p[0] * r12i - pp.p[1] * r6i;
    fs = inv_r2 * (12.0 * pp.p[0] * r12i - 6.0 * pp.p[1] * r6i);
  } else if (pp.type == 1 && pp.nonzero) {
    const double r = sqrt(r2), A = pp.p[0], B = pp.p[1];
    const int ti = (int)ceil(B * r / TT_MAX * (double)TT_GRID);
    const double* t = tt + 4 * (ti - 1);
    const double* dt = dtt + 4 * (ti - 1);
    const double r6i = inv_r2 * inv_r2 * inv_r2, r8i = r6i * inv_r2, r10i = r8i * inv_r2, r12i = r10i * inv_r2;
    const double ex = exp(-B * r);
    e_vdw = A * ex - t[0] * pp.p[2] * r6i - t[1] * pp.p[3] * r8i - t[2] * pp.p[4] * r10i - t[3] * pp.p[5] * r12i;
    const double fac = r * A * B * ex + (r * B * dt[0] - 6.0 * t[0]) * pp.p[2] * r6i + (r * B * dt[1] - 8.0 * t[1]) * pp.p[3] * r8i +
                       (r * B * dt[2] - 10.0 * t[2]) * pp.p[4] * r10i + (r * B * dt[3] - 12.0 * t[3]) * pp.p[5] * r12i;
    fs = inv_r2 * fac;
  }
}

// Ewald-only part of a pair inside the cutoff
__device__ __forceinline__ void ewald_interaction(double r2, double qq, const double2* __restrict__ etab, double inv_dx, double& fs, double& e_el) {
  const double r = sqrt(r2), inv_r2 = 1.0 / r2, inv_r = r * inv_r2;
  const double x1 = r * inv_dx;
  const int idx = (int)ceil(x1);
  const double c2 = (x1 + 1.0) - (double)idx, c1 = 1.0 - c2;
  const double2 t0 = etab[idx - 1], t1 = etab[idx];
  const double qr = qq * inv_r;
  e_el = qr * (c1 * t0.x + c2 * t1.x);
  fs = qr * inv_r2 * (c1 * t0.y + c2 * t1.y);
}

// intra_pme_exclusion (src/pair_int_real_space.f90:781-816): -q_i q_j erf(alpha r)/r for excluded pairs
__device__ __forceinline__ void pme_exclusion(double r2, double qq, double alpha, double erf_factor, double conv, double& fs, double& e_el) {
  const double r = sqrt(r2);
  if (r < 1e-8) { e_el = -erf_factor * qq * conv; fs = 0.0; return; }
  const double t = r * alpha, ec = erfc(t) - 1.0;
  e_el = qq * ec / r * conv;
  fs = qq * (ec / (r2 * r) + erf_factor * exp(-(t * t)) / r2) * conv;
}
