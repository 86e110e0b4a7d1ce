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
  const int idx = max((int)ceil(x1), 1);      // r = 0 / NaN (a blown-up trajectory) must not read in front of the table
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
    const int ti = min(max((int)ceil(B * r / TT_MAX * (double)TT_GRID), 1), TT_GRID);
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

// Two-stage form for the pair kernels: ewald_fetch issues the two table-row loads of a pair UNCONDITIONALLY (row index clamped
// into the table, so lanes outside the cutoff load harmless rows) - several pairs' loads are then in flight before any of them
// is consumed; pair_interaction_rows finishes the pair with exactly the arithmetic of pair_interaction.
struct EwaldRows { double r, inv_r, x1; int idx; double2 t0, t1; };
// r = sqrt(r2) and 1/r from ONE v_rsq_f64 seed and two coupled Newton (Goldschmidt) steps, all FMAs: ~10 VALU instructions instead
// of ~45 for an IEEE sqrt plus an IEEE division; both results are within 1-2 ulp (the pair loop is VALU-bound on these).
__device__ __forceinline__ void sqrt_rsqrt(double a, double& g, double& y) {
#pragma clang fp contract(fast)
  const double y0 = __builtin_amdgcn_rsq(a);
  g = a * y0;
  double h = 0.5 * y0;
  double e = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, e, g); h = __builtin_fma(h, e, h);
  e = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, e, g); h = __builtin_fma(h, e, h);
  y = h + h;
}
__device__ __forceinline__ void ewald_fetch(double r2, double inv_dx, int emax, const double2* __restrict__ etab, EwaldRows& w) {
  sqrt_rsqrt(r2, w.r, w.inv_r);
  w.x1 = w.r * inv_dx;
  w.idx = (int)ceil(w.x1);
  const int row = min(max(w.idx, 1), emax);
  w.t0 = etab[row - 1]; w.t1 = etab[row];
}
// Row of the Tang-Toennies table for a pair whose B r sits on a row boundary to within rounding.  The table is a STEP function of B r
// (row = ceil(B r / 50 * 1000), no interpolation: src/pair_int_real_space.f90:674), so the last bit of r picks the row; the kernels' r (v_rsq_f64 seed + Newton
// steps, tile-relative coordinates) differs from the reference's sqrt(dot_product(dr, dr)) in that bit.  Such pairs take the reference's own expression on the
// raw coordinates: tt_arg_exact(r2, B) with r2 from min_image_r2 (pair.hip) / pf_exact_r2 (pair_flat.hip).  NoTieFix: callers whose r IS the reference's.
__device__ __forceinline__ double tt_arg_exact(double r2_exact, double B) {
#pragma clang fp contract(off)
  const double r = sqrt(r2_exact);
  return B * r / TT_MAX * (double)TT_GRID;
}
struct NoTieFix { static constexpr bool on = false; __device__ __forceinline__ double operator()(double) const { return 0.0; } };
template <bool SAPT = true, class TieFix = NoTieFix>
__device__ __forceinline__ void pair_interaction_rows(double r2, double qq, const PairParam& pp, const EwaldRows& w, const double* __restrict__ tt,
                                                      const double* __restrict__ dtt, double& fs, double& e_el, double& e_vdw, const TieFix& tie_fix = TieFix()) {
#pragma clang fp contract(fast)
  const double r = w.r;
  const double inv_r = w.inv_r;
  const double inv_r2 = inv_r * inv_r;
  const double c2 = (w.x1 + 1.0) - (double)w.idx, c1 = 1.0 - c2;
  const double ev = c1 * w.t0.x + c2 * w.t1.x, sv = c1 * w.t0.y + c2 * w.t1.y;
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
    double xi = B * r / TT_MAX * (double)TT_GRID;
    if (TieFix::on) { if (__builtin_expect(fabs(xi - rint(xi)) <= 1e-9 * fmax(xi, 1.0), 0)) xi = tie_fix(B); }      // on a row boundary to within rounding: the reference's arithmetic decides
    const int ti = min(max((int)ceil(xi), 1), TT_GRID);
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
    const int ti = min(max((int)ceil(B * r / TT_MAX * (double)TT_GRID), 1), TT_GRID);
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
  const int idx = max((int)ceil(x1), 1);      // r = 0 / NaN (a blown-up trajectory) must not read in front of the table
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
