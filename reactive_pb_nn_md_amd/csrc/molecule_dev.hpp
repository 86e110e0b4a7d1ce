// Device functions for one molecule (bonded terms and intra-molecular non-bonded pairs); shared by molecule.hip and evb.hip.
#pragma once
#include "ctx.hpp"
#include "devmath.hpp"

__device__ __forceinline__ double d3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// Per-thread [atom][3] / [atom] arrays kept in LDS with the thread index fastest (conflict-free, and dynamically indexable: the bonded
// terms address atoms through the molecule-type tables, which forces private arrays into scratch memory).  x[i][d] / q[i] syntax as for
// plain arrays, so the molecule functions below are templates over the array type.
template <int STRIDE>
struct LdsRows {
  double* p;      // base + thread index
  struct Row { double* q; __device__ __forceinline__ double& operator[](int d) const { return q[d * STRIDE]; } };
  __device__ __forceinline__ Row operator[](int i) const { return Row{p + i * 3 * STRIDE}; }
};
template <int STRIDE>
struct LdsVec {
  double* p;
  __device__ __forceinline__ double& operator[](int i) const { return p[i * STRIDE]; }
};

// force accumulation: plain add when one thread owns the molecule, LDS atomic add when the terms of one molecule are spread over lanes
template <bool SHARED>
__device__ __forceinline__ void fadd(double& dst, double v) { if (SHARED) atomicAdd(&dst, v); else dst += v; }

// bond / angle / dihedral energy+force of one molecule whose atom coordinates are x[k][3]; adds into f[k][3]
// (SHARED: the caller's lanes t0, t0 + tstep, ... share the terms, f lives in LDS and the energies are per-lane partial sums)
template <bool SHARED = false, class X, class F>
__device__ __forceinline__ void molecule_bonded(const MolTypeDev& mt, X x, F f, double pi_ref, double& eb, double& ea, double& ed, int t0 = 0, int tstep = 1) {
  eb = ea = ed = 0.0;
  for (int b = t0; b < mt.nb; b += tstep) {   // pairwise_bond_energy_force (src/intra_bonded_interactions.f90:125-168)
    const int i = mt.bond[b][0], j = mt.bond[b][1];
    double r[3] = {x[i][0] - x[j][0], x[i][1] - x[j][1], x[i][2] - x[j][2]};
    const double rm = sqrt(d3(r, r));
    const double* p = mt.bond_par[b];
    double fs = 0.0;
    if (mt.bond_type[b] == 1) { eb += 0.5 * p[1] * (rm - p[0]) * (rm - p[0]); fs = -p[1] * (rm - p[0]) / rm; }
    else if (mt.bond_type[b] == 2) { const double t = rm * rm - p[0] * p[0]; eb += 0.25 * p[1] * t * t; fs = -p[1] * t; }
    else { const double ex = exp(-p[1] * (rm - p[2])); eb += p[0] * (1.0 - ex) * (1.0 - ex); fs = -2.0 * p[0] * p[1] * ex * (1.0 - ex) / rm; }
    for (int d = 0; d < 3; ++d) { fadd<SHARED>(f[i][d], fs * r[d]); fadd<SHARED>(f[j][d], -(fs * r[d])); }
  }
  for (int a = t0; a < mt.na; a += tstep) {   // trimer_angle_energy_force (:219-284)
    const int i = mt.ang[a][0], j = mt.ang[a][1], k = mt.ang[a][2];
    double rij[3], rkj[3];
    for (int d = 0; d < 3; ++d) { rij[d] = x[i][d] - x[j][d]; rkj[d] = x[k][d] - x[j][d]; }
    const double rij_m = sqrt(d3(rij, rij)), rkj_m = sqrt(d3(rkj, rkj));
    const double cosine = d3(rij, rkj) / rij_m / rkj_m;
    const double th0 = mt.ang_par[a][0], cth = mt.ang_par[a][1];
    double fac;
    if (mt.ang_type[a] == 1) {
      double theta;
      if (cosine < -0.999999999) theta = pi_ref; else if (cosine > 0.999999999) theta = 0.0; else theta = acos(cosine);
      ea += 0.5 * cth * (theta - th0) * (theta - th0);
      fac = fabs(theta - th0) < 1e-4 ? 0.0 : cth * (theta - th0) / sqrt(1.0 - cosine * cosine);
    } else {
      const double c0 = cos(th0);
      ea += 0.5 * cth * (cosine - c0) * (cosine - c0);
      fac = -cth * (cosine - c0);
    }
    for (int d = 0; d < 3; ++d) {
      const double fi = fac * (rkj[d] / rij_m / rkj_m - cosine * rij[d] / (rij_m * rij_m));
      const double fk = fac * (rij[d] / rij_m / rkj_m - cosine * rkj[d] / (rkj_m * rkj_m));
      fadd<SHARED>(f[i][d], fi); fadd<SHARED>(f[k][d], fk); fadd<SHARED>(f[j][d], -(fi + fk));
    }
  }
  for (int q = t0; q < mt.nd; q += tstep) {   // quartet_dihedral_energy_force (:343-552), dot-product formulation
    const int i = mt.dih[q][0], j = mt.dih[q][1], k = mt.dih[q][2], l = mt.dih[q][3];
    double rji[3], rkj[3], rlk[3];
    for (int d = 0; d < 3; ++d) { rji[d] = x[j][d] - x[i][d]; rkj[d] = x[k][d] - x[j][d]; rlk[d] = x[l][d] - x[k][d]; }
    const double rji2 = d3(rji, rji), rkj2 = d3(rkj, rkj), rlk2 = d3(rlk, rlk);
    const double dkj_ji = d3(rkj, rji), dlk_kj = d3(rlk, rkj), dlk_ji = d3(rlk, rji);
    const double adb = dkj_ji * dlk_kj - dlk_ji * rkj2, ada = rji2 * rkj2 - dkj_ji * dkj_ji, bdb = rlk2 * rkj2 - dlk_kj * dlk_kj;
    const double sa = sqrt(ada), sb = sqrt(bdb), cosine = adb / sa / sb;
    double xi;
    if (cosine < -0.999999999) xi = pi_ref; else if (cosine > 0.999999999) xi = 0.0; else xi = acos(cosine);
    const double* p = mt.dih_par[q];
    double fac = 0.0, sgn = 1.0;
    if (mt.dih_type[q] == 1) {
      const double c2 = cosine * cosine;
      ed += p[1] * (1.0 + cos(p[2] * xi - p[0]));
      fac = fabs(c2 - 1.0) < 1e-4 ? 0.0 : p[1] * -sin(p[2] * xi - p[0]) * p[2] / sqrt(1.0 - c2);
    } else if (mt.dih_type[q] == 2) {
      if (xi > pi_ref / 2.0) { xi = fabs(xi - pi_ref); sgn = -1.0; }
      ed += 0.5 * p[1] * (xi - p[0]) * (xi - p[0]);
      fac = fabs(xi - p[0]) < 1e-4 ? 0.0 : p[1] * (xi - p[0]) / sqrt(1.0 - cosine * cosine);
    } else {
      const double cc = cosine, c2 = cc * cc, c3 = c2 * cc, c4 = c2 * c2, c5 = c4 * cc;
      ed += p[0] - p[1] * cc + p[2] * c2 - p[3] * c3 + p[4] * c4 - p[5] * c5;
      fac = p[1] - 2.0 * p[2] * cc + 3.0 * p[3] * c2 - 4.0 * p[4] * c3 + 5.0 * p[5] * c4;
    }
    fac *= sgn;
    const double g0 = 1.0 / sa / sb, ga = 0.5 * adb / (ada * sa) / sb, gb = 0.5 * adb / sa / (bdb * sb);
    for (int d = 0; d < 3; ++d) {
      const double dab_ji = rkj[d] * dlk_kj - rlk[d] * rkj2;
      const double dab_kj = rji[d] * dlk_kj + dkj_ji * rlk[d] - dlk_ji * 2.0 * rkj[d];
      const double dab_lk = dkj_ji * rkj[d] - rji[d] * rkj2;
      const double daa_ji = rkj2 * 2.0 * rji[d] - 2.0 * dkj_ji * rkj[d], daa_kj = rji2 * 2.0 * rkj[d] - 2.0 * dkj_ji * rji[d];
      const double dbb_kj = rlk2 * 2.0 * rkj[d] - 2.0 * dlk_kj * rlk[d], dbb_lk = rkj2 * 2.0 * rlk[d] - 2.0 * dlk_kj * rkj[d];
      const double fji = fac * (dab_ji * g0 - ga * daa_ji);
      const double fkj = fac * (dab_kj * g0 - ga * daa_kj - gb * dbb_kj);
      const double flk = fac * (dab_lk * g0 - gb * dbb_lk);
      fadd<SHARED>(f[i][d], -fji); fadd<SHARED>(f[j][d], fji - fkj); fadd<SHARED>(f[k][d], fkj - flk); fadd<SHARED>(f[l][d], flk);
    }
  }
}

// intra-molecular non-bonded pairs of one molecule (no minimum image: molecules are whole)
template <bool SHARED = false, class X, class Q, class F>
__device__ __forceinline__ void molecule_nonbonded(const MolTypeDev& mt, int n, X x, Q q, F f, const double2* __restrict__ etab,
                                   double inv_dx, const double* __restrict__ tt, const double* __restrict__ dtt, double rc2, double alpha,
                                   double erf_factor, double conv, double& e_el, double& e_vdw, int t0 = 0, int tstep = 1) {
  e_el = e_vdw = 0.0;
  int t = 0;
  for (int i = 0; i < n; ++i)
    for (int j = i + 1; j < n; ++j, ++t) {
      if (SHARED && (t % tstep) != t0) continue;
      double dr[3] = {x[i][0] - x[j][0], x[i][1] - x[j][1], x[i][2] - x[j][2]};
      const double r2 = d3(dr, dr), qq = q[i] * q[j];
      double fs = 0.0, e1, f1;
      if (mt.excl[i][j] == 1) { pme_exclusion(r2, qq, alpha, erf_factor, conv, f1, e1); fs += f1; e_el += e1; }
      else {
        if (r2 < rc2) { ewald_interaction(r2, qq, etab, inv_dx, f1, e1); fs += f1; e_el += e1; }
        vdw_interaction(r2, mt.nb_par[i][j], tt, dtt, f1, e1);
        fs += f1; e_vdw += e1;
      }
      for (int d = 0; d < 3; ++d) { fadd<SHARED>(f[i][d], fs * dr[d]); fadd<SHARED>(f[j][d], -(fs * dr[d])); }
    }
}

