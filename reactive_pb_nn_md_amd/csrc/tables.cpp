// Host-side construction of the lookup tables the kernels read (init time; reference:
// src/initialize_routines.f90:193-264 initialize_energy_force, src/pme.f90:505-656 B_spline / CB_array / bm_sq /
// update_Ewald_self / periodic_box_change).  The reference's mixed single/double precision is reproduced because
// the tables DEFINE the potential (SURVEY.md 8a "numerics row"): tests compare them bit-for-bit (sha256).
#include "ctx.hpp"

namespace {

// Cox-de Boor recursion kept in float storage, double arithmetic (src/pme.f90:505-532)
double bspline_f32(double u, int n) {
  float m[8][8];
  for (int i = 1; i <= n - 1; ++i) {
    const double ui = u - double(i - 1);
    m[1][i] = (ui < 0.0 || ui > 2.0) ? 0.0f : float(1.0 - std::fabs(ui - 1.0));
  }
  for (int j = 2; j <= n - 1; ++j)
    for (int i = 1; i <= n - j; ++i) {
      const double ui = u - double(i - 1);
      const double lo = (ui / double(j)) * double(m[j - 1][i]);
      const double hi = ((double(j + 1) - ui) / double(j)) * double(m[j - 1][i + 1]);
      m[j][i] = float(lo + hi);
    }
  return double(m[n - 1][1]);
}

// |b(m)|^2 of Essmann eq. 4.4 with the reference's single-precision phase factors (src/pme.f90:578-595)
double b_mod_sq(double pi_ref, int m, int n, int K) {
  double re = 0.0, im = 0.0;
  for (int i = 0; i <= n - 2; ++i) {
    const double ang = 2.0 * pi_ref * double(m * i) / double(K);
    const double b = bspline_f32(double(i + 1), n);
    re = re + b * double(float(std::cos(ang)));
    im = im + b * double(float(std::sin(ang)));
  }
  const double den = re * re + im * im;
  const double br = re / den, bi = -im / den;
  return br * br + bi * bi;
}

double ipow(double a, int b) {  // what `x ** n` lowers to: binary exponentiation
  double r = 1.0;
  bool first = true;
  for (;;) {
    if (b & 1) { r = first ? a : r * a; first = false; }
    b >>= 1;
    if (!b) break;
    a = a * a;
  }
  return r;
}

}  // namespace

double volume32(const double box[3]) {  // `real function volume` (src/general_routines.f90:1936)
  const float v = float(box[0] * (box[1] * box[2] - 0.0 * 0.0) - 0.0 * (0.0 * box[2] - 0.0 * 0.0) + 0.0 * (0.0 * 0.0 - box[1] * 0.0));
  return double(std::fabs(v));
}

void recip_lattice(const double box[3], double vol, double kk[3][3]) {  // src/general_routines.f90:473-490, orthorhombic
  const double a[3] = {box[0], 0, 0}, b[3] = {0, box[1], 0}, cc[3] = {0, 0, box[2]};
  auto cross = [](const double* x, const double* y, double* o) {
    o[0] = x[1] * y[2] - x[2] * y[1];
    o[1] = -x[0] * y[2] + x[2] * y[0];
    o[2] = x[0] * y[1] - x[1] * y[0];
  };
  double ka[3], kb[3], kc[3];
  cross(a, b, kc);
  cross(b, cc, ka);
  cross(cc, a, kb);
  for (int i = 0; i < 3; ++i) { kk[0][i] = ka[i] / vol; kk[1][i] = kb[i] / vol; kk[2][i] = kc[i] / vol; }
}

// everything that depends on the box: CB(K,K,K) and the reciprocal lattice of the scaled coordinates.  Called at set-up and by
// msevb_set_box (periodic_box_change, src/pme.f90:634-664)
int build_box_tables(msevb_ctx* c) {
  const int K = c->K, n = c->order;
  // CB(i,j,l): periodic_box_change uses volume = box(1,1)**3 (cubic assumption of the reference, src/pme.f90:646)
  const double vol = c->box[0] * c->box[0] * c->box[0];
  double kk[3][3];
  recip_lattice(c->box, vol, kk);
  const double pi32 = double(3.14159265f);  // `real*8, parameter :: pi=3.14159265` (src/pme.f90:542)
  std::vector<double> bm(K);
  for (int m = 0; m < K; ++m) bm[m] = b_mod_sq(c->pi, m, n, K);
  c->t_CB.resize(size_t(K) * K * K);
  for (int i = 0; i < K; ++i) {
    const int m1 = i > K / 2 ? i - K : i;
    for (int j = 0; j < K; ++j) {
      const int m2 = j > K / 2 ? j - K : j;
      for (int l = 0; l < K; ++l) {
        const int m3 = l > K / 2 ? l - K : l;
        double mm[3];
        for (int d = 0; d < 3; ++d) mm[d] = double(m1) * kk[0][d] + double(m2) * kk[1][d] + double(m3) * kk[2][d];
        const double mag = (mm[0] * mm[0] + mm[1] * mm[1]) + mm[2] * mm[2];
        c->t_CB[i + size_t(K) * (j + size_t(K) * l)] =
            1.0 / (vol * pi32) * std::exp(-(pi32 * pi32) * mag / (c->alpha * c->alpha)) / mag * bm[i] * bm[j] * bm[l];
      }
    }
  }
  c->t_CB[0] = 0.0;
  // reciprocal lattice used for the scaled coordinates: fp32-rounded volume (src/general_routines.f90:484)
  recip_lattice(c->box, volume32(c->box), c->kk);
  return MSEVB_OK;
}

int build_tables(msevb_ctx* c) {
  const int G = c->spline_grid, E = c->erfc_grid, n = c->order;
  if (n != 6) FAIL(MSEVB_ERR_ARG, "only spline_order = 6 is self-consistent in the reference (src/pme.f90:247 divides by 6)");
  c->t_B6.resize(G);
  c->t_B5.resize(G);
  for (int i = 1; i <= G; ++i) {
    c->t_B6[i - 1] = bspline_f32(6.0 / double(G) * double(i), 6);
    c->t_B5[i - 1] = bspline_f32(5.0 / double(G) * double(i), 5);
  }
  c->t_erfc.resize(E + 1);
  c->t_scale.resize(E + 1);
  c->t_ewE.resize(E + 1);
  c->erfc_dx = c->rc / double(E);
  for (int i = 1; i <= E + 1; ++i) {
    const double r = double(i) * c->erfc_dx, ar = r * c->alpha;
    c->t_erfc[i - 1] = std::erfc(ar) * c->conv;
    c->t_scale[i - 1] = c->t_erfc[i - 1] + ar * 2.0 / c->pi_sqrt * std::exp(-ar * ar) * c->conv;
    c->t_ewE[i - 1] = 2.0 / c->pi_sqrt * std::exp(-ar * ar) * c->conv;      // E = -dT/d(alpha r): scale = erfc + (alpha r) E; what pair_flat.hip expands both tables with
  }
  double fact[13];
  fact[1] = 1.0;
  for (int i = 2; i <= 12; ++i) fact[i] = double(i) * fact[i - 1];
  c->t_tt.resize(4 * TT_GRID);
  c->t_dtt.resize(4 * TT_GRID);
  for (int i = 1; i <= TT_GRID; ++i) {  // Tang_Toennies_damp / dTang_Toennies_damp (src/general_routines.f90:1782-1815)
    const double x = TT_MAX / double(TT_GRID) * double(i);
    for (int q = 0; q < 4; ++q) {
      const int nn = 6 + 2 * q;
      double sum = 1.0, xn = 1.0;
      for (int k = 1; k <= nn; ++k) { xn = xn * x; sum = sum + xn / fact[k]; }
      c->t_tt[(i - 1) * 4 + q] = 1.0 - sum * std::exp(-x);
      c->t_dtt[(i - 1) * 4 + q] = std::exp(-x) * ipow(x, nn) / fact[nn];
    }
  }
  CHK(build_box_tables(c));
  // update_Ewald_self (src/pme.f90:608-625)
  double es = 0.0;
  for (int i = 0; i < c->N; ++i) es = es - c->h_chg[i] * c->h_chg[i];
  es = es * c->alpha / c->pi_sqrt;
  c->ewald_self = es * c->conv;
  return MSEVB_OK;
}
