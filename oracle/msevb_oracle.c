/*
 * TEST INFRASTRUCTURE ONLY (oracle/) - see msevb_oracle.h.
 * CPU restatement of the reference's per-timestep force/energy path, written from the
 * algorithm (operation order kept so that results agree with the reference's golden
 * vectors to rounding).  Build: gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp.
 * Part 1 (this file): tables, Verlet list, pair loop, intra-molecular non-bonded, PME,
 * bonded terms, total force call, NVE integrator.  Part 2: msevb_oracle_evb.c (MS-EVB).
 */
#include "msevb_oracle_internal.h"

/* ------------------------------------------------------------------ small helpers */
static double powi(double a, int b) { /* integer power by binary exponentiation (what x**n compiles to) */
  double r = 1.0; int first = 1;
  for (;;) { if (b & 1) { r = first ? a : r * a; first = 0; } b >>= 1; if (!b) break; a = a * a; }
  return r;
}
static inline double dot3(const double* a, const double* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

OrcSys* orc_create(int N, int M, int nt, int nmt, int K, const double box[3], double rc, double rv, double alpha) {
  OrcSys* s = (OrcSys*)calloc(1, sizeof(OrcSys));
  s->N = N; s->M = M; s->nt = nt; s->nmt = nmt; s->K = K; s->order = 6;
  s->spline_grid = 100000; s->erfc_grid = 100000;
  for (int i = 0; i < 3; ++i) s->box[i] = box[i];
  s->rc = rc; s->rv = rv; s->alpha = alpha; s->nthreads = 1;
  s->pi = 3.141592654; s->pi_sqrt = 1.772453851;            /* src/glob_v.f90:386-387 */
  s->conv = (double)1389.35465f;                              /* src/glob_v.f90:389 (fp32 literal) */
  s->atype = (int*)calloc(N, sizeof(int)); s->chg = (double*)calloc(N, sizeof(double));
  s->mass = (double*)calloc(N, sizeof(double)); s->xyz = (double*)calloc(3 * (size_t)N, sizeof(double));
  s->mol_of = (int*)calloc(N, sizeof(int));
  s->mol_type = (int*)calloc(M, sizeof(int)); s->mol_natom = (int*)calloc(M, sizeof(int));
  s->mol_atoms = (int*)calloc((size_t)M * ORC_MAX_MOL_ATOMS, sizeof(int));
  s->r_com = (double*)calloc(3 * (size_t)M, sizeof(double));
  s->vdw_type = (int*)calloc(nt * nt, sizeof(int)); s->vdw_par = (double*)calloc(nt * nt * 6, sizeof(double));
  s->vdw_par14 = (double*)calloc(nt * nt * 6, sizeof(double));
  s->bond_type = (int*)calloc(nt * nt, sizeof(int)); s->bond_par = (double*)calloc(nt * nt * 3, sizeof(double));
  size_t K3 = (size_t)K * K * K;
  s->Q = (double*)calloc(K3, sizeof(double)); s->theta = (double*)calloc(K3, sizeof(double));
  s->CB = (double*)calloc(K3, sizeof(double));
  s->xyz_scale = (double*)calloc(3 * (size_t)N, sizeof(double)); s->force_recip = (double*)calloc(3 * (size_t)N, sizeof(double));
  s->vstore = (double*)calloc(3 * (size_t)N, sizeof(double)); s->vdisp = (double*)calloc(3 * (size_t)N, sizeof(double));
  s->verlet_point = (int*)calloc(N + 1, sizeof(int));
  s->hydronium = -1; s->max_chain = 3;
  return s;
}
void orc_destroy(OrcSys* s) {
  if (!s) return;
  free(s->atype); free(s->chg); free(s->mass); free(s->xyz); free(s->mol_of); free(s->mol_type); free(s->mol_natom);
  free(s->mol_atoms); free(s->r_com); free(s->vdw_type); free(s->vdw_par); free(s->vdw_par14); free(s->bond_type);
  free(s->bond_par); free(s->ang_idx); free(s->ang_par); free(s->dih_idx); free(s->dih_par); free(s->Q); free(s->theta);
  free(s->CB); free(s->xyz_scale); free(s->force_recip); free(s->vstore); free(s->vdisp); free(s->verlet_point);
  free(s->neighbor_list); free(s->B6); free(s->B5); free(s->erfc_t); free(s->scale_t);
  free(s->evb_Fdiag); free(s->evb_Foff);
  free(s);
}
void orc_set_threads(OrcSys* s, int n) { s->nthreads = n < 1 ? 1 : n; }

void orc_set_atoms(OrcSys* s, const int* atype, const double* chg, const double* mass, const int* mol_first,
                   const int* mol_natom, const int* mol_type) {
  memcpy(s->atype, atype, s->N * sizeof(int)); memcpy(s->chg, chg, s->N * sizeof(double));
  memcpy(s->mass, mass, s->N * sizeof(double));
  for (int m = 0; m < s->M; ++m) {
    s->mol_type[m] = mol_type[m]; s->mol_natom[m] = mol_natom[m];
    for (int k = 0; k < mol_natom[m]; ++k) { s->mol_atoms[m * ORC_MAX_MOL_ATOMS + k] = mol_first[m] + k; s->mol_of[mol_first[m] + k] = m; }
  }
}
void orc_set_vdw(OrcSys* s, const int* t, const double* p, const double* p14) {
  memcpy(s->vdw_type, t, s->nt * s->nt * sizeof(int)); memcpy(s->vdw_par, p, s->nt * s->nt * 6 * sizeof(double));
  memcpy(s->vdw_par14, p14, s->nt * s->nt * 6 * sizeof(double));
}
void orc_set_bonded(OrcSys* s, const int* bt, const double* bp, int n_ang, const int* ai, const double* ap, int n_dih,
                    const int* di, const double* dp) {
  memcpy(s->bond_type, bt, s->nt * s->nt * sizeof(int)); memcpy(s->bond_par, bp, s->nt * s->nt * 3 * sizeof(double));
  s->n_ang = n_ang; s->ang_idx = (int*)malloc((n_ang + 1) * 4 * sizeof(int)); s->ang_par = (double*)malloc((n_ang + 1) * 2 * sizeof(double));
  memcpy(s->ang_idx, ai, n_ang * 4 * sizeof(int)); memcpy(s->ang_par, ap, n_ang * 2 * sizeof(double));
  s->n_dih = n_dih; s->dih_idx = (int*)malloc((n_dih + 1) * 5 * sizeof(int)); s->dih_par = (double*)malloc((n_dih + 1) * 6 * sizeof(double));
  memcpy(s->dih_idx, di, n_dih * 5 * sizeof(int)); memcpy(s->dih_par, dp, n_dih * 6 * sizeof(double));
}
void orc_set_moltype(OrcSys* s, int it, int natom, const int* atype, int nb, const int* bonds, int na, const int* angles,
                     int nd, const int* dihedrals, const int* excl, const int* rp, const int* rb) {
  OrcMolType* t = &s->mt[it];
  memset(t, 0, sizeof(*t));
  t->natom = natom; t->nb = nb; t->na = na; t->nd = nd;
  memcpy(t->atype, atype, natom * sizeof(int));
  memcpy(t->bonds, bonds, nb * 2 * sizeof(int)); memcpy(t->angles, angles, na * 3 * sizeof(int));
  memcpy(t->dihedrals, dihedrals, nd * 4 * sizeof(int));
  for (int i = 0; i < natom; ++i) for (int j = 0; j < natom; ++j) t->excl[i][j] = excl[i * natom + j];
  if (rp) memcpy(t->reactive_protons, rp, natom * sizeof(int));
  if (rb) memcpy(t->reactive_basic, rb, natom * sizeof(int));
}
void orc_set_xyz(OrcSys* s, const double* xyz) { memcpy(s->xyz, xyz, 3 * (size_t)s->N * sizeof(double)); }
void orc_get_xyz(OrcSys* s, double* xyz) { memcpy(xyz, s->xyz, 3 * (size_t)s->N * sizeof(double)); }

/* ------------------------------------------------------------------ tables */
/* B_spline: src/pme.f90:505-532 - Cox-de Boor recursion held in a single-precision work array */
double orc_bspline(double u, int n) {
  float mn[8][8];
  for (int i = 1; i <= n - 1; ++i) {
    double ui = u - (double)(i - 1);
    if (ui < 0.0 || ui > 2.0) mn[1][i] = 0.0f; else mn[1][i] = (float)(1.0 - fabs(ui - 1.0));
  }
  for (int j = 2; j <= n - 1; ++j)
    for (int i = 1; i <= n - j; ++i) {
      double ui = u - (double)(i - 1);
      mn[j][i] = (float)((ui / (double)j) * (double)mn[j - 1][i] + (((double)(j + 1) - ui) / (double)j) * (double)mn[j - 1][i + 1]);
    }
  return (double)mn[n - 1][1];
}
/* bm_sq: src/pme.f90:578-595 - exp(2 pi i m k/K) rounded to single-precision complex */
static double bm_sq(const OrcSys* s, int m, int n, int K) {
  double sr = 0.0, si = 0.0;
  for (int i = 0; i <= n - 2; ++i) {
    double tmp = 2.0 * s->pi * (double)(m * i) / (double)K;
    double b = orc_bspline((double)(i + 1), n);
    float c = (float)cos(tmp), sn = (float)sin(tmp);
    sr = sr + b * (double)c; si = si + b * (double)sn;
  }
  /* bm = 1/sum (complex), bm_sq = |bm|^2 */
  double den = sr * sr + si * si;
  double br = sr / den, bi = -si / den;
  return br * br + bi * bi;
}
/* volume(): src/general_routines.f90:1936 - a REAL (fp32) function */
static double volume32(const double box[3]) {
  double a[3] = {box[0], 0, 0}, b[3] = {0, box[1], 0}, c[3] = {0, 0, box[2]};
  float v = (float)(a[0] * (b[1] * c[2] - b[2] * c[1]) - a[1] * (b[0] * c[2] - b[2] * c[0]) + a[2] * (b[0] * c[1] - b[1] * c[0]));
  return (double)fabsf(v);
}
static void cross(const double* a, const double* b, double* o) {
  o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = -a[0] * b[2] + a[2] * b[0]; o[2] = a[0] * b[1] - a[1] * b[0];
}
/* reciprocal lattice rows kk(1,:)=b x c / vol ... (src/general_routines.f90:473-490) */
void orc_recip_lattice(const double box[3], double vol, double kk[3][3]) {
  double a[3] = {box[0], 0, 0}, b[3] = {0, box[1], 0}, c[3] = {0, 0, box[2]}, ka[3], kb[3], kc[3];
  cross(a, b, kc); cross(b, c, ka); cross(c, a, kb);
  for (int i = 0; i < 3; ++i) { kk[0][i] = ka[i] / vol; kk[1][i] = kb[i] / vol; kk[2][i] = kc[i] / vol; }
}
double orc_ewald_self(OrcSys* s) { /* src/pme.f90:608-625 */
  double e = 0.0;
  for (int i = 0; i < s->N; ++i) e = e - s->chg[i] * s->chg[i];
  e = e * s->alpha / s->pi_sqrt;
  e = e * s->conv;
  return e;
}
void orc_build_tables(OrcSys* s) {
  int G = s->spline_grid, K = s->K, n = s->order;
  free(s->B6); free(s->B5); free(s->erfc_t); free(s->scale_t);
  s->B6 = (double*)malloc(G * sizeof(double)); s->B5 = (double*)malloc(G * sizeof(double));
  for (int i = 1; i <= G; ++i) {      /* src/initialize_routines.f90:214-217 */
    s->B6[i - 1] = orc_bspline(6.0 / (double)G * (double)i, 6);
    s->B5[i - 1] = orc_bspline(5.0 / (double)G * (double)i, 5);
  }
  int E = s->erfc_grid;
  s->erfc_t = (double*)malloc((E + 1) * sizeof(double)); s->scale_t = (double*)malloc((E + 1) * sizeof(double));
  s->erfc_dx = s->rc / (double)E;
  for (int i = 1; i <= E + 1; ++i) {  /* src/initialize_routines.f90:237-242 */
    double r = (double)i * s->erfc_dx, ar = r * s->alpha;
    s->erfc_t[i - 1] = erfc(ar) * s->conv;
    s->scale_t[i - 1] = s->erfc_t[i - 1] + ar * 2.0 / s->pi_sqrt * exp(-ar * ar) * s->conv;
  }
  double fact[13]; fact[1] = 1.0;
  for (int i = 2; i <= 12; ++i) fact[i] = (double)i * fact[i - 1];
  for (int i = 1; i <= ORC_TT_GRID; ++i) {  /* src/initialize_routines.f90:250-263, general_routines.f90:1782-1815 */
    double x = ORC_TT_MAX / (double)ORC_TT_GRID * (double)i;
    for (int q = 0; q < 4; ++q) {
      int nn = 6 + 2 * q; double sum = 1.0, xn = 1.0;
      for (int k = 1; k <= nn; ++k) { xn = xn * x; sum = sum + xn / fact[k]; }
      s->tt[i - 1][q] = 1.0 - sum * exp(-x);
      s->dtt[i - 1][q] = exp(-x) * powi(x, nn) / fact[nn];
    }
  }
  /* CB: periodic_box_change + CB_array (src/pme.f90:634-656, :537-573); volume = box(1,1)**3 here */
  double vol = s->box[0] * s->box[0] * s->box[0], kk[3][3];
  orc_recip_lattice(s->box, vol, kk);
  const double pi32 = (double)3.14159265f;
  double* bm = (double*)malloc(K * sizeof(double));
  for (int m = 0; m < K; ++m) bm[m] = bm_sq(s, m, n, K);
  for (int i = 0; i < K; ++i) { int m1 = i > K / 2 ? i - K : i;
    for (int j = 0; j < K; ++j) { int m2 = j > K / 2 ? j - K : j;
      for (int l = 0; l < K; ++l) { int m3 = l > K / 2 ? l - K : l;
        double mm[3];
        for (int d = 0; d < 3; ++d) mm[d] = (double)m1 * kk[0][d] + (double)m2 * kk[1][d] + (double)m3 * kk[2][d];
        double mag = dot3(mm, mm);
        double v = 1.0 / (vol * pi32) * exp(-(pi32 * pi32) * mag / (s->alpha * s->alpha)) / mag * bm[i] * bm[j] * bm[l];
        s->CB[i + (size_t)K * (j + (size_t)K * l)] = v;
      } } }
  s->CB[0] = 0.0;
  free(bm);
  s->ewald_self = orc_ewald_self(s);
}
int orc_get_table(OrcSys* s, const char* name, double* out, int n) {
  const double* src = NULL; int len = 0;
  if (!strcmp(name, "B6")) { src = s->B6; len = s->spline_grid; }
  else if (!strcmp(name, "B5")) { src = s->B5; len = s->spline_grid; }
  else if (!strcmp(name, "erfc")) { src = s->erfc_t; len = s->erfc_grid + 1; }
  else if (!strcmp(name, "ewaldscale")) { src = s->scale_t; len = s->erfc_grid + 1; }
  else if (!strcmp(name, "tt")) { src = &s->tt[0][0]; len = 4 * ORC_TT_GRID; }
  else if (!strcmp(name, "dtt")) { src = &s->dtt[0][0]; len = 4 * ORC_TT_GRID; }
  else if (!strcmp(name, "CB")) { src = s->CB; len = s->K * s->K * s->K; }
  if (src && out) memcpy(out, src, (n < len ? n : len) * sizeof(double));
  return len;
}

/* ------------------------------------------------------------------ Verlet list */
/* construct_verlet_list_grid: src/general_routines.f90:1408-1595.  Same linked-cell insertion order and the
   same (ia,ib,ic) traversal, so neighbor_list comes out in the reference's order. */
int orc_verlet_build_topo(OrcSys* s, const OrcTopo* tp, int ns, int* verlet_point, int** list_out) {
  int N = s->N, nx = ns, ny = ns, nz = ns;
  int ncell = nx * ny * nz;
  int *head = (int*)calloc(ncell, sizeof(int)), *end = (int*)calloc(ncell, sizeof(int)), *next = (int*)calloc(N + 1, sizeof(int));
  int* cellof = (int*)malloc(3 * (size_t)N * sizeof(int));
  double inv[3] = {1.0 / s->box[0], 1.0 / s->box[1], 1.0 / s->box[2]};
  for (int a = 0; a < N; ++a) {            /* atoms in (molecule, atom) order == index order */
    int c[3], nn[3] = {nx, ny, nz};
    for (int d = 0; d < 3; ++d) {
      double rt = inv[d] * tp->xyz[3 * a + d];
      int ix = (int)floor(rt * nn[d]) + 1;
      ix = ix - (int)floor((double)(ix - 1) / nn[d]) * nn[d];
      c[d] = ix;
    }
    cellof[3 * a] = c[0]; cellof[3 * a + 1] = c[1]; cellof[3 * a + 2] = c[2];
    int ci = (c[0] - 1) + nx * ((c[1] - 1) + ny * (c[2] - 1));
    if (head[ci] == 0) { head[ci] = a + 1; end[ci] = a + 1; }
    else { next[end[ci]] = a + 1; end[ci] = a + 1; }
  }
  double rv2 = s->rv * s->rv;
  int dia = (int)floor(s->rv * inv[0] * (double)nx) + 1, dib = (int)floor(s->rv * inv[1] * (double)ny) + 1,
      dic = (int)floor(s->rv * inv[2] * (double)nz) + 1;
  size_t cap = (size_t)N * 64 + 1024, cnt = 0;
  int* list = (int*)malloc(cap * sizeof(int));
  for (int a = 0; a < N; ++a) {
    verlet_point[a] = (int)cnt;
    int ix = cellof[3 * a], iy = cellof[3 * a + 1], iz = cellof[3 * a + 2], im = tp->mol_of[a];
    for (int ia = -dia; ia <= dia; ++ia) {
      int g1 = ix + ia; g1 = g1 - (int)floor((double)(g1 - 1) / (double)nx) * nx;
      for (int ib = -dib; ib <= dib; ++ib) {
        int g2 = iy + ib; g2 = g2 - (int)floor((double)(g2 - 1) / (double)ny) * ny;
        for (int ic = -dic; ic <= dic; ++ic) {
          int g3 = iz + ic; g3 = g3 - (int)floor((double)(g3 - 1) / (double)nz) * nz;
          int j = head[(g1 - 1) + nx * ((g2 - 1) + ny * (g3 - 1))];
          while (j != 0) {
            int jb = j - 1;
            if (a < jb && tp->mol_of[jb] != im) {
              double r[3];
              for (int d = 0; d < 3; ++d) {
                r[d] = tp->xyz[3 * a + d] - tp->xyz[3 * jb + d];
                double sh = s->box[d] * floor(r[d] / s->box[d] + 0.5);
                r[d] = r[d] - sh;
              }
              if (dot3(r, r) < rv2) {
                if (cnt >= cap) { cap *= 2; list = (int*)realloc(list, cap * sizeof(int)); }
                list[cnt++] = jb;
              }
            }
            j = next[j];
          }
        } } }
  }
  verlet_point[N] = (int)cnt;
  *list_out = list;
  free(head); free(end); free(next); free(cellof);
  return (int)cnt;
}
int orc_verlet_build(OrcSys* s, int ns) {
  OrcTopo t = orc_principal_topo(s);
  s->nslist = ns;
  free(s->neighbor_list); s->neighbor_list = NULL;
  s->n_neighbors = orc_verlet_build_topo(s, &t, ns, s->verlet_point, &s->neighbor_list);
  return s->n_neighbors;
}
void orc_verlet_get(OrcSys* s, int* vp, int* nl) {
  memcpy(vp, s->verlet_point, (s->N + 1) * sizeof(int));
  if (nl) memcpy(nl, s->neighbor_list, (size_t)s->n_neighbors * sizeof(int));
}
/* update_verlet_displacements: src/general_routines.f90:1259-1337 */
int orc_verlet_check(OrcSys* s, int initialize) {
  int N = s->N;
  if (initialize) { memcpy(s->vstore, s->xyz, 3 * (size_t)N * sizeof(double)); memset(s->vdisp, 0, 3 * (size_t)N * sizeof(double)); return 0; }
  double d1 = 0, d2 = 0;
  for (int i = 0; i < N; ++i) {
    double dr[3];
    for (int d = 0; d < 3; ++d) {
      double dc = s->xyz[3 * i + d] - s->vstore[3 * i + d];
      double sh = floor(dc * (1.0 / s->box[d]) + 0.5) * s->box[d];
      dr[d] = s->xyz[3 * i + d] - s->vstore[3 * i + d] - sh;
      s->vdisp[3 * i + d] += dr[d];
    }
    double nrm = sqrt(dot3(&s->vdisp[3 * i], &s->vdisp[3 * i]));
    if (nrm > d2) { d2 = nrm; if (d2 > d1) { double t = d1; d1 = d2; d2 = t; } }
  }
  memcpy(s->vstore, s->xyz, 3 * (size_t)N * sizeof(double));
  double skin = 1.2 * (s->rv - s->rc);       /* verlet_thresh = 1.2d0, src/glob_v.f90:394 */
  return (d1 + d2) > skin ? 1 : 0;
}

/* ------------------------------------------------------------------ pair kernels */
/* pairwise_real_space_ewald + linear_interpolation_ewald_tables: src/pair_int_real_space.f90:698-759 */
void orc_k_ewald(const OrcSys* s, int n, const double* dr, const double* dr2, const double* qq, double* E, double* f) {
  double e = 0.0;
  for (int k = 0; k < n; ++k) {
    double r = sqrt(dr2[k]), r3 = dr2[k] * r;
    double x1 = r / s->erfc_dx; int idx = (int)ceil(x1);
    double c2 = (x1 + 1.0) - (double)idx, c1 = 1.0 - c2;
    double ev = c1 * s->erfc_t[idx - 1] + c2 * s->erfc_t[idx];
    double sv = c1 * s->scale_t[idx - 1] + c2 * s->scale_t[idx];
    e = e + qq[k] / r * ev;
    for (int d = 0; d < 3; ++d) f[3 * k + d] = f[3 * k + d] + qq[k] / r3 * dr[3 * k + d] * sv;
  }
  *E = e;
}
/* pairwise_real_space_LJ: src/pair_int_real_space.f90:621-645 */
void orc_k_lj(int n, const double* dr, const double* dr2, const double* par, double* E, double* f) {
  double e = 0.0;
  for (int k = 0; k < n; ++k) {
    double dr6 = (dr2[k] * dr2[k]) * dr2[k], dr12 = dr6 * dr6;
    e = e + (par[6 * k] / dr12 - par[6 * k + 1] / dr6);
    double fac = 12.0 * par[6 * k] / dr12 - 6.0 * par[6 * k + 1] / dr6;
    for (int d = 0; d < 3; ++d) f[3 * k + d] = f[3 * k + d] + dr[3 * k + d] / dr2[k] * fac;
  }
  *E = e;
}
/* pairwise_real_space_sapt: src/pair_int_real_space.f90:651-690 */
void orc_k_sapt(const OrcSys* s, int n, const double* dr, const double* dr2, const double* par, double* E, double* f) {
  double e = 0.0;
  for (int k = 0; k < n; ++k) {
    const double* p = &par[6 * k];
    double r1 = sqrt(dr2[k]), r6 = (dr2[k] * dr2[k]) * dr2[k], r8 = r6 * dr2[k], r10 = r8 * dr2[k], r12 = r10 * dr2[k];
    int idx = (int)ceil(p[1] * r1 / ORC_TT_MAX * (double)ORC_TT_GRID);
    const double* tt = s->tt[idx - 1]; double dtt[4];
    for (int q = 0; q < 4; ++q) dtt[q] = p[1] * s->dtt[idx - 1][q];
    double ex = exp(-1 * p[1] * r1);
    e = e + (p[0] * ex - tt[0] * p[2] / r6 - tt[1] * p[3] / r8 - tt[2] * p[4] / r10 - tt[3] * p[5] / r12);
    double fac = r1 * p[0] * p[1] * ex + r1 * dtt[0] * p[2] / r6 - tt[0] * 6.0 * p[2] / r6 + r1 * dtt[1] * p[3] / r8 -
                 tt[1] * 8.0 * p[3] / r8 + r1 * dtt[2] * p[4] / r10 - tt[2] * 10.0 * p[4] / r10 + r1 * dtt[3] * p[5] / r12 -
                 tt[3] * 12.0 * p[5] / r12;
    for (int d = 0; d < 3; ++d) f[3 * k + d] = f[3 * k + d] + dr[3 * k + d] / dr2[k] * fac;
  }
  *E = e;
}
/* intra_pme_exclusion: src/pair_int_real_space.f90:781-816 */
static void k_pme_excl(const OrcSys* s, int n, const double* dr, const double* dr2, const double* qq, double* E, double* f) {
  double erf_factor = 2.0 * s->alpha / s->pi_sqrt, e = 0.0;
  for (int k = 0; k < n; ++k) {
    double r = sqrt(dr2[k]), r3 = dr2[k] * r;
    if (r < 1e-8) e = e - erf_factor * qq[k] * s->conv;
    else {
      double t = r * s->alpha, ec = erfc(t) - 1.0;
      e = e + qq[k] * ec / r * s->conv;
      double g = ec / r3 + erf_factor * exp(-(t * t)) / dr2[k];
      for (int d = 0; d < 3; ++d) f[3 * k + d] = f[3 * k + d] + qq[k] * dr[3 * k + d] * g * s->conv;
    }
  }
  *E = e;
}

/* one central atom i against a list of candidate atoms: min-image, cutoff split, three kernels, force scatter.
   This is the body of pairwise_real_space_verlet's i-loop (src/pair_int_real_space.f90:190-315) and, with
   screen != NULL and the list = all atoms, of ms_evb_diabat_force_energy_update_real_space (src/ms_evb.f90:1629-1732). */
void orc_atom_vs_list(const OrcSys* s, const OrcTopo* tp, int i, int nl, const int* list, const unsigned char* screen,
                      double* E_elec, double* E_vdw, double* F, OrcScratch* w) {
  if (nl <= 0) return;
  orc_scratch_reserve(w, nl);
  double rc2 = s->rc * s->rc;
  const double* xi = &tp->xyz[3 * i];
  int ti = tp->atype[i]; double qi = tp->chg[i];
  int nc = 0, nlj = 0, nsp = 0;
  for (int k = 0; k < nl; ++k) {
    int j = list ? list[k] : k;
    if (screen && !screen[j]) continue;
    double dr[3];
    for (int d = 0; d < 3; ++d) { dr[d] = xi[d] - tp->xyz[3 * j + d]; dr[d] = dr[d] - s->box[d] * floor(dr[d] / s->box[d] + 0.5); }
    double r2 = dot3(dr, dr);
    if (r2 < rc2) {
      int tj = tp->atype[j], vt = s->vdw_type[ti * s->nt + tj];
      memcpy(&w->dr[3 * nc], dr, 24); w->dr2[nc] = r2; w->qq[nc] = qi * tp->chg[j]; w->idx[nc] = j; nc++;
      if (vt == 0) { memcpy(&w->dr_lj[3 * nlj], dr, 24); w->dr2_lj[nlj] = r2; memcpy(&w->par_lj[6 * nlj], &s->vdw_par[(ti * s->nt + tj) * 6], 48); w->idx_lj[nlj] = j; nlj++; }
      else if (vt == 1) { memcpy(&w->dr_sp[3 * nsp], dr, 24); w->dr2_sp[nsp] = r2; memcpy(&w->par_sp[6 * nsp], &s->vdw_par[(ti * s->nt + tj) * 6], 48); w->idx_sp[nsp] = j; nsp++; }
    }
  }
  double e_el = 0, e_lj = 0, e_sp = 0;
  memset(w->f, 0, 24 * (size_t)nc); memset(w->f_lj, 0, 24 * (size_t)nlj); memset(w->f_sp, 0, 24 * (size_t)nsp);
  if (nc > 0) orc_k_ewald(s, nc, w->dr, w->dr2, w->qq, &e_el, w->f);
  if (nlj > 0) orc_k_lj(nlj, w->dr_lj, w->dr2_lj, w->par_lj, &e_lj, w->f_lj);
  if (nsp > 0) orc_k_sapt(s, nsp, w->dr_sp, w->dr2_sp, w->par_sp, &e_sp, w->f_sp);
  *E_elec = *E_elec + e_el; *E_vdw = *E_vdw + e_lj; *E_vdw = *E_vdw + e_sp;
  for (int k = 0; k < nlj; ++k) for (int d = 0; d < 3; ++d) { F[3 * i + d] += w->f_lj[3 * k + d]; F[3 * w->idx_lj[k] + d] -= w->f_lj[3 * k + d]; }
  for (int k = 0; k < nsp; ++k) for (int d = 0; d < 3; ++d) { F[3 * i + d] += w->f_sp[3 * k + d]; F[3 * w->idx_sp[k] + d] -= w->f_sp[3 * k + d]; }
  for (int k = 0; k < nc; ++k) for (int d = 0; d < 3; ++d) { F[3 * i + d] += w->f[3 * k + d]; F[3 * w->idx[k] + d] -= w->f[3 * k + d]; }
}

void orc_scratch_reserve(OrcScratch* w, int n) {
  if (n <= w->cap) return;
  int c = n + 64;
#define RE(p, sz) w->p = realloc(w->p, (size_t)c * (sz))
  RE(dr, 24); RE(dr2, 8); RE(qq, 8); RE(f, 24); RE(idx, 4);
  RE(dr_lj, 24); RE(dr2_lj, 8); RE(par_lj, 48); RE(f_lj, 24); RE(idx_lj, 4);
  RE(dr_sp, 24); RE(dr2_sp, 8); RE(par_sp, 48); RE(f_sp, 24); RE(idx_sp, 4);
#undef RE
  w->cap = c;
}
void orc_scratch_free(OrcScratch* w) {
  free(w->dr); free(w->dr2); free(w->qq); free(w->f); free(w->idx); free(w->dr_lj); free(w->dr2_lj); free(w->par_lj);
  free(w->f_lj); free(w->idx_lj); free(w->dr_sp); free(w->dr2_sp); free(w->par_sp); free(w->f_sp); free(w->idx_sp);
  memset(w, 0, sizeof(*w));
}
OrcTopo orc_principal_topo(OrcSys* s) {
  OrcTopo t; t.atype = s->atype; t.chg = s->chg; t.xyz = s->xyz; t.mol_of = s->mol_of; t.mol_type = s->mol_type;
  t.mol_natom = s->mol_natom; t.mol_atoms = s->mol_atoms; t.r_com = s->r_com; t.hydronium = s->hydronium; return t;
}

/* pairwise_real_space_verlet: src/pair_int_real_space.f90:135-331 */
void orc_pair_topo(OrcSys* s, const OrcTopo* tp, const int* vp, const int* nl, double* E, double* F) {
  int N = s->N, nth = s->nthreads;
  double e_el = 0.0, e_vdw = 0.0;
  double* lf = (double*)calloc((size_t)nth * 3 * N, sizeof(double));
#pragma omp parallel num_threads(nth) reduction(+ : e_el, e_vdw)
  {
    int tid = omp_get_thread_num();
    OrcScratch w; memset(&w, 0, sizeof(w));
    double* F_local = lf + (size_t)tid * 3 * N;
#pragma omp for schedule(dynamic, 64)
    for (int i = 0; i < N; ++i) {
      int st = vp[i], nn = vp[i + 1] - st;
      if (nn > 0) orc_atom_vs_list(s, tp, i, nn, &nl[st], NULL, &e_el, &e_vdw, F_local, &w);
    }
    orc_scratch_free(&w);
  }
  for (int t = 0; t < nth; ++t) for (size_t k = 0; k < 3 * (size_t)N; ++k) F[k] = F[k] + lf[(size_t)t * 3 * N + k];
  free(lf);
  E[0] = E[0] + e_el; E[1] = E[1] + e_vdw;
}
void orc_pair_real_space(OrcSys* s, double* E, double* F) { OrcTopo t = orc_principal_topo(s); orc_pair_topo(s, &t, s->verlet_point, s->neighbor_list, E, F); }

/* intra_molecular_pairwise_energy_force: src/pair_int_real_space.f90:386-588, for one molecule */
void orc_intra_nb_mol(const OrcSys* s, const OrcTopo* tp, int m, double* E_elec, double* E_vdw, double* F) {
  int n = tp->mol_natom[m], mtp = tp->mol_type[m];
  const int* at = &tp->mol_atoms[m * ORC_MAX_MOL_ATOMS];
  const OrcMolType* mt = &s->mt[mtp];
  double rc2 = s->rc * s->rc;
  for (int i = 0; i < n; ++i) {
    int gi = at[i], ti = tp->atype[gi];
    double dr_ex[24], dr2_ex[8], qq_ex[8], f_ex[24]; int id_ex[8], n_ex = 0;
    double dr_c[24], dr2_c[8], qq_c[8], f_c[24]; int id_c[8], n_c = 0, n_non = 0;
    double dr_l[24], dr2_l[8], par_l[48], f_l[24]; int id_l[8], n_l = 0;
    double dr_s[24], dr2_s[8], par_s[48], f_s[24]; int id_s[8], n_s = 0;
    for (int j = i + 1; j < n; ++j) {
      int gj = at[j], tj = tp->atype[gj];
      double dr[3]; for (int d = 0; d < 3; ++d) dr[d] = tp->xyz[3 * gi + d] - tp->xyz[3 * gj + d];
      double r2 = dot3(dr, dr), qq = tp->chg[gi] * tp->chg[gj];
      if (mt->excl[i][j] != 1) {
        n_non++;
        if (r2 < rc2) { memcpy(&dr_c[3 * n_c], dr, 24); dr2_c[n_c] = r2; qq_c[n_c] = qq; id_c[n_c] = gj; n_c++; }
        int vt = s->vdw_type[ti * s->nt + tj];
        if (vt == 0) {
          memcpy(&dr_l[3 * n_l], dr, 24); dr2_l[n_l] = r2; id_l[n_l] = gj;
          const double* src = mt->excl[i][j] == 2 ? s->vdw_par14 : s->vdw_par;
          memcpy(&par_l[6 * n_l], &src[(ti * s->nt + tj) * 6], 48); n_l++;
        } else if (vt == 1) {
          memcpy(&dr_s[3 * n_s], dr, 24); dr2_s[n_s] = r2; id_s[n_s] = gj;
          memcpy(&par_s[6 * n_s], &s->vdw_par[(ti * s->nt + tj) * 6], 48); n_s++;
        }
      } else { memcpy(&dr_ex[3 * n_ex], dr, 24); dr2_ex[n_ex] = r2; qq_ex[n_ex] = qq; id_ex[n_ex] = gj; n_ex++; }
    }
    memset(f_ex, 0, sizeof f_ex); memset(f_c, 0, sizeof f_c); memset(f_l, 0, sizeof f_l); memset(f_s, 0, sizeof f_s);
    double e;
    if (n_ex > 0) { k_pme_excl(s, n_ex, dr_ex, dr2_ex, qq_ex, &e, f_ex); *E_elec = *E_elec + e; }
    if (n_non > 0) {
      double e_el = 0, e_l = 0, e_s = 0;
      orc_k_ewald(s, n_c, dr_c, dr2_c, qq_c, &e_el, f_c);
      if (n_l > 0) orc_k_lj(n_l, dr_l, dr2_l, par_l, &e_l, f_l);
      if (n_s > 0) orc_k_sapt(s, n_s, dr_s, dr2_s, par_s, &e_s, f_s);
      *E_elec = *E_elec + e_el; *E_vdw = *E_vdw + e_l; *E_vdw = *E_vdw + e_s;
    }
    for (int k = 0; k < n_ex; ++k) for (int d = 0; d < 3; ++d) { F[3 * gi + d] += f_ex[3 * k + d]; F[3 * id_ex[k] + d] -= f_ex[3 * k + d]; }
    for (int k = 0; k < n_c; ++k) for (int d = 0; d < 3; ++d) { F[3 * gi + d] += f_c[3 * k + d]; F[3 * id_c[k] + d] -= f_c[3 * k + d]; }
    for (int k = 0; k < n_l; ++k) for (int d = 0; d < 3; ++d) { F[3 * gi + d] += f_l[3 * k + d]; F[3 * id_l[k] + d] -= f_l[3 * k + d]; }
    for (int k = 0; k < n_s; ++k) for (int d = 0; d < 3; ++d) { F[3 * gi + d] += f_s[3 * k + d]; F[3 * id_s[k] + d] -= f_s[3 * k + d]; }
  }
}
void orc_intra_nb_topo(OrcSys* s, const OrcTopo* tp, double* E, double* F) {
  for (int m = 0; m < s->M; ++m) if (tp->mol_natom[m] > 1) orc_intra_nb_mol(s, tp, m, &E[0], &E[1], F);
}
void orc_intra_nonbonded(OrcSys* s, double* E, double* F) { OrcTopo t = orc_principal_topo(s); orc_intra_nb_topo(s, &t, E, F); }

/* ------------------------------------------------------------------ PME */
/* create_scaled_direct_coordinates: src/general_routines.f90:497-524 */
void orc_scaled_coords(const double kk[3][3], int K, int n, const double* xyz, double* u) {
  for (int i = 0; i < n; ++i)
    for (int l = 0; l < 3; ++l) {
      double v = (double)K * dot3(kk[l], &xyz[3 * i]);
      if (v < 0.0) v = v + (double)K; else if (v >= (double)K) v = v - (double)K;
      if (fabs(fmod(v, 1.0)) < 1e-6) v = v + 1e-6;
      u[3 * i + l] = v;
    }
}
/* grid_Q / modify_Q_grid for one atom: src/pme.f90:218-260, :295-334 */
void orc_spread_atom(const OrcSys* s, double* Q, double q, const double* u, double sign) {
  int K = s->K, n = s->order, G = s->spline_grid;
  int np[3]; for (int d = 0; d < 3; ++d) np[d] = (int)floor(u[d]);
  for (int k3 = 0; k3 < n; ++k3) { int n3 = np[2] - k3; double a3 = u[2] - (double)n3; if (n3 < 0) n3 += K;
    int s3 = (int)ceil(a3 / 6.0 * (double)G);
    for (int k2 = 0; k2 < n; ++k2) { int n2 = np[1] - k2; double a2 = u[1] - (double)n2; if (n2 < 0) n2 += K;
      int s2 = (int)ceil(a2 / 6.0 * (double)G);
      for (int k1 = 0; k1 < n; ++k1) { int n1 = np[0] - k1; double a1 = u[0] - (double)n1; if (n1 < 0) n1 += K;
        int s1 = (int)ceil(a1 / 6.0 * (double)G);
        double v = q * s->B6[s1 - 1] * s->B6[s2 - 1] * s->B6[s3 - 1];
        size_t id = n1 + (size_t)K * (n2 + (size_t)K * n3);
        if (sign < 0) Q[id] = Q[id] - v; else Q[id] = Q[id] + v;
      } } }
}
/* derivative_grid_Q: src/pme.f90:346-498 */
void orc_gather_atom(const OrcSys* s, const double* theta, double q, const double* u, const double kk[3][3], double* force) {
  int K = s->K, n = s->order, G = s->spline_grid;
  int np[3]; for (int d = 0; d < 3; ++d) np[d] = (int)floor(u[d]);
  double fo[3] = {0, 0, 0};
  for (int k3 = 0; k3 < n; ++k3) { int n3 = np[2] - k3; double a13 = u[2] - (double)n3, a23 = a13 - 1.0; if (n3 < 0) n3 += K;
    for (int k2 = 0; k2 < n; ++k2) { int n2 = np[1] - k2; double a12 = u[1] - (double)n2, a22 = a12 - 1.0; if (n2 < 0) n2 += K;
      for (int k1 = 0; k1 < n; ++k1) { int n1 = np[0] - k1; double a11 = u[0] - (double)n1, a21 = a11 - 1.0; if (n1 < 0) n1 += K;
        double a1[3] = {a11, a12, a13}, a2[3] = {a21, a22, a23}, fac[3] = {0, 0, 0};
        int g2[3], g1n[3], g1m[3];
        for (int d = 0; d < 3; ++d) {
          g2[d] = (int)ceil(a2[d] / 5.0 * (double)G); g1n[d] = (int)ceil(a1[d] / 6.0 * (double)G); g1m[d] = (int)ceil(a1[d] / 5.0 * (double)G);
        }
        for (int d = 0; d < 3; ++d) {
          int o1 = (d + 1) % 3, o2 = (d + 2) % 3;
          /* the reference multiplies B5(d) * B6(lower other) * B6(higher other): x:(y,z) y:(x,z) z:(x,y) */
          int lo = o1 < o2 ? o1 : o2, hi = o1 < o2 ? o2 : o1;
          if (a1[d] < 5.0) fac[d] = q * (s->B5[g1m[d] - 1] * s->B6[g1n[lo] - 1] * s->B6[g1n[hi] - 1]);
          if (0.0 < a2[d]) fac[d] = fac[d] + q * (-s->B5[g2[d] - 1] * s->B6[g1n[lo] - 1] * s->B6[g1n[hi] - 1]);
        }
        double th = theta[n1 + (size_t)K * (n2 + (size_t)K * n3)];
        for (int d = 0; d < 3; ++d) fo[d] = fo[d] + fac[d] * th * s->conv;
      } } }
  for (int i = 0; i < 3; ++i) { double t = 0.0; for (int j = 0; j < 3; ++j) t = t - (double)K * kk[j][i] * fo[j]; force[i] = t; }
}

/* 3-D in-place complex DFT (unnormalised).  The reference calls MKL DFTI (src/pme.f90:85,113); any exact DFT is
   equivalent up to rounding.  Radix-2 when K is a power of two, O(K^2) line DFT otherwise. */
static void fft_line(double* re, double* im, int n, int sign, double* wr, double* wi, double* tr, double* ti) {
  if ((n & (n - 1)) == 0) {
    for (int i = 1, j = 0; i < n; ++i) { int bit = n >> 1; for (; j & bit; bit >>= 1) j ^= bit; j ^= bit;
      if (i < j) { double t = re[i]; re[i] = re[j]; re[j] = t; t = im[i]; im[i] = im[j]; im[j] = t; } }
    for (int len = 2; len <= n; len <<= 1) { int step = n / len;
      for (int i = 0; i < n; i += len) for (int k = 0; k < len / 2; ++k) {
        double cr = wr[k * step], ci = sign * wi[k * step];
        double xr = re[i + k + len / 2] * cr - im[i + k + len / 2] * ci, xi = re[i + k + len / 2] * ci + im[i + k + len / 2] * cr;
        re[i + k + len / 2] = re[i + k] - xr; im[i + k + len / 2] = im[i + k] - xi; re[i + k] += xr; im[i + k] += xi; } }
  } else {
    for (int k = 0; k < n; ++k) { double sr = 0, si = 0;
      for (int j = 0; j < n; ++j) { int id = (int)(((long)k * j) % n); double cr = wr[id], ci = sign * wi[id];
        sr += re[j] * cr - im[j] * ci; si += re[j] * ci + im[j] * cr; }
      tr[k] = sr; ti[k] = si; }
    memcpy(re, tr, n * sizeof(double)); memcpy(im, ti, n * sizeof(double));
  }
}
void orc_fft3d(int K, double* re, double* im, int sign) {
  double *wr = malloc(K * 8), *wi = malloc(K * 8);
  for (int k = 0; k < K; ++k) { wr[k] = cos(2.0 * M_PI * k / K); wi[k] = -sin(2.0 * M_PI * k / K); }
#pragma omp parallel
  {
    double *lr = malloc(K * 8), *li = malloc(K * 8), *tr = malloc(K * 8), *ti = malloc(K * 8);
    for (int axis = 0; axis < 3; ++axis) {
      size_t stride = axis == 0 ? 1 : axis == 1 ? (size_t)K : (size_t)K * K;
#pragma omp for
      for (int l = 0; l < K * K; ++l) {
        int a = l % K, b = l / K; size_t base;
        if (axis == 0) base = (size_t)K * (a + (size_t)K * b); else if (axis == 1) base = a + (size_t)K * K * b; else base = a + (size_t)K * b;
        for (int k = 0; k < K; ++k) { lr[k] = re[base + k * stride]; li[k] = im[base + k * stride]; }
        fft_line(lr, li, K, sign, wr, wi, tr, ti);
        for (int k = 0; k < K; ++k) { re[base + k * stride] = lr[k]; im[base + k * stride] = li[k]; }
      }
    }
    free(lr); free(li); free(tr); free(ti);
  }
  free(wr); free(wi);
}
/* theta = Re( IFFT( CB * FFT(Q) ) ), E = 0.5*sum(Q*theta)*conv : src/pme.f90:73-128 */
double orc_convolve(OrcSys* s, const double* Q, double* theta) {
  size_t K3 = (size_t)s->K * s->K * s->K;
  double *re = malloc(K3 * 8), *im = calloc(K3, 8);
  memcpy(re, Q, K3 * 8);
  orc_fft3d(s->K, re, im, +1);
  for (size_t i = 0; i < K3; ++i) { re[i] *= s->CB[i]; im[i] *= s->CB[i]; }
  orc_fft3d(s->K, re, im, -1);
  memcpy(theta, re, K3 * 8);
  double e = 0.0; for (size_t i = 0; i < K3; ++i) e += Q[i] * theta[i];
  free(re); free(im);
  return 0.5 * e * s->conv;
}
void orc_pme_topo(OrcSys* s, const OrcTopo* tp, double* E, double* F) {
  int N = s->N, K = s->K; size_t K3 = (size_t)K * K * K;
  double kk[3][3]; orc_recip_lattice(s->box, volume32(s->box), kk);
  memcpy(s->kk, kk, sizeof kk);
  orc_scaled_coords(kk, K, N, tp->xyz, s->xyz_scale);
  memset(s->Q, 0, K3 * 8);
  for (int i = 0; i < N; ++i) orc_spread_atom(s, s->Q, tp->chg[i], &s->xyz_scale[3 * i], +1);
  s->E_recip = orc_convolve(s, s->Q, s->theta);
  E[0] = E[0] + s->E_recip;
#pragma omp parallel for num_threads(s->nthreads) schedule(static)
  for (int i = 0; i < N; ++i) orc_gather_atom(s, s->theta, tp->chg[i], &s->xyz_scale[3 * i], kk, &s->force_recip[3 * i]);
  for (size_t k = 0; k < 3 * (size_t)N; ++k) F[k] = F[k] + s->force_recip[k];
}
void orc_pme_recip(OrcSys* s, double* E, double* F) { OrcTopo t = orc_principal_topo(s); orc_pme_topo(s, &t, E, F); }
void orc_get_grid(OrcSys* s, double* Q, double* theta, double* u, double* kk) {
  size_t K3 = (size_t)s->K * s->K * s->K;
  if (Q) memcpy(Q, s->Q, K3 * 8); if (theta) memcpy(theta, s->theta, K3 * 8);
  if (u) memcpy(u, s->xyz_scale, 3 * (size_t)s->N * 8); if (kk) memcpy(kk, s->kk, 72);
}
void orc_get_force_recip(OrcSys* s, double* F) { memcpy(F, s->force_recip, 3 * (size_t)s->N * 8); }

/* ------------------------------------------------------------------ bonded */
static int find_ang(const OrcSys* s, int i, int j, int k) {
  for (int a = 0; a < s->n_ang; ++a) if (s->ang_idx[4 * a] == i && s->ang_idx[4 * a + 1] == j && s->ang_idx[4 * a + 2] == k) return a;
  return -1;
}
static int find_dih(const OrcSys* s, int i, int j, int k, int l) {
  for (int a = 0; a < s->n_dih; ++a) if (s->dih_idx[5 * a] == i && s->dih_idx[5 * a + 1] == j && s->dih_idx[5 * a + 2] == k && s->dih_idx[5 * a + 3] == l) return a;
  return -1;
}
/* pairwise_bond_energy_force: src/intra_bonded_interactions.f90:125-168 */
static double k_bond(const OrcSys* s, int ti, int tj, const double* r, double rm, double* f) {
  int bt = s->bond_type[ti * s->nt + tj]; const double* p = &s->bond_par[(ti * s->nt + tj) * 3]; double e = 0;
  if (bt == 1) { e = 0.5 * p[1] * ((rm - p[0]) * (rm - p[0])); for (int d = 0; d < 3; ++d) f[d] = -p[1] * (rm - p[0]) * r[d] / rm; }
  else if (bt == 2) { double t = rm * rm - p[0] * p[0]; e = 0.25 * p[1] * (t * t); for (int d = 0; d < 3; ++d) f[d] = -p[1] * t * r[d]; }
  else if (bt == 3) { double D = p[0], beta = p[1], b0 = p[2], ex = exp(-beta * (rm - b0)); e = D * ((1.0 - ex) * (1.0 - ex));
    for (int d = 0; d < 3; ++d) f[d] = -2.0 * D * beta * ex * (1.0 - ex) * r[d] / rm; }
  else { fprintf(stderr, "oracle: bond type %d not implemented (types %d %d)\n", bt, ti, tj); abort(); }
  return e;
}
/* trimer_angle_energy_force: src/intra_bonded_interactions.f90:219-284 */
static double k_angle(const OrcSys* s, int ti, int tj, int tk, const double* rij, const double* rkj, double* fij, double* fkj) {
  int a = find_ang(s, ti, tj, tk);
  if (a < 0) { fprintf(stderr, "oracle: no angle type %d %d %d\n", ti, tj, tk); abort(); }
  int at = s->ang_idx[4 * a + 3]; double th0 = s->ang_par[2 * a], cth = s->ang_par[2 * a + 1];
  double rij_m = sqrt(dot3(rij, rij)), rkj_m = sqrt(dot3(rkj, rkj)), cosine = dot3(rij, rkj) / rij_m / rkj_m, e, fac;
  if (at == 1) {
    double theta;
    if (cosine < -0.999999999) theta = s->pi; else if (cosine > 0.999999999) theta = 0.0; else theta = acos(cosine);
    e = 0.5 * cth * ((theta - th0) * (theta - th0));
    if (fabs(theta - th0) < 1e-4) fac = 0.0; else fac = cth * (theta - th0) / sqrt(1.0 - cosine * cosine);
  } else { double c0 = cos(th0); e = 0.5 * cth * ((cosine - c0) * (cosine - c0)); fac = -cth * (cosine - c0); }
  for (int d = 0; d < 3; ++d) {
    fij[d] = fac * (rkj[d] / rij_m / rkj_m - cosine * rij[d] / (rij_m * rij_m));
    fkj[d] = fac * (rij[d] / rij_m / rkj_m - cosine * rkj[d] / (rkj_m * rkj_m));
  }
  return e;
}
/* quartet_dihedral_energy_force: src/intra_bonded_interactions.f90:343-552 */
static double k_dihedral(const OrcSys* s, int ti, int tj, int tk, int tl, const double* rji, const double* rkj, const double* rlk,
                         double* fji, double* fkj, double* flk) {
  int a = find_dih(s, ti, tj, tk, tl);
  if (a < 0) { fprintf(stderr, "oracle: no dihedral type %d %d %d %d\n", ti, tj, tk, tl); abort(); }
  int dt = s->dih_idx[5 * a + 4]; const double* p = &s->dih_par[6 * a];
  double rji2 = dot3(rji, rji), rkj2 = dot3(rkj, rkj), rlk2 = dot3(rlk, rlk);
  double d_kj_ji = dot3(rkj, rji), d_lk_kj = dot3(rlk, rkj), d_lk_ji = dot3(rlk, rji);
  double adb = d_kj_ji * d_lk_kj - d_lk_ji * rkj2, ada = rji2 * rkj2 - d_kj_ji * d_kj_ji, bdb = rlk2 * rkj2 - d_lk_kj * d_lk_kj;
  double dab_ji[3], dab_kj[3], dab_lk[3], daa_ji[3], daa_kj[3], daa_lk[3], dbb_ji[3], dbb_kj[3], dbb_lk[3];
  for (int d = 0; d < 3; ++d) {
    dab_ji[d] = rkj[d] * d_lk_kj - rlk[d] * rkj2;
    dab_kj[d] = rji[d] * d_lk_kj + d_kj_ji * rlk[d] - d_lk_ji * 2.0 * rkj[d];
    dab_lk[d] = d_kj_ji * rkj[d] - rji[d] * rkj2;
    daa_ji[d] = rkj2 * 2.0 * rji[d] - 2.0 * d_kj_ji * rkj[d]; daa_kj[d] = rji2 * 2.0 * rkj[d] - 2.0 * d_kj_ji * rji[d]; daa_lk[d] = 0.0;
    dbb_ji[d] = 0.0; dbb_kj[d] = rlk2 * 2.0 * rkj[d] - 2.0 * d_lk_kj * rlk[d]; dbb_lk[d] = rkj2 * 2.0 * rlk[d] - 2.0 * d_lk_kj * rkj[d];
  }
  double sa = sqrt(ada), sb = sqrt(bdb), cosine = adb / sa / sb, xi, e = 0, fac = 0; int shift = 0;
  if (cosine < -0.999999999) xi = s->pi; else if (cosine > 0.999999999) xi = 0.0; else xi = acos(cosine);
  if (dt == 1) {
    double xi0 = p[0], kxi = p[1], nm = p[2], c2 = cosine * cosine;
    e = kxi * (1.0 + cos(nm * xi - xi0));
    if (fabs(c2 - 1.0) < 1e-4) fac = 0.0; else fac = kxi * -sin(nm * xi - xi0) * nm / sqrt(1.0 - c2);
  } else if (dt == 2) {
    if (xi > s->pi / 2.0) { xi = fabs(xi - s->pi); shift = 1; }
    double xi0 = p[0], kxi = p[1];
    e = 0.5 * kxi * ((xi - xi0) * (xi - xi0));
    if (fabs(xi - xi0) < 1e-4) fac = 0.0; else fac = kxi * (xi - xi0) / sqrt(1.0 - cosine * cosine);
  } else {
    double c = cosine;
    e = p[0] - p[1] * c + p[2] * (c * c) - p[3] * powi(c, 3) + p[4] * powi(c, 4) - p[5] * powi(c, 5);
    fac = p[1] - 2.0 * p[2] * c + 3.0 * p[3] * (c * c) - 4.0 * p[4] * powi(c, 3) + 5.0 * p[5] * powi(c, 4);
  }
  double ada15 = pow(ada, 1.5), bdb15 = pow(bdb, 1.5);
  for (int d = 0; d < 3; ++d) {
    fji[d] = fac * (dab_ji[d] / sa / sb - 0.5 * adb / ada15 / sb * daa_ji[d] - 0.5 * adb / sa / bdb15 * dbb_ji[d]);
    fkj[d] = fac * (dab_kj[d] / sa / sb - 0.5 * adb / ada15 / sb * daa_kj[d] - 0.5 * adb / sa / bdb15 * dbb_kj[d]);
    flk[d] = fac * (dab_lk[d] / sa / sb - 0.5 * adb / ada15 / sb * daa_lk[d] - 0.5 * adb / sa / bdb15 * dbb_lk[d]);
    if (shift) { fji[d] = -fji[d]; fkj[d] = -fkj[d]; flk[d] = -flk[d]; }
  }
  return e;
}
/* bond + angle + dihedral of ONE molecule: src/intra_bonded_interactions.f90:84-335 */
void orc_bonded_mol(const OrcSys* s, const OrcTopo* tp, int m, double* Eb, double* Ea, double* Ed, double* F) {
  const OrcMolType* mt = &s->mt[tp->mol_type[m]]; const int* at = &tp->mol_atoms[m * ORC_MAX_MOL_ATOMS];
  double eb = 0, ea = 0, ed = 0;
  for (int b = 0; b < mt->nb; ++b) {
    int i = at[mt->bonds[2 * b]], j = at[mt->bonds[2 * b + 1]]; double r[3], f[3];
    for (int d = 0; d < 3; ++d) r[d] = tp->xyz[3 * i + d] - tp->xyz[3 * j + d];
    eb = eb + k_bond(s, tp->atype[i], tp->atype[j], r, sqrt(dot3(r, r)), f);
    for (int d = 0; d < 3; ++d) { F[3 * i + d] += f[d]; F[3 * j + d] -= f[d]; }
  }
  for (int a = 0; a < mt->na; ++a) {
    int i = at[mt->angles[3 * a]], j = at[mt->angles[3 * a + 1]], k = at[mt->angles[3 * a + 2]]; double rij[3], rkj[3], fij[3], fkj[3];
    for (int d = 0; d < 3; ++d) { rij[d] = tp->xyz[3 * i + d] - tp->xyz[3 * j + d]; rkj[d] = tp->xyz[3 * k + d] - tp->xyz[3 * j + d]; }
    ea = ea + k_angle(s, tp->atype[i], tp->atype[j], tp->atype[k], rij, rkj, fij, fkj);
    for (int d = 0; d < 3; ++d) { F[3 * i + d] += fij[d]; F[3 * k + d] += fkj[d]; F[3 * j + d] = F[3 * j + d] - fij[d] - fkj[d]; }
  }
  for (int q = 0; q < mt->nd; ++q) {
    int i = at[mt->dihedrals[4 * q]], j = at[mt->dihedrals[4 * q + 1]], k = at[mt->dihedrals[4 * q + 2]], l = at[mt->dihedrals[4 * q + 3]];
    double rji[3], rkj[3], rlk[3], fji[3], fkj[3], flk[3];
    for (int d = 0; d < 3; ++d) { rji[d] = tp->xyz[3 * j + d] - tp->xyz[3 * i + d]; rkj[d] = tp->xyz[3 * k + d] - tp->xyz[3 * j + d]; rlk[d] = tp->xyz[3 * l + d] - tp->xyz[3 * k + d]; }
    ed = ed + k_dihedral(s, tp->atype[i], tp->atype[j], tp->atype[k], tp->atype[l], rji, rkj, rlk, fji, fkj, flk);
    for (int d = 0; d < 3; ++d) {
      F[3 * i + d] = F[3 * i + d] - fji[d]; F[3 * j + d] = F[3 * j + d] + fji[d] - fkj[d];
      F[3 * k + d] = F[3 * k + d] + fkj[d] - flk[d]; F[3 * l + d] = F[3 * l + d] + flk[d];
    }
  }
  *Eb = eb; *Ea = ea; *Ed = ed;
}
void orc_bonded_topo(OrcSys* s, const OrcTopo* tp, double* E, double* F) {
  double eb = 0, ea = 0, ed = 0;
#pragma omp parallel for num_threads(s->nthreads) reduction(+ : eb, ea, ed) schedule(static) if (s->nthreads > 1)
  for (int m = 0; m < s->M; ++m) { double b, a, d; orc_bonded_mol(s, tp, m, &b, &a, &d, F); eb = eb + b; ea = ea + a; ed = ed + d; }
  E[2] = eb; E[3] = ea; E[4] = ed;
}
void orc_bonded(OrcSys* s, double* E, double* F) { OrcTopo t = orc_principal_topo(s); orc_bonded_topo(s, &t, E, F); }

/* calculate_total_force_energy: src/total_energy_forces.f90:19-99 (Verlet freshness handled by the caller) */
double orc_total_topo(OrcSys* s, const OrcTopo* tp, const int* vp, const int* nl, double* E, double* F) {
  memset(F, 0, 3 * (size_t)s->N * 8); for (int k = 0; k < 5; ++k) E[k] = 0.0;
  orc_pair_topo(s, tp, vp, nl, E, F);
  orc_intra_nb_topo(s, tp, E, F);
  orc_pme_topo(s, tp, E, F);
  E[0] = E[0] + s->ewald_self;
  orc_bonded_topo(s, tp, E, F);
  return E[0] + E[1] + E[2] + E[3] + E[4];
}
double orc_total(OrcSys* s, double* E, double* F) { OrcTopo t = orc_principal_topo(s); return orc_total_topo(s, &t, s->verlet_point, s->neighbor_list, E, F); }

/* ------------------------------------------------------------------ integrator */
void orc_update_com(OrcSys* s) { /* update_r_com: src/general_routines.f90:420, pos_com :398 */
  for (int m = 0; m < s->M; ++m) {
    double c[3] = {0, 0, 0}, mt = 0; const int* at = &s->mol_atoms[m * ORC_MAX_MOL_ATOMS];
    for (int k = 0; k < s->mol_natom[m]; ++k) { int a = at[k]; for (int d = 0; d < 3; ++d) c[d] = c[d] + s->xyz[3 * a + d] * s->mass[a]; mt = mt + s->mass[a]; }
    for (int d = 0; d < 3; ++d) s->r_com[3 * m + d] = c[d] / mt;
  }
}
static void shift_into_box(OrcSys* s) { /* shift_molecules_into_box: src/general_routines.f90:1145 */
  for (int m = 0; m < s->M; ++m) {
    double sh[3];
    for (int d = 0; d < 3; ++d) { double db = (1.0 / s->box[d]) * s->r_com[3 * m + d]; double v = 0.0; if (db < 0.0) v = 1.0; else if (db > 1.0) v = -1.0; sh[d] = 0.0 + v * s->box[d]; }
    const int* at = &s->mol_atoms[m * ORC_MAX_MOL_ATOMS];
    for (int d = 0; d < 3; ++d) s->r_com[3 * m + d] = s->r_com[3 * m + d] + sh[d];
    for (int k = 0; k < s->mol_natom[m]; ++k) for (int d = 0; d < 3; ++d) s->xyz[3 * at[k] + d] = s->xyz[3 * at[k] + d] + sh[d];
  }
}
static int frozen_atom(const OrcSys* s, int a) { return s->atype_freeze[s->atype[a]] == 1; }
void orc_set_atype_freeze(OrcSys* s, const int* f) { for (int t = 0; t < s->nt; ++t) s->atype_freeze[t] = f[t]; }
static void remove_com_momentum(OrcSys* s, double* vel) { /* src/md_integration.f90:125-177; frozen atom types carry no momentum and are not counted */
  double rho[3] = {0, 0, 0}; int n = 0;
  for (int m = 0; m < s->M; ++m) { const int* at = &s->mol_atoms[m * ORC_MAX_MOL_ATOMS];
    for (int k = 0; k < s->mol_natom[m]; ++k) { int a = at[k]; if (frozen_atom(s, a)) continue; n++; for (int d = 0; d < 3; ++d) rho[d] = rho[d] + s->mass[a] * vel[3 * a + d]; } }
  for (int d = 0; d < 3; ++d) rho[d] = rho[d] / (double)n;
  for (int a = 0; a < s->N; ++a) { if (frozen_atom(s, a)) continue; for (int d = 0; d < 3; ++d) vel[3 * a + d] = vel[3 * a + d] - rho[d] / s->mass[a]; }
}
double orc_kinetic_energy(OrcSys* s, const double* vel) { /* src/total_energy_forces.f90:106-121 */
  double ke = 0; for (int a = 0; a < s->N; ++a) ke = ke + 0.5 * s->mass[a] * dot3(&vel[3 * a], &vel[3 * a]) / 100.0; return ke;
}
int orc_verlet_flag(OrcSys* s) { return s->verlet_flag; }   /* flag_verlet_list as the last force call left it (src/total_energy_forces.f90:30-41) */
/* md_integrate_atomic, NVE: src/md_integration.f90:438-541 */
double orc_md_step(OrcSys* s, double* vel, double* F, double dt, int evb) {
  int N = s->N; double E[5], epot;
  for (int a = 0; a < N; ++a) { if (frozen_atom(s, a)) continue;      /* :473 */
    for (int d = 0; d < 3; ++d) {
      vel[3 * a + d] = vel[3 * a + d] + dt / 2.0 / s->mass[a] * F[3 * a + d] * 100.0;
      s->xyz[3 * a + d] = s->xyz[3 * a + d] + vel[3 * a + d] * dt;
    } }
  orc_update_com(s); shift_into_box(s);
  if (s->verlet_flag == 1) { orc_verlet_build(s, s->nslist); orc_verlet_check(s, 1); s->verlet_flag = 0; }
  else s->verlet_flag = orc_verlet_check(s, 0);
  if (evb) { epot = evb == 2 ? orc_evb_force_energy_delta_noverlet(s, F) : orc_evb_force_energy_noverlet(s, F); orc_evb_apply_principal(s); } else epot = orc_total(s, E, F);      /* evb = 2: the reference's delta-update formulation */
  for (int a = 0; a < N; ++a) { if (frozen_atom(s, a)) continue;      /* :512 */
    for (int d = 0; d < 3; ++d) vel[3 * a + d] = vel[3 * a + d] + dt / 2.0 / s->mass[a] * F[3 * a + d] * 100.0; }
  remove_com_momentum(s, vel);
  return epot;
}
