/*
 * TEST INFRASTRUCTURE ONLY (oracle/) - MS-EVB part of the CPU restatement (see msevb_oracle.h).
 *
 * Restates src/ms_evb.f90:181-3123 with one deliberate difference of FORMULATION (not of result):
 * the reference obtains every non-principal diabat by a delta update (donor/acceptor terms
 * subtracted in the old topology, added in the new one, src/ms_evb.f90:1421-1559, and a reciprocal
 * space update from stored dQ_dr, :1962-2248); here every diabat is evaluated FROM SCRATCH in its
 * own topology (test (4) "delta-vs-full").  The sums telescope to the same
 * Hamiltonian; two reference behaviours are kept on purpose: the Ewald self term is NOT re-evaluated
 * with the diabat's charges (only calculate_total_force_energy of the principal diabat adds it,
 * src/total_energy_forces.f90:71), and the acceptor heavy atom keeps the charge of its conjugate
 * type (src/ms_evb.f90:898-905).
 * Atom numbering never changes: a topology is (types, charges, molecule membership, proton image).
 */
#include "msevb_oracle_internal.h"

static inline double dot3(const double* a, const double* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

void orc_set_atype_chg(OrcSys* s, const double* c) { memcpy(s->atype_chg, c, s->nt * sizeof(double)); }

void orc_set_evb(OrcSys* s, const int* acid, const int* base, const int* conj_mol, const int* proton_type, const int* heavy_type,
                 const int* conj_atype, const double* ref_energy, const double* exch_atomic, const double* exch_proton, int n_da,
                 const int* da_types, const double* da_par, int n_pa, const int* pa_types, const double* pa_par, int n_cp,
                 const int* cp_types, const int* cp_func, const double* cp_par, int hydronium_mol, int max_chain) {
  for (int i = 0; i < s->nmt; ++i) { s->acid[i] = acid[i]; s->base[i] = base[i]; s->conj_mol[i] = conj_mol[i]; s->proton_type[i] = proton_type[i];
    s->heavy_type[i] = heavy_type[i]; s->ref_energy[i] = ref_energy[i];
    for (int j = 0; j < s->nmt; ++j) s->exch_proton[i][j] = exch_proton[i * s->nmt + j]; }
  for (int i = 0; i < s->nt; ++i) { s->conj_atype[i] = conj_atype[i]; s->exch_atomic[i] = exch_atomic[i]; }
  s->n_da = n_da; memcpy(s->da_types, da_types, n_da * 3 * sizeof(int)); memcpy(s->da_par, da_par, n_da * 6 * sizeof(double));
  s->n_pa = n_pa; memcpy(s->pa_types, pa_types, n_pa * 2 * sizeof(int)); memcpy(s->pa_par, pa_par, n_pa * 5 * sizeof(double));
  s->n_cp = n_cp; memcpy(s->cp_types, cp_types, n_cp * 3 * sizeof(int)); memcpy(s->cp_func, cp_func, n_cp * sizeof(int));
  memcpy(s->cp_par, cp_par, n_cp * 10 * sizeof(double));
  s->hydronium = hydronium_mol; s->max_chain = max_chain;
}

/* pbc_shift for an orthorhombic box (src/general_routines.f90:535-558): shift vector of r_j relative to r_i */
static void pbc_shift(const OrcSys* s, const double* ri, const double* rj, double* sh) {
  for (int d = 0; d < 3; ++d) { double db = (1.0 / s->box[d]) * (rj[d] - ri[d]); sh[d] = 0.0 + floor(db + 0.5) * s->box[d]; }
}
static inline void pbc_dr(const double* ri, const double* rj, const double* sh, double* o) { for (int d = 0; d < 3; ++d) o[d] = rj[d] - ri[d] - sh[d]; }

/* ---------------------------------------------------------------- topology copies and proton hops */
static OrcTopo topo_clone(const OrcSys* s, const OrcTopo* src) {
  OrcTopo t; size_t N = s->N, M = s->M;
  t.atype = malloc(N * 4); memcpy(t.atype, src->atype, N * 4);
  t.chg = malloc(N * 8); memcpy(t.chg, src->chg, N * 8);
  t.xyz = malloc(N * 24); memcpy(t.xyz, src->xyz, N * 24);
  t.mol_of = malloc(N * 4); memcpy(t.mol_of, src->mol_of, N * 4);
  t.mol_type = malloc(M * 4); memcpy(t.mol_type, src->mol_type, M * 4);
  t.mol_natom = malloc(M * 4); memcpy(t.mol_natom, src->mol_natom, M * 4);
  t.mol_atoms = malloc(M * ORC_MAX_MOL_ATOMS * 4); memcpy(t.mol_atoms, src->mol_atoms, M * ORC_MAX_MOL_ATOMS * 4);
  t.r_com = malloc(M * 24); memcpy(t.r_com, src->r_com, M * 24);
  t.hydronium = src->hydronium;
  return t;
}
static void topo_free(OrcTopo* t) { free(t->atype); free(t->chg); free(t->xyz); free(t->mol_of); free(t->mol_type); free(t->mol_natom); free(t->mol_atoms); free(t->r_com); }
static void mol_com(const OrcSys* s, const OrcTopo* t, int m, double* c) { /* pos_com: src/general_routines.f90:398 */
  double mt = 0; c[0] = c[1] = c[2] = 0; const int* at = &t->mol_atoms[m * ORC_MAX_MOL_ATOMS];
  for (int k = 0; k < t->mol_natom[m]; ++k) { int a = at[k]; for (int d = 0; d < 3; ++d) c[d] = c[d] + t->xyz[3 * a + d] * s->mass[a]; mt = mt + s->mass[a]; }
  for (int d = 0; d < 3; ++d) c[d] = c[d] / mt;
}
/* evb_change_data_structures_proton_transfer: src/ms_evb.f90:843-932 (+ :941-1006, :2992-3004), on index lists */
static void apply_hop(const OrcSys* s, OrcTopo* t, int md, int ip_local, int ma, int iheavy_local) {
  int* ad = &t->mol_atoms[md * ORC_MAX_MOL_ATOMS]; int* aa = &t->mol_atoms[ma * ORC_MAX_MOL_ATOMS];
  int gp = ad[ip_local];
  for (int k = ip_local; k < t->mol_natom[md] - 1; ++k) ad[k] = ad[k + 1];
  t->mol_natom[md] -= 1;
  int na = t->mol_natom[ma]; aa[na] = gp; t->mol_natom[ma] = na + 1; t->mol_of[gp] = ma; na += 1;
  t->hydronium = ma;
  /* make_molecule_whole on the acceptor: src/general_routines.f90:1065-1086 */
  for (int i = 1; i < na; ++i) {
    double sh[3]; pbc_shift(s, &t->xyz[3 * aa[i - 1]], &t->xyz[3 * aa[i]], sh);
    if (fabs(sh[0]) > 1e-6 || fabs(sh[1]) > 1e-6 || fabs(sh[2]) > 1e-6) {
      double dr[3]; pbc_dr(&t->xyz[3 * aa[i - 1]], &t->xyz[3 * aa[i]], sh, dr);
      for (int d = 0; d < 3; ++d) t->xyz[3 * aa[i] + d] = t->xyz[3 * aa[i - 1] + d] + dr[d];
    }
  }
  mol_com(s, t, md, &t->r_com[3 * md]); mol_com(s, t, ma, &t->r_com[3 * ma]);
  int base_type = t->mol_type[ma], acid_type = s->conj_mol[base_type];
  t->atype[gp] = s->proton_type[acid_type];
  for (int i = 0; i < na; ++i) {
    int a = aa[i], tn = (i != na - 1) ? s->conj_atype[t->atype[a]] : t->atype[a];
    t->atype[a] = tn; t->chg[a] = s->atype_chg[tn];
  }
  t->atype[aa[iheavy_local]] = s->heavy_type[acid_type];
  for (int i = 0; i < t->mol_natom[md]; ++i) { int a = ad[i], tn = s->conj_atype[t->atype[a]]; t->atype[a] = tn; t->chg[a] = s->atype_chg[tn]; }
  t->mol_type[md] = s->conj_mol[t->mol_type[md]];
  t->mol_type[ma] = acid_type;
  /* reorder_molecule_data_structures: src/ms_evb.f90:941-1006 */
  const OrcMolType* mt = &s->mt[acid_type];
  for (int i = 0; i < mt->natom; ++i) {
    if (mt->atype[i] != t->atype[aa[i]]) {
      int idx = -1;
      for (int j = i + 1; j < na; ++j) if (mt->atype[i] == t->atype[aa[j]]) { idx = j; break; }
      if (idx < 0) { fprintf(stderr, "oracle: reorder failed\n"); abort(); }
      int keep = aa[idx];
      for (int j = idx; j > i; --j) aa[j] = aa[j - 1];
      aa[i] = keep;
    }
  }
}
static void build_diabat_topo(const OrcSys* s, const OrcTopo* principal, int d, OrcTopo* out) {
  *out = topo_clone(s, principal);
  for (int h = 0; h < s->max_chain; ++h) {
    if (s->plog[d][h][0] < 0) break;
    apply_hop(s, out, out->hydronium, s->plog[d][h][1], s->plog[d][h][3], s->plog[d][h][4]);
  }
}

/* ---------------------------------------------------------------- diabat enumeration */
/* find_evb_reactive_neighbors: src/ms_evb.f90:702-764 */
static int find_neighbors(const OrcSys* s, const OrcTopo* t, int im, int ia_local, int nb[][2]) {
  int n = 0; const int* ai = &t->mol_atoms[im * ORC_MAX_MOL_ATOMS];
  for (int jm = 0; jm < s->M; ++jm) {
    if (jm == im) continue;
    double sh[3], dc[3]; pbc_shift(s, &t->r_com[3 * im], &t->r_com[3 * jm], sh); pbc_dr(&t->r_com[3 * im], &t->r_com[3 * jm], sh, dc);
    if (dot3(dc, dc) < 5.0 * 5.0) {
      const OrcMolType* mt = &s->mt[t->mol_type[jm]]; const int* aj = &t->mol_atoms[jm * ORC_MAX_MOL_ATOMS];
      for (int ja = 0; ja < t->mol_natom[jm]; ++ja) if (mt->reactive_basic[ja] == 1) {
        double r[3]; pbc_dr(&t->xyz[3 * ai[ia_local]], &t->xyz[3 * aj[ja]], sh, r);
        if (dot3(r, r) < 2.5 * 2.5 && n < 10) { nb[n][0] = jm; nb[n][1] = ja; n++; }
      }
    }
  }
  return n;
}
static int bonded_heavy(const OrcMolType* mt, int h) { /* find_bonded_atom_hydrogen: src/general_routines.f90:575 */
  int r = -1;
  for (int b = 0; b < mt->nb; ++b) { if (mt->bonds[2 * b] == h) r = mt->bonds[2 * b + 1]; else if (mt->bonds[2 * b + 1] == h) r = mt->bonds[2 * b]; }
  return r;
}
/* evb_conduct_proton_transfer_recursive: src/ms_evb.f90:498-607 (always on the principal topology) */
static void enumerate(OrcSys* s, const OrcTopo* t, int im_donor, int d_donor) {
  int count = 0;
  for (int h = 0; h < s->max_chain; ++h) { if (s->plog[d_donor][h][0] < 0) break; count++; }
  if (count >= s->max_chain) return;
  const OrcMolType* mt = &s->mt[t->mol_type[im_donor]];
  for (int ia = 0; ia < t->mol_natom[im_donor]; ++ia) {
    if (mt->reactive_protons[ia] != 1) continue;
    int nb[10][2]; int nn = find_neighbors(s, t, im_donor, ia, nb);
    for (int k = 0; k < nn; ++k) {
      if (s->D >= ORC_MAX_DIABATS) { fprintf(stderr, "oracle: too many diabats\n"); abort(); }
      int d = s->D++;
      s->parent[d] = d_donor;
      for (int h = 0; h < ORC_MAX_CHAIN; ++h) for (int q = 0; q < 5; ++q) s->plog[d][h][q] = h < count ? s->plog[d_donor][h][q] : -1;
      s->plog[d][count][0] = im_donor; s->plog[d][count][1] = ia; s->plog[d][count][2] = bonded_heavy(mt, ia);
      s->plog[d][count][3] = nb[k][0]; s->plog[d][count][4] = nb[k][1];
      if (nb[k][0] != s->hydronium) enumerate(s, t, nb[k][0], d);
    }
  }
}

/* ---------------------------------------------------------------- MS-EVB3 repulsion terms */
static void rep_switch(double r, double rs, double rc, double* sw, double* dsw) { /* src/ms_evb.f90:2484-2504 */
  *sw = 0.0; *dsw = 0.0;
  if (r < rc) { if (r < rs) *sw = 1.0; else { double t1 = ((r - rs) * (r - rs)) / ((rc - rs) * (rc - rs) * (rc - rs)), t2 = 3.0 * rc - rs - 2.0 * r;
      *sw = 1.0 - t1 * t2; *dsw = -2.0 * (r - rs) * t2 / ((rc - rs) * (rc - rs) * (rc - rs)) + 2.0 * t1; } }
}
static int heavy_local_acid(const OrcSys* s, int acid_type) { /* get_heavy_atom_transfer_acid: src/ms_evb.f90:2917 */
  const OrcMolType* mt = &s->mt[acid_type];
  for (int i = 0; i < mt->natom; ++i) if (mt->atype[i] == s->heavy_type[acid_type]) return i;
  fprintf(stderr, "oracle: heavy atom not found\n"); abort();
}
/* ms_evb_intermolecular_repulsion = three_atom (src/ms_evb.f90:2295-2399) + born_mayer (:2405-2478) */
static double evb_repulsion(const OrcSys* s, const OrcTopo* t, double* F) {
  double E = 0.0; int ih = t->hydronium; const int* ai = &t->mol_atoms[ih * ORC_MAX_MOL_ATOMS]; int ni = t->mol_natom[ih];
  int type_H = t->atype[ai[ni - 1]], iheavy = heavy_local_acid(s, t->mol_type[ih]), gheavy = ai[iheavy], type_heavy = t->atype[gheavy];
  for (int jm = 0; jm < s->M; ++jm) {
    if (jm == ih) continue;
    const int* aj = &t->mol_atoms[jm * ORC_MAX_MOL_ATOMS];
    for (int ja = 0; ja < t->mol_natom[jm]; ++ja) {
      int gj = aj[ja], tj = t->atype[gj], idx = -1;
      for (int q = 0; q < s->n_da; ++q) if (s->da_types[q][0] == tj && s->da_types[q][1] == type_heavy && s->da_types[q][2] == type_H) { idx = q; break; }
      if (idx < 0) continue;
      const double* p = s->da_par[idx]; double B = p[0], bl = p[1], d0 = p[2], blp = p[3], rs = p[4], rc = p[5];
      double sh[3], rO[3]; pbc_shift(s, &t->xyz[3 * gheavy], &t->xyz[3 * gj], sh); pbc_dr(&t->xyz[3 * gheavy], &t->xyz[3 * gj], sh, rO);
      for (int d = 0; d < 3; ++d) rO[d] = -rO[d];
      double rOO = sqrt(dot3(rO, rO)), sw, dsw; rep_switch(rOO, rs, rc, &sw, &dsw);
      double facOO = B * exp(-bl * (rOO - d0)), sum = 0.0;
      for (int ia = 0; ia < ni; ++ia) {
        int gi = ai[ia];
        if (t->atype[gi] != type_H) continue;
        double rij[3], q[3]; pbc_dr(&t->xyz[3 * gi], &t->xyz[3 * gj], sh, rij);
        for (int d = 0; d < 3; ++d) rij[d] = -rij[d];
        for (int d = 0; d < 3; ++d) q[d] = (2.0 * t->xyz[3 * gj + d] + rO[d]) / 2.0 - (t->xyz[3 * gj + d] + rij[d]);
        double q2 = dot3(q, q), eq = exp(-blp * q2);
        sum = sum + eq;
        for (int d = 0; d < 3; ++d) {
          F[3 * gi + d] = F[3 * gi + d] + sw * facOO * eq * -blp * 2.0 * q[d];
          F[3 * gheavy + d] = F[3 * gheavy + d] + sw * facOO * eq * blp * q[d];
          F[3 * gj + d] = F[3 * gj + d] + sw * facOO * eq * blp * q[d];
        }
      }
      E = E + sw * facOO * sum;
      for (int d = 0; d < 3; ++d) { double f = rO[d] / rOO * facOO * sum * (sw * bl - dsw); F[3 * gheavy + d] = F[3 * gheavy + d] + f; F[3 * gj + d] = F[3 * gj + d] - f; }
    }
  }
  for (int ia = 0; ia < ni; ++ia) {
    int gi = ai[ia], ti = t->atype[gi];
    for (int jm = 0; jm < s->M; ++jm) {
      if (jm == ih) continue;
      const int* aj = &t->mol_atoms[jm * ORC_MAX_MOL_ATOMS];
      for (int ja = 0; ja < t->mol_natom[jm]; ++ja) {
        int gj = aj[ja], tj = t->atype[gj], idx = -1;
        for (int q = 0; q < s->n_pa; ++q) if (s->pa_types[q][0] == tj && s->pa_types[q][1] == ti) { idx = q; break; }
        if (idx < 0) continue;
        const double* p = s->pa_par[idx]; double C = p[0], cl = p[1], d0 = p[2], rs = p[3], rc = p[4];
        double sh[3], rij[3]; pbc_shift(s, &t->xyz[3 * gi], &t->xyz[3 * gj], sh); pbc_dr(&t->xyz[3 * gi], &t->xyz[3 * gj], sh, rij);
        for (int d = 0; d < 3; ++d) rij[d] = -rij[d];
        double r = sqrt(dot3(rij, rij)), fac = C * exp(-cl * (r - d0)), sw, dsw; rep_switch(r, rs, rc, &sw, &dsw);
        E = E + sw * fac;
        for (int d = 0; d < 3; ++d) { double f = rij[d] / r * fac * (sw * cl - dsw); F[3 * gi + d] = F[3 * gi + d] + f; F[3 * gj + d] = F[3 * gj + d] - f; }
      }
    }
  }
  return E;
}

/* ---------------------------------------------------------------- off-diagonal coupling */
/* evb_diabatic_coupling_function: src/ms_evb.f90:1180-1266; dA[3][3] = d/d(O_donor), d/d(O_acceptor), d/d(H) */
static void coupling_function(int ftype, const double* p, const double* q, const double* rOO, double* A, double* Vc, double dA[3][3]) {
  double rm = sqrt(dot3(rOO, rOO)), q2 = dot3(q, q), qm = sqrt(q2);
  if (ftype == 1) {
    double gamma = p[1], P = p[2], k = p[3], D = p[4], beta = p[5], R0 = p[6], Pp = p[7], alpha = p[8], rl0 = p[9];
    *Vc = p[0];
    double f1 = exp(-gamma * q2), f2 = 1.0 + P * exp(-k * ((rm - D) * (rm - D))), f3 = 0.5 * (1.0 - tanh(beta * (rm - R0))) + Pp * exp(-alpha * (rm - rl0));
    double ch = cosh(beta * (rm - R0));
    double d1 = -gamma * 2.0 * qm * exp(-gamma * q2), d2 = P * -k * 2.0 * (rm - D) * exp(-k * ((rm - D) * (rm - D))),
           d3 = -0.5 * beta / (ch * ch) - Pp * alpha * exp(-alpha * (rm - rl0));
    *A = f1 * f2 * f3;
    for (int d = 0; d < 3; ++d) {
      dA[0][d] = d1 * f2 * f3 * 0.5 * q[d] / qm; dA[0][d] = dA[0][d] + f1 * d2 * f3 * rOO[d] / rm; dA[0][d] = dA[0][d] + f1 * f2 * d3 * rOO[d] / rm;
      dA[1][d] = d1 * f2 * f3 * 0.5 * q[d] / qm; dA[1][d] = dA[1][d] + f1 * d2 * f3 * -rOO[d] / rm; dA[1][d] = dA[1][d] + f1 * f2 * d3 * -rOO[d] / rm;
      dA[2][d] = d1 * f2 * f3 * -q[d] / qm;
    }
  } else {
    double gamma = p[1], k = p[2], D = p[3];
    *Vc = p[0];
    double f1 = exp(-gamma * q2), f2 = exp(-k * ((rm - D) * (rm - D)));
    double d1 = -gamma * 2.0 * qm * exp(-gamma * q2), d2 = -k * 2.0 * (rm - D) * exp(-k * ((rm - D) * (rm - D)));
    *A = f1 * f2;
    for (int d = 0; d < 3; ++d) {
      dA[0][d] = d1 * f2 * 0.5 * q[d] / qm; dA[0][d] = dA[0][d] + f1 * d2 * rOO[d] / rm;
      dA[1][d] = d1 * f2 * 0.5 * q[d] / qm; dA[1][d] = dA[1][d] + f1 * d2 * -rOO[d] / rm;
      dA[2][d] = d1 * f2 * -q[d] / qm;
    }
  }
}
/* evb_diabatic_coupling (src/ms_evb.f90:1021-1104) in topology t (acceptor topology of diabat d); F is overwritten */
static double evb_coupling(const OrcSys* s, const OrcTopo* t, int md, int ma, double* F) {
  int N = s->N; memset(F, 0, 24 * (size_t)N);
  double* dV = calloc(3 * (size_t)N, 8); double Vex = 0.0;
  const int* ad = &t->mol_atoms[md * ORC_MAX_MOL_ATOMS]; const int* aa = &t->mol_atoms[ma * ORC_MAX_MOL_ATOMS];
  int nd = t->mol_natom[md], na = t->mol_natom[ma];
  /* zundel_r_com: src/ms_evb.f90:2946-2982 */
  double md_tot = 0, ma_tot = 0;
  for (int i = 0; i < nd; ++i) md_tot = md_tot + s->mass[ad[i]];
  for (int i = 0; i < na; ++i) ma_tot = ma_tot + s->mass[aa[i]];
  double shifta[3], rda[3], rca[3], zc[3], shiftd[3] = {0, 0, 0};
  pbc_shift(s, &t->r_com[3 * md], &t->r_com[3 * ma], shifta); pbc_dr(&t->r_com[3 * md], &t->r_com[3 * ma], shifta, rda);
  for (int d = 0; d < 3; ++d) { rca[d] = t->r_com[3 * md + d] + rda[d]; zc[d] = (md_tot * t->r_com[3 * md + d] + ma_tot * rca[d]) / (md_tot + ma_tot); }
  double q_tr = s->exch_proton[t->mol_type[ma]][t->mol_type[md]];
  /* evb_diabatic_coupling_electrostatics: src/ms_evb.f90:1276-1403 */
  for (int jm = 0; jm < s->M; ++jm) {
    if (jm == md || jm == ma) continue;
    double sh[3]; pbc_shift(s, zc, &t->r_com[3 * jm], sh);
    const int* aj = &t->mol_atoms[jm * ORC_MAX_MOL_ATOMS];
    for (int ja = 0; ja < t->mol_natom[jm]; ++ja) {
      int gj = aj[ja]; double qj = t->chg[gj];
      for (int pass = 0; pass < 2; ++pass) {
        const int* al = pass == 0 ? ad : aa; int nl = pass == 0 ? nd : na; const double* shm = pass == 0 ? shiftd : shifta;
        for (int ia = 0; ia < nl; ++ia) {
          int gi = al[ia]; double qi = (pass == 1 && ia == na - 1) ? q_tr : s->exch_atomic[t->atype[gi]];
          double dr[3], xz[3], rij[3]; pbc_dr(zc, &t->xyz[3 * gi], shm, dr);
          for (int d = 0; d < 3; ++d) xz[d] = zc[d] + dr[d];
          pbc_dr(xz, &t->xyz[3 * gj], sh, rij); for (int d = 0; d < 3; ++d) rij[d] = -rij[d];
          double rm = sqrt(dot3(rij, rij));
          Vex = Vex + qi * qj / rm * s->conv;
          for (int d = 0; d < 3; ++d) { double dv = -qi * qj / (rm * rm * rm) * rij[d] * s->conv; dV[3 * gi + d] = dV[3 * gi + d] + dv; dV[3 * gj + d] = dV[3 * gj + d] - dv; }
        }
      }
    }
  }
  /* evb_diabatic_coupling_geometric: src/ms_evb.f90:1117-1174 */
  int acid_of_donor = s->conj_mol[t->mol_type[md]];
  int i_don = heavy_local_acid(s, acid_of_donor), i_acc = heavy_local_acid(s, t->mol_type[ma]);
  int gO1 = ad[i_don], gO2 = aa[i_acc], gH = aa[na - 1];
  double rO1[3], rO2[3], rH[3], sh[3], rij[3], rOO[3], q[3];
  memcpy(rO1, &t->xyz[3 * gO1], 24);
  pbc_shift(s, rO1, &t->xyz[3 * gO2], sh); pbc_dr(rO1, &t->xyz[3 * gO2], sh, rij); for (int d = 0; d < 3; ++d) rO2[d] = rO1[d] + rij[d];
  pbc_dr(rO1, &t->xyz[3 * gH], sh, rij); for (int d = 0; d < 3; ++d) rH[d] = rO1[d] + rij[d];
  for (int d = 0; d < 3; ++d) { rOO[d] = rO1[d] - rO2[d]; q[d] = (rO1[d] + rO2[d]) / 2.0 - rH[d]; }
  int t1 = t->atype[gO1], t2 = t->atype[gO2], t3 = t->atype[gH], idx = -1;
  for (int k = 0; k < s->n_cp; ++k) if (s->cp_types[k][0] == t1 && s->cp_types[k][1] == t2 && s->cp_types[k][2] == t3) { idx = k; break; }
  if (idx < 0) { fprintf(stderr, "oracle: no coupling parameters for types %d %d %d\n", t1, t2, t3); abort(); }
  double A, Vc, dA[3][3]; coupling_function(s->cp_func[idx], s->cp_par[idx], q, rOO, &A, &Vc, dA);
  double E = (Vc + Vex) * A;
  for (int d = 0; d < 3; ++d) { F[3 * gO1 + d] = -(Vc + Vex) * dA[0][d]; F[3 * gO2 + d] = -(Vc + Vex) * dA[1][d]; F[3 * gH + d] = -(Vc + Vex) * dA[2][d]; }
  for (size_t k = 0; k < 3 * (size_t)N; ++k) F[k] = F[k] - dV[k] * A;
  free(dV);
  return E;
}

/* ---------------------------------------------------------------- jacobi: src/general_routines.f90:2013-2088 */
static void jrot(double* a1, int s1, double* a2, int s2, int n, double sn, double tau) {
  for (int k = 0; k < n; ++k) { double w = a1[k * s1], x = a2[k * s2]; a1[k * s1] = w - sn * (x + w * tau); a2[k * s2] = x + sn * (w - x * tau); }
}
static int jacobi(int n, double* a /*[n][n] row-major, a[i][j]*/, double* d, double* v) {
  double b[ORC_MAX_DIABATS], z[ORC_MAX_DIABATS]; int nrot = 0;
#define A(i, j) a[(i) * n + (j)]
#define V(i, j) v[(i) * n + (j)]
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) V(i, j) = i == j ? 1.0 : 0.0;
  for (int i = 0; i < n; ++i) { b[i] = A(i, i); d[i] = b[i]; z[i] = 0.0; }
  for (int it = 1; it <= 50; ++it) {
    double sm = 0.0;
    for (int j = 0; j < n; ++j) for (int i = 0; i < j; ++i) sm += fabs(A(i, j));   /* column-major sum over the upper triangle */
    if (sm == 0.0) return nrot;
    double tresh = it < 4 ? 0.2 * sm / (double)(n * n) : 0.0;
    for (int ip = 0; ip < n - 1; ++ip) for (int iq = ip + 1; iq < n; ++iq) {
      double g = 100.0 * fabs(A(ip, iq));
      if (it > 4 && fabs(d[ip]) + g == fabs(d[ip]) && fabs(d[iq]) + g == fabs(d[iq])) A(ip, iq) = 0.0;
      else if (fabs(A(ip, iq)) > tresh) {
        double h = d[iq] - d[ip], t;
        if (fabs(h) + g == fabs(h)) t = A(ip, iq) / h;
        else { double theta = 0.5 * h / A(ip, iq); t = 1.0 / (fabs(theta) + sqrt(1.0 + theta * theta)); if (theta < 0.0) t = -t; }
        double c = 1.0 / sqrt(1 + t * t), sn = t * c, tau = sn / (1.0 + c);
        h = t * A(ip, iq);
        z[ip] -= h; z[iq] += h; d[ip] -= h; d[iq] += h; A(ip, iq) = 0.0;
        jrot(&A(0, ip), n, &A(0, iq), n, ip, sn, tau);
        jrot(&A(ip, ip + 1), 1, &A(ip + 1, iq), n, iq - ip - 1, sn, tau);
        jrot(&A(ip, iq + 1), 1, &A(iq, iq + 1), 1, n - iq - 1, sn, tau);
        jrot(&V(0, ip), n, &V(0, iq), n, n, sn, tau);
        nrot++;
      }
    }
    for (int i = 0; i < n; ++i) { b[i] += z[i]; d[i] = b[i]; z[i] = 0.0; }
  }
  fprintf(stderr, "oracle: too many iterations in jacobi\n"); abort();
#undef A
#undef V
}

/* ms_evb_intermolecular_repulsion of the principal topology alone (src/ms_evb.f90:2259): energy returned, forces ADDED to F; and the switching function */
double orc_evb_repulsion_principal(OrcSys* s, double* F) { OrcTopo P = orc_principal_topo(s); return evb_repulsion(s, &P, F); }
void orc_evb_rep_switch(double r, double rs, double rc, double* sw, double* dsw) { rep_switch(r, rs, rc, sw, dsw); }

/* diagonalize_evb_hamiltonian (src/ms_evb.f90:242-351): ground state, Hellmann-Feynman blend of the stored diabat / coupling forces, principal diabat */
static double evb_diag_blend(OrcSys* s, double* F) {
  int N = s->N, D = s->D;
  double *a = malloc(D * D * 8), *v = malloc(D * D * 8), ev[ORC_MAX_DIABATS];
  for (int i = 0; i < D; ++i) for (int j = i; j < D; ++j) { a[i * D + j] = s->H[i][j]; a[j * D + i] = s->H[i][j]; }
  jacobi(D, a, ev, v);
  int g = 0; double eg = ev[0];
  for (int i = 1; i < D; ++i) if (ev[i] < eg) { eg = ev[i]; g = i; }
  for (int i = 0; i < D; ++i) s->eigvec[i] = v[i * D + g];
  memset(F, 0, 24 * (size_t)N);
  for (int i = 0; i < D; ++i) for (int j = i; j < D; ++j) {
    const double* Fs = NULL; double fac;
    if (i == j) { Fs = s->evb_Fdiag + (size_t)i * 3 * N; fac = s->eigvec[i] * s->eigvec[j]; }
    else if (s->parent[j] == i) { Fs = s->evb_Foff + (size_t)j * 3 * N; fac = 2.0 * s->eigvec[i] * s->eigvec[j]; }
    else continue;
    for (size_t k = 0; k < 3 * (size_t)N; ++k) F[k] = F[k] + fac * Fs[k];
  }
  int pr = 0; double sc = fabs(s->eigvec[0]);
  for (int i = 0; i < D; ++i) if (sc < fabs(s->eigvec[i])) { sc = fabs(s->eigvec[i]); pr = i; }
  s->principal = pr; s->new_hydronium = s->hydronium;
  for (int h = 0; h < s->max_chain; ++h) { if (s->plog[pr][h][0] < 0) break; s->new_hydronium = s->plog[pr][h][3]; }
  free(a); free(v);
  return eg;
}

/* ---------------------------------------------------------------- the EVB force call */
/* construct_evb_hamiltonian (src/ms_evb.f90:375-489) + diagonalize_evb_hamiltonian (:242-351) */
double orc_evb_force_energy_noverlet(OrcSys* s, double* F) {
  int N = s->N; double E[5];
  OrcTopo P = orc_principal_topo(s);
  orc_update_com(s);
  /* principal diabat */
  for (int d = 0; d < ORC_MAX_DIABATS; ++d) { s->parent[d] = -1; for (int h = 0; h < ORC_MAX_CHAIN; ++h) for (int q = 0; q < 5; ++q) s->plog[d][h][q] = -1; }
  memset(s->H, 0, sizeof s->H);
  s->D = 1;
  enumerate(s, &P, s->hydronium, 0);
  int D = s->D;
  if (s->evb_cap < D) { free(s->evb_Fdiag); free(s->evb_Foff); s->evb_Fdiag = malloc((size_t)D * 3 * N * 8); s->evb_Foff = malloc((size_t)D * 3 * N * 8); s->evb_cap = D; }
  memset(s->evb_Foff, 0, (size_t)D * 3 * N * 8);
  double* F1 = s->evb_Fdiag;
  double e1 = orc_total_topo(s, &P, s->verlet_point, s->neighbor_list, E, F1);
  e1 = e1 + evb_repulsion(s, &P, F1);
  e1 = e1 + s->ref_energy[P.mol_type[P.hydronium]];
  s->H[0][0] = e1;
  /* keep the principal diabat's PME products (the product's delta path starts from them) */
  size_t K3 = (size_t)s->K * s->K * s->K; double *Q1 = malloc(K3 * 8), *th1 = malloc(K3 * 8), *fr1 = malloc(24 * (size_t)N), *us1 = malloc(24 * (size_t)N); double er1 = s->E_recip;
  memcpy(Q1, s->Q, K3 * 8); memcpy(th1, s->theta, K3 * 8); memcpy(fr1, s->force_recip, 24 * (size_t)N); memcpy(us1, s->xyz_scale, 24 * (size_t)N);
  for (int d = 1; d < D; ++d) {
    OrcTopo T; build_diabat_topo(s, &P, d, &T);
    int* vp = malloc((N + 1) * 4); int* nl = NULL;
    orc_verlet_build_topo(s, &T, s->nslist, vp, &nl);
    double* Fd = s->evb_Fdiag + (size_t)d * 3 * N;
    double ed = orc_total_topo(s, &T, vp, nl, E, Fd);
    ed = ed + evb_repulsion(s, &T, Fd);
    ed = ed + s->ref_energy[T.mol_type[T.hydronium]];
    s->H[d][d] = ed;
    /* last hop's donor / acceptor */
    int md = s->hydronium, ma = s->hydronium;
    for (int h = 0; h < s->max_chain; ++h) { if (s->plog[d][h][0] < 0) break; md = ma; ma = s->plog[d][h][3]; }
    s->H[s->parent[d]][d] = evb_coupling(s, &T, md, ma, s->evb_Foff + (size_t)d * 3 * N);
    free(vp); free(nl); topo_free(&T);
  }
  memcpy(s->Q, Q1, K3 * 8); memcpy(s->theta, th1, K3 * 8); memcpy(s->force_recip, fr1, 24 * (size_t)N); memcpy(s->xyz_scale, us1, 24 * (size_t)N); s->E_recip = er1;
  free(Q1); free(th1); free(fr1); free(us1);
  return evb_diag_blend(s, F);
}
double orc_evb_force_energy(OrcSys* s, double* F) {
  if (s->verlet_flag == 1 || !s->neighbor_list) { orc_verlet_build(s, s->nslist); orc_verlet_check(s, 1); s->verlet_flag = 0; }
  return orc_evb_force_energy_noverlet(s, F);
}
/* The same call in the reference's DELTA-UPDATE formulation (ms_evb_diabat_force_energy, src/ms_evb.f90:1421-1560, with its real-space part
   ms_evb_diabat_force_energy_update_real_space :1566-1860 and the per-diabat reciprocal space calculate_reciprocal_space_pme :1962-2090): a diabat's
   energy and force START from the principal diabat's.  The molecules the hops of its chain changed (every donor / acceptor) are taken out in the
   principal topology and put back in the diabat's own: their intra-molecular terms, their real-space pairs with ALL atoms inside the cutoff (one
   all-atom scan per changed atom - no Verlet list, :1629-1732), their stencils on a copy of the principal charge grid (modify_Q_grid, src/pme.f90:275);
   then ONE reciprocal-space pass per diabat on the patched grid - transform, convolution, back-transform, force update of ALL atoms.  The reference
   walks the chain hop by hop; the intermediate topologies cancel, so this takes principal -> diabat d in one step (same scans, same grid, same sums).
   Two users: the delta-vs-full known-answer test (tests/test_oracle_vs_reference.py: equal to orc_evb_force_energy, which evaluates every diabat from
   scratch) and bench.py's timed CPU baseline (what the reference does per MD step). */
static void evb_changed_terms(const OrcSys* s, const OrcTopo* X, int nchg, const int* mols, double sign, double* e_acc, double* Fd, double* Ftmp,
                              unsigned char* screen, double* Q, const double kk[3][3], OrcScratch* w) {
  int N = s->N;
  memset(Ftmp, 0, 24 * (size_t)N);
  memset(screen, 1, (size_t)N);
  double e_el = 0.0, e_vdw = 0.0, eb = 0.0, ea = 0.0, ed = 0.0;
  for (int k = 0; k < nchg; ++k) {
    int m = mols[k], n = X->mol_natom[m]; const int* at = &X->mol_atoms[m * ORC_MAX_MOL_ATOMS];
    for (int q = 0; q < n; ++q) screen[at[q]] = 0;      /* own molecule (intra-molecular terms below) and the changed molecules already done (pairs count once) */
    for (int q = 0; q < n; ++q) orc_atom_vs_list(s, X, at[q], N, NULL, screen, &e_el, &e_vdw, Ftmp, w);
    if (n > 1) orc_intra_nb_mol(s, X, m, &e_el, &e_vdw, Ftmp);
    { double b, a, d; orc_bonded_mol(s, X, m, &b, &a, &d, Ftmp); eb = eb + b; ea = ea + a; ed = ed + d; }
    for (int q = 0; q < n; ++q) { double u[3]; orc_scaled_coords(kk, s->K, 1, &X->xyz[3 * at[q]], u); orc_spread_atom(s, Q, X->chg[at[q]], u, sign); }
  }
  *e_acc = *e_acc + sign * (e_el + e_vdw + eb + ea + ed);
  for (size_t k = 0; k < 3 * (size_t)N; ++k) Fd[k] = Fd[k] + sign * Ftmp[k];
}
double orc_evb_force_energy_delta_noverlet(OrcSys* s, double* F) {
  int N = s->N, K = s->K; double E[5];
  size_t K3 = (size_t)K * K * K, n3 = 3 * (size_t)N;
  OrcTopo P = orc_principal_topo(s);
  orc_update_com(s);
  for (int d = 0; d < ORC_MAX_DIABATS; ++d) { s->parent[d] = -1; for (int h = 0; h < ORC_MAX_CHAIN; ++h) for (int q = 0; q < 5; ++q) s->plog[d][h][q] = -1; }
  memset(s->H, 0, sizeof s->H);
  s->D = 1;
  enumerate(s, &P, s->hydronium, 0);
  int D = s->D;
  if (s->evb_cap < D) { free(s->evb_Fdiag); free(s->evb_Foff); s->evb_Fdiag = malloc((size_t)D * n3 * 8); s->evb_Foff = malloc((size_t)D * n3 * 8); s->evb_cap = D; }
  memset(s->evb_Foff, 0, (size_t)D * n3 * 8);
  /* principal diabat: the full force call (src/ms_evb.f90:411) */
  double* F1 = s->evb_Fdiag;
  double e1_terms = orc_total_topo(s, &P, s->verlet_point, s->neighbor_list, E, F1);      /* F1: without the repulsion terms */
  double *Q1 = malloc(K3 * 8), *th1 = malloc(K3 * 8), *fr1 = malloc(8 * n3), *us1 = malloc(8 * n3), *Ftmp = malloc(8 * n3), *Qd = malloc(K3 * 8), *thd = malloc(K3 * 8), *usd = malloc(8 * n3);
  unsigned char* screen = malloc((size_t)N);
  double er1 = s->E_recip; double kk[3][3]; memcpy(kk, s->kk, sizeof kk);
  memcpy(Q1, s->Q, K3 * 8); memcpy(th1, s->theta, K3 * 8); memcpy(fr1, s->force_recip, 8 * n3); memcpy(us1, s->xyz_scale, 8 * n3);
  OrcScratch w; memset(&w, 0, sizeof w);
  for (int d = 1; d < D; ++d) {
    OrcTopo T; build_diabat_topo(s, &P, d, &T);
    int mols[ORC_MAX_CHAIN + 1], nchg = 0, md = s->hydronium, ma = s->hydronium, md_last = s->hydronium;
    mols[nchg++] = s->hydronium;
    for (int h = 0; h < s->max_chain; ++h) { if (s->plog[d][h][0] < 0) break; md_last = ma; ma = s->plog[d][h][3]; int seen = 0; for (int q = 0; q < nchg; ++q) if (mols[q] == ma) seen = 1; if (!seen) mols[nchg++] = ma; }
    md = md_last;
    double* Fd = s->evb_Fdiag + (size_t)d * n3;
    for (size_t k = 0; k < n3; ++k) Fd[k] = F1[k] - fr1[k];      /* everything but the reciprocal-space force, which is the diabat's own below */
    double ed = e1_terms - er1;
    memcpy(Qd, Q1, K3 * 8);
    evb_changed_terms(s, &P, nchg, mols, -1.0, &ed, Fd, Ftmp, screen, Qd, kk, &w);      /* out, in the principal topology */
    evb_changed_terms(s, &T, nchg, mols, +1.0, &ed, Fd, Ftmp, screen, Qd, kk, &w);      /* in, in the diabat's */
    /* calculate_reciprocal_space_pme: the diabat's grid through the transforms, then the force on every atom */
    double er_d = orc_convolve(s, Qd, thd);
    orc_scaled_coords(kk, K, N, T.xyz, usd);
#pragma omp parallel for num_threads(s->nthreads) schedule(static)
    for (int i = 0; i < N; ++i) { double f[3]; orc_gather_atom(s, thd, T.chg[i], &usd[3 * i], kk, f); for (int q = 0; q < 3; ++q) Fd[3 * i + q] = Fd[3 * i + q] + f[q]; }
    ed = ed + er_d;
    ed = ed + evb_repulsion(s, &T, Fd);
    ed = ed + s->ref_energy[T.mol_type[T.hydronium]];
    s->H[d][d] = ed;
    s->H[s->parent[d]][d] = evb_coupling(s, &T, md, ma, s->evb_Foff + (size_t)d * n3);
    topo_free(&T);
  }
  orc_scratch_free(&w);
  double e1 = e1_terms + evb_repulsion(s, &P, F1);
  e1 = e1 + s->ref_energy[P.mol_type[P.hydronium]];
  s->H[0][0] = e1;
  free(Q1); free(th1); free(fr1); free(us1); free(Ftmp); free(Qd); free(thd); free(usd); free(screen);
  return evb_diag_blend(s, F);
}
double orc_evb_force_energy_delta(OrcSys* s, double* F) {
  if (s->verlet_flag == 1 || !s->neighbor_list) { orc_verlet_build(s, s->nslist); orc_verlet_check(s, 1); s->verlet_flag = 0; }
  return orc_evb_force_energy_delta_noverlet(s, F);
}
int orc_evb_n_diabat(OrcSys* s) { return s->D; }
void orc_evb_get(OrcSys* s, int* plog, int* parent, double* H, double* eig, double* Fd, double* Fo) {
  int D = s->D; size_t n3 = 3 * (size_t)s->N;
  if (plog) for (int d = 0; d < D; ++d) for (int h = 0; h < s->max_chain; ++h) for (int q = 0; q < 5; ++q) plog[(d * s->max_chain + h) * 5 + q] = s->plog[d][h][q];
  if (parent) memcpy(parent, s->parent, D * 4);
  if (H) for (int i = 0; i < D; ++i) for (int j = 0; j < D; ++j) H[i * D + j] = s->H[i][j];
  if (eig) memcpy(eig, s->eigvec, D * 8);
  if (Fd) memcpy(Fd, s->evb_Fdiag, D * n3 * 8);
  if (Fo) memcpy(Fo, s->evb_Foff, D * n3 * 8);
}
int orc_evb_principal(OrcSys* s, int* nh) { if (nh) *nh = s->new_hydronium; return s->principal; }
/* src/ms_evb.f90:218-227: adopt the new principal diabat's topology permanently, rebuild the Verlet list */
void orc_evb_apply_principal(OrcSys* s) {
  if (s->new_hydronium == s->hydronium) return;
  OrcTopo P = orc_principal_topo(s);
  for (int h = 0; h < s->max_chain; ++h) { if (s->plog[s->principal][h][0] < 0) break;
    apply_hop(s, &P, P.hydronium, s->plog[s->principal][h][1], s->plog[s->principal][h][3], s->plog[s->principal][h][4]); }
  s->hydronium = P.hydronium;
  orc_verlet_build(s, s->nslist); orc_verlet_check(s, 1); s->verlet_flag = 0;
}
void orc_get_topology(OrcSys* s, int* atype, double* chg, int* mol_natom, int* mol_type, int* mol_atoms) {
  memcpy(atype, s->atype, s->N * 4); memcpy(chg, s->chg, s->N * 8); memcpy(mol_natom, s->mol_natom, s->M * 4);
  memcpy(mol_type, s->mol_type, s->M * 4); memcpy(mol_atoms, s->mol_atoms, (size_t)s->M * ORC_MAX_MOL_ATOMS * 4);
}
