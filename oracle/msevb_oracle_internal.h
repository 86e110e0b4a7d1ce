/* TEST INFRASTRUCTURE ONLY (oracle/) - private definitions shared by msevb_oracle.c / msevb_oracle_evb.c */
#ifndef MSEVB_ORACLE_INTERNAL_H
#define MSEVB_ORACLE_INTERNAL_H
#include <math.h>
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "msevb_oracle.h"

typedef struct {
  int natom, nb, na, nd;
  int atype[ORC_MAX_MOL_ATOMS];
  int bonds[2 * 16], angles[3 * 16], dihedrals[4 * 16];
  int excl[ORC_MAX_MOL_ATOMS][ORC_MAX_MOL_ATOMS];
  int reactive_protons[ORC_MAX_MOL_ATOMS], reactive_basic[ORC_MAX_MOL_ATOMS];
} OrcMolType;

/* a molecular topology over a FIXED atom numbering (the principal diabat's): a diabat differs from the principal
   one by atom types/charges of donor+acceptor molecules, by which molecule owns the moved proton(s), and by the
   periodic image of a moved proton (make_molecule_whole). */
typedef struct {
  int* atype; double* chg; double* xyz; int* mol_of; int* mol_type; int* mol_natom; int* mol_atoms; double* r_com;
  int hydronium;
} OrcTopo;

typedef struct {
  int cap;
  double *dr, *dr2, *qq, *f; int* idx;
  double *dr_lj, *dr2_lj, *par_lj, *f_lj; int* idx_lj;
  double *dr_sp, *dr2_sp, *par_sp, *f_sp; int* idx_sp;
} OrcScratch;

struct OrcSys {
  int N, M, nt, nmt, K, order, spline_grid, erfc_grid, nthreads;
  double box[3], rc, rv, alpha, pi, pi_sqrt, conv;
  int *atype; double *chg, *mass, *xyz; int* mol_of;
  int *mol_type, *mol_natom, *mol_atoms; double* r_com;
  int* vdw_type; double *vdw_par, *vdw_par14; int* bond_type; double* bond_par;
  int n_ang; int* ang_idx; double* ang_par; int n_dih; int* dih_idx; double* dih_par;
  OrcMolType mt[ORC_MAX_MOLTYPES];
  double *B6, *B5, *erfc_t, *scale_t, erfc_dx; double tt[ORC_TT_GRID][4], dtt[ORC_TT_GRID][4];
  double *CB, *Q, *theta, *xyz_scale, *force_recip, kk[3][3], E_recip, ewald_self;
  int *verlet_point, *neighbor_list, n_neighbors, nslist, verlet_flag; double *vstore, *vdisp;
  /* EVB tables */
  double atype_chg[ORC_MAX_TYPES];
  int atype_freeze[ORC_MAX_TYPES];   /* src/glob_v.f90:323 */
  int acid[ORC_MAX_MOLTYPES], base[ORC_MAX_MOLTYPES], conj_mol[ORC_MAX_MOLTYPES], proton_type[ORC_MAX_MOLTYPES], heavy_type[ORC_MAX_MOLTYPES];
  int conj_atype[ORC_MAX_TYPES]; double ref_energy[ORC_MAX_MOLTYPES], exch_atomic[ORC_MAX_TYPES], exch_proton[ORC_MAX_MOLTYPES][ORC_MAX_MOLTYPES];
  int n_da, da_types[ORC_MAX_INTERACTION][3]; double da_par[ORC_MAX_INTERACTION][6];
  int n_pa, pa_types[ORC_MAX_INTERACTION][2]; double pa_par[ORC_MAX_INTERACTION][5];
  int n_cp, cp_types[ORC_MAX_INTERACTION][3], cp_func[ORC_MAX_INTERACTION]; double cp_par[ORC_MAX_INTERACTION][10];
  int hydronium, max_chain;
  /* EVB results of the last call */
  int D, plog[ORC_MAX_DIABATS][ORC_MAX_CHAIN][5], parent[ORC_MAX_DIABATS], principal, new_hydronium;
  double H[ORC_MAX_DIABATS][ORC_MAX_DIABATS], eigvec[ORC_MAX_DIABATS];
  double *evb_Fdiag, *evb_Foff; int evb_cap;
};

double orc_bspline(double u, int n);
void orc_recip_lattice(const double box[3], double vol, double kk[3][3]);
void orc_scaled_coords(const double kk[3][3], int K, int n, const double* xyz, double* u);
void orc_spread_atom(const OrcSys* s, double* Q, double q, const double* u, double sign);
void orc_gather_atom(const OrcSys* s, const double* theta, double q, const double* u, const double kk[3][3], double* force);
void orc_fft3d(int K, double* re, double* im, int sign);
double orc_convolve(OrcSys* s, const double* Q, double* theta);
void orc_k_ewald(const OrcSys* s, int n, const double* dr, const double* dr2, const double* qq, double* E, double* f);
void orc_k_lj(int n, const double* dr, const double* dr2, const double* par, double* E, double* f);
void orc_k_sapt(const OrcSys* s, int n, const double* dr, const double* dr2, const double* par, double* E, double* f);
void orc_atom_vs_list(const OrcSys* s, const OrcTopo* tp, int i, int nl, const int* list, const unsigned char* screen,
                      double* E_elec, double* E_vdw, double* F, OrcScratch* w);
void orc_scratch_reserve(OrcScratch* w, int n);
void orc_scratch_free(OrcScratch* w);
OrcTopo orc_principal_topo(OrcSys* s);
void orc_pair_topo(OrcSys* s, const OrcTopo* tp, const int* vp, const int* nl, double* E, double* F);
int orc_verlet_build_topo(OrcSys* s, const OrcTopo* tp, int ns, int* verlet_point, int** list_out);
void orc_intra_nb_mol(const OrcSys* s, const OrcTopo* tp, int m, double* E_elec, double* E_vdw, double* F);
void orc_intra_nb_topo(OrcSys* s, const OrcTopo* tp, double* E, double* F);
void orc_pme_topo(OrcSys* s, const OrcTopo* tp, double* E, double* F);
void orc_bonded_mol(const OrcSys* s, const OrcTopo* tp, int m, double* Eb, double* Ea, double* Ed, double* F);
void orc_bonded_topo(OrcSys* s, const OrcTopo* tp, double* E, double* F);
double orc_total_topo(OrcSys* s, const OrcTopo* tp, const int* vp, const int* nl, double* E, double* F);
void orc_update_com(OrcSys* s);
double orc_evb_force_energy_noverlet(OrcSys* s, double* F);
double orc_evb_force_energy_delta_noverlet(OrcSys* s, double* F);
void orc_set_atype_chg(OrcSys* s, const double* atype_chg);
void orc_set_atype_freeze(OrcSys* s, const int* atype_freeze);
#endif
