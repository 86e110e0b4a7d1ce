/*
 * TEST INFRASTRUCTURE ONLY (oracle/) - CPU restatement of the reference hot path.
 *
 * Plain-C, double precision, function-for-function restatement of the reference's
 * per-timestep force/energy path.  It exists to CHECK the HIP
 * product and to be timed as the "port" CPU baseline of bench.py; nothing in
 * reactive_pb_nn_md_amd/ may include, link or call it.
 * Parity pin: tests/test_oracle_vs_reference.py compares every routine here with the
 * golden vectors in tests/golden/, which were produced by the real reference
 * (oracle/build_ref.sh + oracle/make_golden.py).
 *
 * All indices in this API are 0-based; arrays are C row-major
 * (xyz[N][3], vdw_par[nt][nt][6], Q[n3][n2][n1] == reference Q(n1+1,n2+1,n3+1)).
 */
#ifndef MSEVB_ORACLE_H
#define MSEVB_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_MOL_ATOMS 8
#define ORC_MAX_TYPES 32
#define ORC_MAX_MOLTYPES 10
#define ORC_TT_GRID 1000
#define ORC_TT_MAX 50.0
#define ORC_MAX_INTERACTION 15
#define ORC_MAX_DIABATS 80
#define ORC_MAX_CHAIN 4

typedef struct OrcSys OrcSys;

OrcSys* orc_create(int n_atom, int n_mole, int n_atom_type, int n_mole_type, int pme_grid,
                   const double box[3], double rc, double rv, double alpha);
void orc_destroy(OrcSys*);
void orc_set_threads(OrcSys*, int nthreads);
void orc_set_atoms(OrcSys*, const int* atype, const double* chg, const double* mass,
                   const int* mol_first, const int* mol_natom, const int* mol_type);
void orc_set_vdw(OrcSys*, const int* vdw_type, const double* vdw_par, const double* vdw_par14);
void orc_set_bonded(OrcSys*, const int* bond_type, const double* bond_par,
                    int n_ang, const int* ang_idx /*[n][4] i,j,k,type*/, const double* ang_par /*[n][2]*/,
                    int n_dih, const int* dih_idx /*[n][5]*/, const double* dih_par /*[n][6]*/);
void orc_set_moltype(OrcSys*, int it, int natom, const int* atype, int nb, const int* bonds,
                     int na, const int* angles, int nd, const int* dihedrals, const int* excl,
                     const int* reactive_protons, const int* reactive_basic);
void orc_set_xyz(OrcSys*, const double* xyz);
void orc_get_xyz(OrcSys*, double* xyz);

/* init-time tables: src/initialize_routines.f90:193-264, src/pme.f90:505-625 */
void orc_build_tables(OrcSys*);
int  orc_get_table(OrcSys*, const char* name, double* out, int n); /* B6,B5,erfc,ewaldscale,tt,dtt,CB; returns length */
double orc_ewald_self(OrcSys*);

/* Verlet list: src/general_routines.f90:1408 (grid build, identical traversal order) */
int  orc_verlet_build(OrcSys*, int nslist);   /* returns number of half-list entries */
void orc_verlet_get(OrcSys*, int* verlet_point /*[N+1] 0-based offsets*/, int* neighbor_list);
int  orc_verlet_check(OrcSys*, int initialize); /* src/general_routines.f90:1259, returns flag */
int  orc_verlet_flag(OrcSys*);                  /* flag_verlet_list after the last force call of orc_md_step */

/* stage routines; F is ACCUMULATED into (caller zeroes); E = {E_elec, E_vdw, E_bond, E_angle, E_dihedral} */
void orc_pair_real_space(OrcSys*, double* E, double* F);   /* src/pair_int_real_space.f90:135 */
void orc_intra_nonbonded(OrcSys*, double* E, double* F);   /* src/pair_int_real_space.f90:101-113, :386 */
void orc_pme_recip(OrcSys*, double* E, double* F);         /* src/pme.f90:28 (stores Q, theta, force_recip) */
void orc_bonded(OrcSys*, double* E, double* F);            /* src/intra_bonded_interactions.f90:17 */
double orc_total(OrcSys*, double* E, double* F);           /* src/total_energy_forces.f90:19 (zeroes E,F) */
void orc_get_grid(OrcSys*, double* Q, double* theta, double* xyz_scale, double* kk);
void orc_get_force_recip(OrcSys*, double* F);

/* integrator: src/md_integration.f90:438 (NVE branch); returns potential energy */
double orc_md_step(OrcSys*, double* vel, double* F, double dt, int evb);
double orc_kinetic_energy(OrcSys*, const double* vel);

/* ---- MS-EVB (src/ms_evb.f90) ---- */
void orc_set_evb(OrcSys*, const int* acid, const int* base, const int* conj_mol, const int* proton_type,
                 const int* heavy_type, const int* conj_atype, const double* ref_energy,
                 const double* exch_atomic, const double* exch_proton /*[nmt][nmt]*/,
                 int n_da, const int* da_types /*[n][3]*/, const double* da_par /*[n][6]*/,
                 int n_pa, const int* pa_types /*[n][2]*/, const double* pa_par /*[n][5]*/,
                 int n_cp, const int* cp_types /*[n][3]*/, const int* cp_func, const double* cp_par /*[n][10]*/,
                 int hydronium_mol, int max_chain);
/* full EVB force call (src/ms_evb.f90:181): returns adiabatic energy, fills F (principal atom order).
   Every diabat is evaluated FROM SCRATCH in its own topology (independent of the product's delta scheme). */
double orc_evb_force_energy(OrcSys*, double* F);
/* the same call in the reference's DELTA-UPDATE formulation (ms_evb_diabat_force_energy, src/ms_evb.f90:1421-1560; calculate_reciprocal_space_pme :1962-2090):
   principal diabat in full, every further diabat as the change of its donor / acceptor molecules + one reciprocal-space pass of its own.  Same results as
   orc_evb_force_energy to rounding (tests); what bench.py times as the CPU baseline. */
double orc_evb_force_energy_delta(OrcSys*, double* F);
double orc_evb_repulsion_principal(OrcSys*, double* F);      /* src/ms_evb.f90:2259-2478 on the principal topology */
void orc_evb_rep_switch(double r, double rs, double rc, double* sw, double* dsw);      /* :2484 */
int  orc_evb_n_diabat(OrcSys*);
void orc_evb_get(OrcSys*, int* proton_log /*[D][max_chain][5]*/, int* parent /*[D]*/, double* H /*[D][D]*/,
                 double* eigvec /*[D]*/, double* F_diag /*[D][N][3]*/, double* F_off /*[D][N][3]*/);
int  orc_evb_principal(OrcSys*, int* new_hydronium_mol);
void orc_evb_apply_principal(OrcSys*);  /* src/ms_evb.f90:218-227: re-topologise to the new principal diabat */
void orc_get_topology(OrcSys*, int* atype, double* chg, int* mol_natom, int* mol_type, int* mol_atoms /*[M][8]*/);

#ifdef __cplusplus
}
#endif
#endif
