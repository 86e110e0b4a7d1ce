/*
 * msevb_hip.h - C ABI of libmsevb_hip.so, the MI355X (gfx950) implementation of the per-timestep
 * force/energy hot path of jmcdaniel43/Reactive_PB_NN_MD.
 *
 * The reference has no FFI: its boundary is the Fortran module-procedure surface called by
 * main_ms_evb.f90 / md_integration.f90.  Each entry point below replaces one
 * of those procedures (cited as file:line of /root/reference) and takes the SAME data the procedure
 * reads from the reference's derived types, as plain pointers:
 *   - indices are 1-based exactly as the reference's arrays hold them,
 *   - arrays are Fortran column-major as the reference stores them (xyz(3,N), box(3,3),
 *     atype_vdw_parameter(ld,ld,6) with leading dimension ld, Q(K,K,K)),
 * so that the iso_c_binding shim (reactive_pb_nn_md_amd/fortran/) passes its arrays without copies.
 * All functions return 0 on success, a negative code on error (msevb_last_error gives the text);
 * the library never aborts the process (the reference `stop`s, SURVEY.md 8b "error convention").
 * Every call is synchronous on return unless suffixed _async.  One context = one GPU = one host thread.
 */
#ifndef MSEVB_HIP_H
#define MSEVB_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct msevb_ctx msevb_ctx;

enum { MSEVB_OK = 0, MSEVB_ERR_ARG = -1, MSEVB_ERR_HIP = -2, MSEVB_ERR_STATE = -3, MSEVB_ERR_FFT = -4, MSEVB_ERR_LIMIT = -5,
       MSEVB_ERR_NUMERIC = -6, MSEVB_ERR_COMM = -7,
       MSEVB_ERR_TIMEOUT = -8 /* a device-side wait of the MS-EVB call ran into its wall-clock limit (not a statement about the coordinates) */ };

/* energies[] layout of msevb_calculate_total_force_energy (system_data%E_* of src/glob_v.f90:137-142) */
enum { MSEVB_E_POTENTIAL = 0, MSEVB_E_ELEC = 1, MSEVB_E_VDW = 2, MSEVB_E_BOND = 3, MSEVB_E_ANGLE = 4, MSEVB_E_DIHEDRAL = 5,
       MSEVB_E_RECIP = 6, MSEVB_N_ENERGY = 8 };

/* ---- life cycle ------------------------------------------------------------------------------ */
int msevb_create(msevb_ctx** ctx, int device_id);
int msevb_destroy(msevb_ctx* ctx);
const char* msevb_last_error(const msevb_ctx* ctx);
int msevb_set_stream(msevb_ctx* ctx, void* hip_stream);      /* NULL = library-owned stream (default) */
int msevb_synchronize(msevb_ctx* ctx);

/* ---- static data (replaces the reads of system_data / PME_data / verlet_list_data / global tables) */
/* scalars of system_data_type, PME_data_type, verlet_list_data_type (src/glob_v.f90:125-275) */
int msevb_set_system(msevb_ctx* ctx, int total_atoms, int n_mole, const double* box /*(3,3)*/, double real_space_cutoff,
                     double verlet_cutoff, double alpha_sqrt, int pme_grid, int spline_order, int spline_grid, int erfc_grid);
/* atom_data%{atom_type_index,charge,mass}, molecule_data(:)%{atom_index(1),n_atom,molecule_type_index} (src/glob_v.f90:157-176) */
int msevb_set_atoms(msevb_ctx* ctx, const int* atom_type_index, const double* charge, const double* mass,
                    const int* mol_first_atom, const int* mol_n_atom, const int* mol_type_index);
/* atype_* tables of module global_variables (src/glob_v.f90:320-337), leading dimension ld (= MAX_N_ATOM_TYPE in the reference) */
int msevb_set_forcefield(msevb_ctx* ctx, int n_atom_type, int ld, const int* atype_vdw_type, const double* atype_vdw_parameter,
                         const double* atype_vdw_parameter_14, const int* atype_bond_type, const double* atype_bond_parameter,
                         const double* atype_chg);
/* non-zero entries of atype_angle_type/parameter (idx(4,n) = i,j,k,type; par(2,n)) and of atype_dihedral_type/parameter
   (idx(5,n) = i,j,k,l,type; par(6,n)) - the 25^3 / 25^4 tensors of src/glob_v.f90:334-337 in sparse form */
int msevb_set_angle_types(msevb_ctx* ctx, int n, const int* idx, const double* par);
int msevb_set_dihedral_types(msevb_ctx* ctx, int n, const int* idx, const double* par);
/* molecule_type_data(i_type) (src/glob_v.f90:299-310) */
int msevb_set_molecule_type(msevb_ctx* ctx, int i_type, int n_atom, const int* atom_type_index, int n_bond, const int* bond_list,
                            int n_angle, const int* angle_list, int n_dihedral, const int* dihedral_list, const int* pair_exclusions,
                            const int* evb_reactive_protons, const int* evb_reactive_basic_atoms);
/* MS-EVB tables of module global_variables (src/glob_v.f90:54-120); *_interaction(ld_i,k), *_parameters(ld_i,k) with
   ld_i = max_interaction_type; evb_exchange_charge_proton(ld_m,ld_m); rows with interaction(r,1)==0 end a table */
int msevb_set_evb(msevb_ctx* ctx, int n_molecule_type, int ld_m, int ld_i, const int* evb_acid_molecule, const int* evb_basic_molecule,
                  const int* evb_conjugate_pairs, const int* evb_proton_index, const int* evb_heavy_acid_index,
                  const int* evb_conjugate_atom_index, const double* evb_reference_energy, const double* evb_exchange_charge_atomic,
                  const double* evb_exchange_charge_proton, const int* evb_donor_acceptor_interaction,
                  const double* evb_donor_acceptor_parameters, const int* evb_proton_acceptor_interaction,
                  const double* evb_proton_acceptor_parameters, const int* evb_diabat_coupling_interaction,
                  const double* evb_diabat_coupling_parameters, const int* evb_diabat_coupling_type, int hydronium_molecule,
                  int evb_max_chain);
/* initialize_energy_force minus its first force call (src/initialize_routines.f90:173-264): Ewald self term, B-spline /
   erfc / Tang-Toennies tables, CB array (periodic_box_change, src/pme.f90:634), FFT plans; uploads everything */
int msevb_finalize(msevb_ctx* ctx);
/* read back a table built by msevb_finalize: "B6_spline","B5_spline","erfc_table","ewaldscale_table",
   "Tang_Toennies_table","dTang_Toennies_table","CB"; returns the number of doubles (n = capacity of out) */
int msevb_get_table(msevb_ctx* ctx, const char* name, double* out, int n);
int msevb_get_ewald_self(msevb_ctx* ctx, double* ewald_self);

/* ---- per-step data ---------------------------------------------------------------------------- */
int msevb_set_positions(msevb_ctx* ctx, const double* xyz /*(3,N) host*/);
/* atype_freeze (src/glob_v.f90, used at src/md_integration.f90:89-169,:473,:512): 1 = atoms of this type are not integrated */
int msevb_set_frozen_types(msevb_ctx* ctx, const int* atype_freeze /*(n_atom_type)*/, int n_atom_type);
/* periodic_box_change (src/pme.f90:634-664): new orthorhombic box lengths; CB, the reciprocal lattice and the Verlet list are rebuilt for them */
int msevb_set_box(msevb_ctx* ctx, const double* box3);
int msevb_get_positions(msevb_ctx* ctx, double* xyz);
int msevb_set_velocities(msevb_ctx* ctx, const double* velocity /*(3,N) host*/);
int msevb_get_velocities(msevb_ctx* ctx, double* velocity);
int msevb_get_forces(msevb_ctx* ctx, double* force /*(3,N) host*/);
/* atom_data%force as the host holds it (a host that changed its arrays between two steps of msevb_md_integrate_atomic: the first
   half-kick of velocity Verlet, src/md_integration.f90:478, uses the forces of the previous force call) */
int msevb_set_forces(msevb_ctx* ctx, const double* force /*(3,N) host*/);
/* Atom order of the CALLER's arrays.  The reference re-packs its host arrays when the principal diabat changes (the proton's slot moves to the
   acceptor's block, src/ms_evb.f90:2677-2836, called from :218-227); the library never moves atoms.  host_to_library(N), 1-based: entry i = the
   library's index of the caller's atom i (NULL: same order).  From then on every (3,N) array that crosses the ABI - positions, velocities,
   forces, in either direction - is permuted on the device; tables, plans, the Verlet list and its capacity stay as they are. */
int msevb_set_host_order(msevb_ctx* ctx, const int* host_to_library);
void* msevb_device_forces(msevb_ctx* ctx);      /* device pointer to force(3,N), valid until destroy */
void* msevb_device_positions(msevb_ctx* ctx);

/* ---- hot path --------------------------------------------------------------------------------- */
/* construct_verlet_list (src/general_routines.f90:1339) + the initialising call of update_verlet_displacements (:1259).
   The list is built on the device from a cell grid; as a SET of pairs {i<j, different molecule, r<verlet_cutoff} it equals
   the reference's neighbor_list. */
int msevb_construct_verlet_list(msevb_ctx* ctx);
/* update_verlet_displacements (src/general_routines.f90:1259): flag=1 when the two largest displacements exceed the skin */
int msevb_update_verlet_displacements(msevb_ctx* ctx, int* flag_verlet_list);
/* verlet_list_data%flag_verlet_list (src/glob_v.f90:236-250) as the last force call left it: 1 = the NEXT force call rebuilds the
   list (src/total_energy_forces.f90:30-41).  Read-only. */
int msevb_get_verlet_flag(msevb_ctx* ctx, int* flag_verlet_list);
/* optional read-back of the half list in the reference's layout (verlet_point(N+1), neighbor_list, 1-based); order within
   an atom's list is ascending j.  Pass NULL to query sizes only. */
int msevb_get_verlet_list(msevb_ctx* ctx, int* verlet_point, int* neighbor_list, long* n_neighbors);

/* stage calls: each ADDS its forces to the device force array and returns its energies (caller zeroes with msevb_zero_forces) */
int msevb_zero_forces(msevb_ctx* ctx);
/* real_space_energy_force (src/pair_int_real_space.f90:60): Verlet pair loop + intra-molecular non-bonded.
   e[0..3] = E_elec(pair), E_vdw(pair), E_elec(intra), E_vdw(intra) */
int msevb_real_space_energy_force(msevb_ctx* ctx, double* e);
/* pme_reciprocal_space_energy_force (src/pme.f90:28): spread (grid_Q :184), FFT, xCB, inverse FFT, energy (:128), gather
   (derivative_grid_Q :346).  Keeps Q_grid, theta_conv_Q and force_recip on the device. */
int msevb_pme_reciprocal_space_energy_force(msevb_ctx* ctx, double* E_recip);
/* intra_molecular_energy_force (src/intra_bonded_interactions.f90:17): e[0..2] = E_bond, E_angle, E_dihedral */
int msevb_intra_molecular_energy_force(msevb_ctx* ctx, double* e);
/* calculate_total_force_energy (src/total_energy_forces.f90:19) including its Verlet-list freshness logic (:30-39).
   force may be NULL (forces stay on the device). */
int msevb_calculate_total_force_energy(msevb_ctx* ctx, double* energies /*[MSEVB_N_ENERGY]*/, double* force);
/* PME intermediates for parity tests: Q_grid(K,K,K), theta_conv_Q(K,K,K), scaled coordinates (3,N), force_recip(3,N);
   any pointer may be NULL */
int msevb_get_pme_state(msevb_ctx* ctx, double* Q_grid, double* theta_conv_Q, double* xyz_scale, double* force_recip);
/* grid indices of the spread (bit-exact contract): nearpt(3,N) = floor(xyz_scale), splindex(3,6,N) = B6 table rows (1-based) */
int msevb_get_pme_indices(msevb_ctx* ctx, int* nearpt, int* splindex);

/* The PME stages at the reference's own granularity, for a host that switches one procedure at a time.  They take what the reference
   procedures take: SCALED coordinates (create_scaled_direct_coordinates, src/general_routines.f90:497), charges and HOST grids; the
   force path above never uses them (it keeps its grids on the device).
   grid_Q (src/pme.f90:184): Q(K,K,K) = sum of the n_atom B-spline stencils */
int msevb_grid_Q(msevb_ctx* ctx, double* Q /*(K,K,K) out*/, const double* chg /*(n_atom)*/, const double* xyz_scale /*(3,n_atom)*/, int n_atom);
/* modify_Q_grid (src/pme.f90:275): Q += (operation > 0) or -= (operation < 0) the stencils of n_atom atoms; charges with |q| <= 1e-6 skipped (:296) */
int msevb_modify_Q_grid(msevb_ctx* ctx, double* Q /*(K,K,K) inout*/, const double* chg, const double* xyz_scale, int n_atom, int operation);
/* derivative_grid_Q (src/pme.f90:346) for n_atom atoms in one call: force(3,n_atom) = -K kk^T sum fac * FQ * conv, FQ = theta_conv_Q(K,K,K),
   kk(3,3) column-major as construct_reciprocal_lattice_vector (src/general_routines.f90:473) returns it */
int msevb_derivative_grid_Q(msevb_ctx* ctx, double* force /*(3,n_atom) out*/, const double* FQ, const double* chg, const double* xyz_scale,
                            int n_atom, const double* kk, double conv_e2A_kJmol);

/* ms_evb_calculate_total_force_energy (src/ms_evb.f90:181): principal diabat, diabat enumeration, per-diabat delta
   updates, EVB Hamiltonian, Jacobi diagonalisation, Hellmann-Feynman forces, principal-diabat switch. */
int msevb_ms_evb_calculate_total_force_energy(msevb_ctx* ctx, double* potential_energy, double* force);
/* results of the last EVB call: number of diabats, evb_diabat_proton_log(D,max_chain,5) (1-based, -1 padded, C order
   [d][hop][5]), evb_diabat_coupling_matrix(D), evb_hamiltonian(D,D) upper triangle (C order), ground-state eigenvector,
   principal diabat (1-based) and the hydronium molecule it implies.  Any pointer may be NULL. */
int msevb_get_evb_state(msevb_ctx* ctx, int* n_diabat, int* proton_log, int* coupling_matrix, double* hamiltonian,
                        double* eigenvector, int* principle_diabat, int* new_hydronium_molecule);
/* stored force of Hamiltonian element (i,j), 1-based, i<=j (evb_forces_store(:,:,evb_forces_lookup_index(i,j)),
   src/ms_evb.f90:2523), principal atom order */
int msevb_get_evb_element_force(msevb_ctx* ctx, int i, int j, double* force);
/* principal-diabat switches (proton hops) since the log was last cleared: log(3, n_hops) = {number of the EVB force call in which the hop
   happened (1-based over the context's life; *evb_calls = the current count), old hydronium molecule, new hydronium molecule}.  The reference
   writes each hop with its own step (src/ms_evb.f90:331-335); a caller that integrates blocks of steps on the device reads them here. */
int msevb_get_hop_log(msevb_ctx* ctx, int* n_hops, int* log /*(3,capacity) or NULL*/, int capacity, long* evb_calls /*or NULL*/, int clear);
/* current topology (changes when the principal diabat changes, src/ms_evb.f90:218-227) */
int msevb_get_topology(msevb_ctx* ctx, int* atom_type_index, double* charge, int* mol_n_atom, int* mol_type_index,
                       int* mol_atom_index /*(8,n_mole) 1-based, 0 padded*/, int* hydronium_molecule);

/* md_integrate_atomic, NVE branch (src/md_integration.f90:438): n_steps of velocity Verlet with positions, velocities
   and forces resident on the device.  use_evb selects the force call as the reference's compile-time switch does
   (src/md_integration.f90:495-500).  energies_out: per step {potential, kinetic} (may be NULL). */
int msevb_md_integrate_atomic(msevb_ctx* ctx, int n_steps, double delta_t, int use_evb, double* energies_out);
int msevb_calculate_kinetic_energy(msevb_ctx* ctx, double* kinetic_energy);   /* src/total_energy_forces.f90:106 */

/* ---- multi-GPU: diabats sharded over ranks ------------------------------ */
/* rank r owns diabats d >= 2 with (d-2) mod n_ranks == r; the principal diabat is computed by every rank */
int msevb_comm_unique_id(void* id_bytes_128);                                       /* ncclGetUniqueId */
int msevb_comm_init(msevb_ctx* ctx, const void* id_bytes_128, int rank, int n_ranks); /* ncclCommInitRank (RCCL) */
int msevb_comm_set_virtual(msevb_ctx* ctx, int rank, int n_ranks);   /* shard without a communicator (tests) */
int msevb_comm_set_virtual_payload(msevb_ctx* ctx, const double* payloads, int n_doubles);   /* virtual rank: the n_ranks payloads (msevb_evb_payload_pack) an all-gather would deliver - the rank then runs the device-side solve and the pre-enqueued blend of a rank with a communicator (src/ms_evb.f90:242-330 on the gathered Hamiltonian); tests, tools/virtual_rank_times.py */
int msevb_comm_get_payload(msevb_ctx* ctx, double* payloads, int n_doubles);   /* the all-gather buffer as it is on the device (this rank's slot: what its energy-fetch kernel assembled, src/ms_evb.f90:375-486 for its diabats); tests play the all-gather between virtual ranks with it */
/* payload of the per-step all-gather as context-free functions (no GPU): P = msevb_evb_payload_size doubles per rank = {H_dd, H_pd} of the
   rank's diabats, H'_11, eight principal energy slots; hamiltonian is D x D row-major (upper triangle), parent[d] 0-based.  Used by the
   library's own RCCL exchange and by the CPU (gloo) test of the protocol */
int msevb_evb_payload_size(int n_diabat, int n_ranks);
int msevb_evb_payload_pack(int n_diabat, int n_ranks, int rank, const double* hamiltonian, const int* parent, const double* energies8 /*or NULL*/, double* out);
int msevb_evb_payload_unpack(int n_diabat, int n_ranks, const double* all_payloads, const int* parent, double* hamiltonian, double* energies8 /*or NULL*/);
int msevb_comm_allreduce_energies(msevb_ctx* ctx); /* principal energy slots when the principal real space is sharded (internal) */
int msevb_comm_allreduce_forces(msevb_ctx* ctx);   /* sum of the ranks' partial Hellmann-Feynman forces (called by the EVB force call itself) */
int msevb_evb_partial_blend(msevb_ctx* ctx, const double* eigvec, double* force);   /* this rank's share of the blend for a given eigenvector (src/ms_evb.f90:293-309; tests of the sharding) */

/* ---- instrumentation --------------------------------------------------------------------------- */
/* average device time (ms) of each phase over the calls since the last reset, names as the reference's debug trace */
int msevb_get_phase_times(msevb_ctx* ctx, int* n, const char** names, double* ms, long* calls);
int msevb_reset_phase_times(msevb_ctx* ctx);
/* Switches of the library (0 / 1 unless a unit is given); the environment variable MSEVB_OPTIONS="name=value,..." applies the same calls at
   msevb_finalize.  Defaults are the measured best on MI355X; every alternative is kept because a test or DESIGN.md / HISTORY.md compares against it.
     physics / integrator : ensemble_nvt (0), friction_coeff (1/ps), temperature (K), langevin_seed; evb_element_stores (0: keep F_dd / F_pd of every
                            Hamiltonian element as the reference's evb_forces_store does), evb_recip_linear (1: diabats' reciprocal space by linear
                            response, 0: one transform pair per diabat as src/ms_evb.f90:1962 does)
     pair loop            : pair_tiles (-1 auto), pair_flat (1: flat per-cell list, pair_flat.hip), pair_taylor (1: one-row Ewald table), pair_prune (1),
                            prune_delta (min(0.75 A, (verlet_cutoff - real_space_cutoff) / 2)), pair_split (0), pair_threads (1024), prune_freeze (diagnostic), nbr_tiles (1), nbr_cap (row stride)
     PME                  : fft_own (1: own 64^3 / 128^3 transforms, 0: rocFFT), spread_tiles (1), pme_weightless (1: no per-atom weight store),
                            pme_side_weights (0), spread_mfma (0), spread_bricks (0), gather_mfma (0), pme_tile_threads (512), fft_batch
     schedule             : overlap (1: four streams), graphs (1: front end replayed as a hipGraph), evb_speculate (1), evb_publish_spin (1),
                            evb_late_mol_join (1), evb_single_join (1), evb_defer_fold (1), evb_pme_split (1), evb_defer_molecules (0),
                            evb_blend_aside (1), evb_fft_patch (1), step_fuse_refresh (1), evb_region_main (0), evb_region_aside (0), evb_pair_first (0), evb_device_solve (1: squaring + Jacobi fallback on the device, 2: Jacobi on the device, 0: host),
                            evb_coupling_blocks (0 = auto), pin_host_arrays (0),
                            pair_self_mode (1), pme_gather_threads (0 = auto: 256 while no cell holds more atoms, else pme_tile_threads)
     multi-GPU            : shard_principal (set by msevb_comm_init)
     measurement          : phase_timing (0), phase_timing_stride (8) */
int msevb_set_option(msevb_ctx* ctx, const char* name, double value);
/* Which kernels serve this context (tests assert that a fixture reaches the production kernels): "pair_kernel" (after a list build: 0 = k_pair on
   the full fixed-stride list, boxes under 4 cells per axis; 1 = k_pair_half; 2 = k_pair_flat), "list_build_tiles", "pme_tiles" (tiled spread / gather),
   "pme_weightless", "fft_own", "n_cells", "cells_per_axis_min", "n_diabat" */
int msevb_get_info(msevb_ctx* ctx, const char* name, double* value);

#ifdef __cplusplus
}
#endif
#endif
