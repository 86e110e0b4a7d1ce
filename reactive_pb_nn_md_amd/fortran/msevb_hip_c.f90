!==============================================================================
! msevb_hip_c : iso_c_binding interfaces of libmsevb_hip.so (include/msevb_hip.h).
! One interface per C entry point (checked by tests/test_host_side.py against the header); arrays are passed exactly as the reference
! stores them (1-based index values, column-major), so no copies are needed.
!==============================================================================
module msevb_hip_c
  use iso_c_binding
  implicit none
  integer, parameter :: MSEVB_N_ENERGY = 8
  interface
     integer(c_int) function msevb_create(ctx, device_id) bind(C, name="msevb_create")
       import :: c_ptr, c_int
       type(c_ptr), intent(out) :: ctx
       integer(c_int), value :: device_id
     end function
     integer(c_int) function msevb_destroy(ctx) bind(C, name="msevb_destroy")
       import :: c_ptr, c_int
       type(c_ptr), value :: ctx
     end function
     type(c_ptr) function msevb_last_error(ctx) bind(C, name="msevb_last_error")
       import :: c_ptr
       type(c_ptr), value :: ctx
     end function
     integer(c_int) function msevb_set_system(ctx, total_atoms, n_mole, box, rc, rv, alpha_sqrt, pme_grid, spline_order, spline_grid, erfc_grid) bind(C, name="msevb_set_system")
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: total_atoms, n_mole, pme_grid, spline_order, spline_grid, erfc_grid
       real(c_double), intent(in) :: box(3,3)
       real(c_double), value :: rc, rv, alpha_sqrt
     end function
     integer(c_int) function msevb_set_atoms(ctx, atom_type_index, charge, mass, mol_first_atom, mol_n_atom, mol_type_index) bind(C, name="msevb_set_atoms")
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       integer(c_int), intent(in) :: atom_type_index(*), mol_first_atom(*), mol_n_atom(*), mol_type_index(*)
       real(c_double), intent(in) :: charge(*), mass(*)
     end function
     integer(c_int) function msevb_set_forcefield(ctx, n_atom_type, ld, vdw_type, vdw_par, vdw_par14, bond_type, bond_par, atype_chg) bind(C, name="msevb_set_forcefield")
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: n_atom_type, ld
       integer(c_int), intent(in) :: vdw_type(*), bond_type(*)
       real(c_double), intent(in) :: vdw_par(*), vdw_par14(*), bond_par(*), atype_chg(*)
     end function
     integer(c_int) function msevb_set_angle_types(ctx, n, idx, par) bind(C, name="msevb_set_angle_types")
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: n
       integer(c_int), intent(in) :: idx(4,*)
       real(c_double), intent(in) :: par(2,*)
     end function
     integer(c_int) function msevb_set_dihedral_types(ctx, n, idx, par) bind(C, name="msevb_set_dihedral_types")
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: n
       integer(c_int), intent(in) :: idx(5,*)
       real(c_double), intent(in) :: par(6,*)
     end function
     integer(c_int) function msevb_set_molecule_type(ctx, i_type, n_atom, atom_type_index, n_bond, bond_list, n_angle, angle_list, n_dihedral, dihedral_list, &
          pair_exclusions, reactive_protons, reactive_basic) bind(C, name="msevb_set_molecule_type")
       import :: c_ptr, c_int
       type(c_ptr), value :: ctx
       integer(c_int), value :: i_type, n_atom, n_bond, n_angle, n_dihedral
       integer(c_int), intent(in) :: atom_type_index(*), bond_list(2,*), angle_list(3,*), dihedral_list(4,*), pair_exclusions(*), reactive_protons(*), reactive_basic(*)
     end function
     integer(c_int) function msevb_set_evb(ctx, n_molecule_type, ld_m, ld_i, acid, base, conj_pairs, proton_index, heavy_index, conj_atom, ref_energy, exch_atomic, &
          exch_proton, da_i, da_p, pa_i, pa_p, cp_i, cp_p, cp_t, hydronium_molecule, evb_max_chain) bind(C, name="msevb_set_evb")
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: n_molecule_type, ld_m, ld_i, hydronium_molecule, evb_max_chain
       integer(c_int), intent(in) :: acid(*), base(*), conj_pairs(*), proton_index(*), heavy_index(*), conj_atom(*), da_i(*), pa_i(*), cp_i(*), cp_t(*)
       real(c_double), intent(in) :: ref_energy(*), exch_atomic(*), exch_proton(*), da_p(*), pa_p(*), cp_p(*)
     end function
     integer(c_int) function msevb_finalize(ctx) bind(C, name="msevb_finalize")
       import :: c_ptr, c_int
       type(c_ptr), value :: ctx
     end function
     integer(c_int) function msevb_set_frozen_types(ctx, atype_freeze, n_atom_type) bind(C, name="msevb_set_frozen_types")
       import :: c_ptr, c_int
       type(c_ptr), value :: ctx
       integer(c_int), intent(in) :: atype_freeze(*)
       integer(c_int), value :: n_atom_type
     end function
     integer(c_int) function msevb_set_box(ctx, box3) bind(C, name="msevb_set_box")
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(in) :: box3(3)
     end function
     integer(c_int) function msevb_set_positions(ctx, xyz) bind(C, name="msevb_set_positions")
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(in) :: xyz(3,*)
     end function
     integer(c_int) function msevb_get_positions(ctx, xyz) bind(C, name="msevb_get_positions")
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(out) :: xyz(3,*)
     end function
     integer(c_int) function msevb_set_velocities(ctx, v) bind(C, name="msevb_set_velocities")
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(in) :: v(3,*)
     end function
     integer(c_int) function msevb_get_velocities(ctx, v) bind(C, name="msevb_get_velocities")
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(out) :: v(3,*)
     end function
     integer(c_int) function msevb_get_forces(ctx, f) bind(C, name="msevb_get_forces")
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(out) :: f(3,*)
     end function
     integer(c_int) function msevb_set_forces(ctx, f) bind(C, name="msevb_set_forces")
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(in) :: f(3,*)
     end function
     integer(c_int) function msevb_construct_verlet_list(ctx) bind(C, name="msevb_construct_verlet_list")
       import :: c_ptr, c_int
       type(c_ptr), value :: ctx
     end function
     integer(c_int) function msevb_update_verlet_displacements(ctx, flag) bind(C, name="msevb_update_verlet_displacements")
       import :: c_ptr, c_int
       type(c_ptr), value :: ctx
       integer(c_int), intent(out) :: flag
     end function
     integer(c_int) function msevb_calculate_total_force_energy(ctx, energies, force) bind(C, name="msevb_calculate_total_force_energy")
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(out) :: energies(*)
       real(c_double), intent(out) :: force(3,*)
     end function
     integer(c_int) function msevb_ms_evb_calculate_total_force_energy(ctx, potential_energy, force) bind(C, name="msevb_ms_evb_calculate_total_force_energy")
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(out) :: potential_energy
       real(c_double), intent(out) :: force(3,*)
     end function
     integer(c_int) function msevb_get_evb_state(ctx, n_diabat, proton_log, coupling_matrix, hamiltonian, eigenvector, principle_diabat, new_hydronium) bind(C, name="msevb_get_evb_state")
       import :: c_ptr, c_int
       type(c_ptr), value :: ctx
       integer(c_int), intent(out) :: n_diabat, principle_diabat, new_hydronium
       type(c_ptr), value :: proton_log, coupling_matrix, hamiltonian, eigenvector
     end function
     integer(c_int) function msevb_get_hop_log(ctx, n_hops, log, capacity, evb_calls, clear) bind(C, name="msevb_get_hop_log")
       import :: c_ptr, c_int, c_long
       type(c_ptr), value :: ctx
       integer(c_int), intent(out) :: n_hops
       integer(c_int), intent(out) :: log(3,*)
       integer(c_int), value :: capacity, clear
       integer(c_long), intent(out) :: evb_calls
     end function
     integer(c_int) function msevb_get_topology(ctx, atom_type_index, charge, mol_n_atom, mol_type_index, mol_atom_index, hydronium_molecule) bind(C, name="msevb_get_topology")
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       integer(c_int), intent(out) :: atom_type_index(*), mol_n_atom(*), mol_type_index(*), mol_atom_index(8,*), hydronium_molecule
       real(c_double), intent(out) :: charge(*)
     end function
     integer(c_int) function msevb_set_host_order(ctx, host_to_library) bind(C, name="msevb_set_host_order")
       import :: c_ptr, c_int
       type(c_ptr), value :: ctx
       integer(c_int), intent(in) :: host_to_library(*)
     end function
     integer(c_int) function msevb_md_integrate_atomic(ctx, n_steps, delta_t, use_evb, energies_out) bind(C, name="msevb_md_integrate_atomic")
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: n_steps, use_evb
       real(c_double), value :: delta_t
       real(c_double), intent(out) :: energies_out(2,*)
     end function
     integer(c_int) function msevb_calculate_kinetic_energy(ctx, ke) bind(C, name="msevb_calculate_kinetic_energy")
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(out) :: ke
     end function
     ! ---- multi-GPU (one process per GPU; the id is created on rank 0 and broadcast by the host, e.g. with MPI_Bcast of 128 bytes)
     integer(c_int) function msevb_comm_unique_id(id_bytes_128) bind(C, name="msevb_comm_unique_id")
       import :: c_int, c_signed_char
       integer(c_signed_char), intent(out) :: id_bytes_128(128)
     end function
     integer(c_int) function msevb_comm_init(ctx, id_bytes_128, rank, n_ranks) bind(C, name="msevb_comm_init")
       import :: c_ptr, c_int, c_signed_char
       type(c_ptr), value :: ctx
       integer(c_signed_char), intent(in) :: id_bytes_128(128)
       integer(c_int), value :: rank, n_ranks
     end function
     integer(c_int) function msevb_set_option(ctx, name, value) bind(C, name="msevb_set_option")
       import :: c_ptr, c_int, c_double, c_char
       type(c_ptr), value :: ctx
       character(kind=c_char), intent(in) :: name(*)      ! NUL-terminated, e.g. "overlap"//c_null_char
       real(c_double), value :: value
     end function
     integer(c_int) function msevb_get_info(ctx, name, value) bind(C, name="msevb_get_info")
       import :: c_ptr, c_int, c_double, c_char
       type(c_ptr), value :: ctx
       character(kind=c_char), intent(in) :: name(*)      ! NUL-terminated, e.g. "overlap"//c_null_char
       real(c_double), intent(out) :: value
     end function
     ! ---- the remaining entry points of include/msevb_hip.h ----
     integer(c_int) function msevb_set_stream(ctx, hip_stream) bind(C, name="msevb_set_stream")
       import :: c_ptr, c_int
       type(c_ptr), value :: ctx, hip_stream
     end function
     integer(c_int) function msevb_synchronize(ctx) bind(C, name="msevb_synchronize")
       import :: c_ptr, c_int
       type(c_ptr), value :: ctx
     end function
     integer(c_int) function msevb_get_table(ctx, name, out, n) bind(C, name="msevb_get_table")
       import :: c_ptr, c_int, c_double, c_char
       type(c_ptr), value :: ctx
       character(kind=c_char), intent(in) :: name(*)
       real(c_double), intent(out) :: out(*)
       integer(c_int), value :: n
     end function
     integer(c_int) function msevb_get_ewald_self(ctx, ewald_self) bind(C, name="msevb_get_ewald_self")
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(out) :: ewald_self
     end function
     type(c_ptr) function msevb_device_forces(ctx) bind(C, name="msevb_device_forces")
       import :: c_ptr
       type(c_ptr), value :: ctx
     end function
     type(c_ptr) function msevb_device_positions(ctx) bind(C, name="msevb_device_positions")
       import :: c_ptr
       type(c_ptr), value :: ctx
     end function
     integer(c_int) function msevb_get_verlet_flag(ctx, flag) bind(C, name="msevb_get_verlet_flag")
       import :: c_ptr, c_int
       type(c_ptr), value :: ctx
       integer(c_int), intent(out) :: flag
     end function
     integer(c_int) function msevb_get_verlet_list(ctx, verlet_point, neighbor_list, n_neighbors) bind(C, name="msevb_get_verlet_list")
       import :: c_ptr, c_int, c_long
       type(c_ptr), value :: ctx
       type(c_ptr), value :: verlet_point, neighbor_list      ! c_loc of integer arrays, or c_null_ptr to query the size
       integer(c_long), intent(out) :: n_neighbors
     end function
     integer(c_int) function msevb_zero_forces(ctx) bind(C, name="msevb_zero_forces")
       import :: c_ptr, c_int
       type(c_ptr), value :: ctx
     end function
     integer(c_int) function msevb_real_space_energy_force(ctx, e) bind(C, name="msevb_real_space_energy_force")
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(out) :: e(4)
     end function
     integer(c_int) function msevb_pme_reciprocal_space_energy_force(ctx, E_recip) bind(C, name="msevb_pme_reciprocal_space_energy_force")
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(out) :: E_recip
     end function
     integer(c_int) function msevb_intra_molecular_energy_force(ctx, e) bind(C, name="msevb_intra_molecular_energy_force")
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(out) :: e(3)
     end function
     integer(c_int) function msevb_get_pme_state(ctx, Q_grid, theta_conv_Q, xyz_scale, force_recip) bind(C, name="msevb_get_pme_state")
       import :: c_ptr, c_int
       type(c_ptr), value :: ctx
       type(c_ptr), value :: Q_grid, theta_conv_Q, xyz_scale, force_recip      ! c_loc of the host arrays, or c_null_ptr
     end function
     integer(c_int) function msevb_get_pme_indices(ctx, nearpt, splindex) bind(C, name="msevb_get_pme_indices")
       import :: c_ptr, c_int
       type(c_ptr), value :: ctx
       integer(c_int), intent(out) :: nearpt(3,*), splindex(3,6,*)
     end function
     integer(c_int) function msevb_grid_Q(ctx, Q, chg, xyz_scale, n_atom) bind(C, name="msevb_grid_Q")
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(out) :: Q(*)
       real(c_double), intent(in) :: chg(*), xyz_scale(3,*)
       integer(c_int), value :: n_atom
     end function
     integer(c_int) function msevb_modify_Q_grid(ctx, Q, chg, xyz_scale, n_atom, operation) bind(C, name="msevb_modify_Q_grid")
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(inout) :: Q(*)
       real(c_double), intent(in) :: chg(*), xyz_scale(3,*)
       integer(c_int), value :: n_atom, operation
     end function
     integer(c_int) function msevb_derivative_grid_Q(ctx, force, FQ, chg, xyz_scale, n_atom, kk, conv) bind(C, name="msevb_derivative_grid_Q")
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(out) :: force(3,*)
       real(c_double), intent(in) :: FQ(*), chg(*), xyz_scale(3,*), kk(3,3)
       integer(c_int), value :: n_atom
       real(c_double), value :: conv
     end function
     integer(c_int) function msevb_get_evb_element_force(ctx, i, j, force) bind(C, name="msevb_get_evb_element_force")
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: i, j
       real(c_double), intent(out) :: force(3,*)
     end function
     integer(c_int) function msevb_evb_partial_blend(ctx, eigvec, force) bind(C, name="msevb_evb_partial_blend")
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(in) :: eigvec(*)
       real(c_double), intent(out) :: force(3,*)
     end function
     integer(c_int) function msevb_comm_set_virtual(ctx, rank, n_ranks) bind(C, name="msevb_comm_set_virtual")
       import :: c_ptr, c_int
       type(c_ptr), value :: ctx
       integer(c_int), value :: rank, n_ranks
     end function
     integer(c_int) function msevb_comm_set_virtual_payload(ctx, payloads, n_doubles) bind(C, name="msevb_comm_set_virtual_payload")
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(in) :: payloads(*)
       integer(c_int), value :: n_doubles
     end function
     integer(c_int) function msevb_comm_get_payload(ctx, payloads, n_doubles) bind(C, name="msevb_comm_get_payload")
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(out) :: payloads(*)
       integer(c_int), value :: n_doubles
     end function
     integer(c_int) function msevb_evb_payload_size(n_diabat, n_ranks) bind(C, name="msevb_evb_payload_size")
       import :: c_int
       integer(c_int), value :: n_diabat, n_ranks
     end function
     integer(c_int) function msevb_evb_payload_pack(n_diabat, n_ranks, rank, hamiltonian, parent, energies8, out) bind(C, name="msevb_evb_payload_pack")
       import :: c_int, c_double, c_ptr
       integer(c_int), value :: n_diabat, n_ranks, rank
       real(c_double), intent(in) :: hamiltonian(*)
       integer(c_int), intent(in) :: parent(*)
       type(c_ptr), value :: energies8
       real(c_double), intent(out) :: out(*)
     end function
     integer(c_int) function msevb_evb_payload_unpack(n_diabat, n_ranks, all_payloads, parent, hamiltonian, energies8) bind(C, name="msevb_evb_payload_unpack")
       import :: c_int, c_double, c_ptr
       integer(c_int), value :: n_diabat, n_ranks
       real(c_double), intent(in) :: all_payloads(*)
       integer(c_int), intent(in) :: parent(*)
       real(c_double), intent(inout) :: hamiltonian(*)
       type(c_ptr), value :: energies8
     end function
     integer(c_int) function msevb_comm_allreduce_energies(ctx) bind(C, name="msevb_comm_allreduce_energies")
       import :: c_ptr, c_int
       type(c_ptr), value :: ctx
     end function
     integer(c_int) function msevb_comm_allreduce_forces(ctx) bind(C, name="msevb_comm_allreduce_forces")
       import :: c_ptr, c_int
       type(c_ptr), value :: ctx
     end function
     integer(c_int) function msevb_get_phase_times(ctx, n, names, ms, calls) bind(C, name="msevb_get_phase_times")
       import :: c_ptr, c_int, c_double, c_long
       type(c_ptr), value :: ctx
       integer(c_int), intent(out) :: n
       type(c_ptr), intent(out) :: names(16)      ! NUL-terminated C strings owned by the library
       real(c_double), intent(out) :: ms(16)
       integer(c_long), intent(out) :: calls(16)
     end function
     integer(c_int) function msevb_reset_phase_times(ctx) bind(C, name="msevb_reset_phase_times")
       import :: c_ptr, c_int
       type(c_ptr), value :: ctx
     end function
  end interface
contains
  ! the reference `stop`s on error (SURVEY.md 8b); the shim keeps that convention for the Fortran caller
  subroutine msevb_check(ctx, rc, where)
    type(c_ptr), intent(in) :: ctx
    integer(c_int), intent(in) :: rc
    character(*), intent(in) :: where
    character(kind=c_char), pointer :: msg(:)
    type(c_ptr) :: p
    integer :: i
    if (rc == 0) return
    write(*,*) "libmsevb_hip error ", rc, " in ", where
    p = msevb_last_error(ctx)
    if (c_associated(p)) then
       call c_f_pointer(p, msg, (/512/))
       do i = 1, 512
          if (msg(i) == c_null_char) exit
          write(*,'(A)',advance='no') msg(i)
       end do
       write(*,*)
    end if
    stop 1
  end subroutine
end module msevb_hip_c
