!==============================================================================
! GPU-backed versions of the reference's STAGE procedures, with the reference's
! own names and argument lists (SURVEY.md 8b "hot-path signatures to preserve"):
!
!   real_space_energy_force            src/pair_int_real_space.f90:60
!   pme_reciprocal_space_energy_force  src/pme.f90:28
!   grid_Q / modify_Q_grid / derivative_grid_Q      src/pme.f90:184 / :275 / :346
!   intra_molecular_energy_force       src/intra_bonded_interactions.f90:17
!   construct_verlet_list / update_verlet_displacements   src/general_routines.f90:1339 / :1259
!   md_integrate_atomic                src/md_integration.f90:438
!
! A caller switches one procedure at a time by importing it from here instead of
! from the reference's module, e.g.
!     use msevb_hip_stages, only : real_space_energy_force
! (the reference modules stay in the link for their cold parts; the names only
! clash if both are imported into one scope).  Semantics are the reference's:
! forces and energies are ADDED to atom_data%force / system_data%E_*.
! The fused force call (module total_energy_forces) does not go through these.
!==============================================================================
module msevb_hip_stages
  use global_variables
  use msevb_hip_c
  use iso_c_binding
  use total_energy_forces, only : msevb_hip_context, msevb_hip_prepare
  implicit none
  private
  public :: real_space_energy_force, pme_reciprocal_space_energy_force, grid_Q, modify_Q_grid, derivative_grid_Q, derivative_grid_Q_all, &
            intra_molecular_energy_force, construct_verlet_list, update_verlet_displacements, md_integrate_atomic
  logical, save :: list_built = .false.

contains

  function need_ctx(who) result(ctx)
    character(*), intent(in) :: who
    type(c_ptr) :: ctx
    ctx = msevb_hip_context()
    if (.not. c_associated(ctx)) then
       write(*,*) who, ": the GPU context does not exist yet (call calculate_total_force_energy, construct_verlet_list or real_space_energy_force first)"
       stop 1
    end if
  end function need_ctx

  !-- adds the device force array to atom_data%force (the stage kernels accumulate into a zeroed device array)
  subroutine add_device_forces(ctx, atom_data, n)
    type(c_ptr), intent(in) :: ctx
    type(atom_data_type), intent(inout) :: atom_data
    integer, intent(in) :: n
    real(c_double), allocatable :: f(:,:)
    allocate(f(3,n))
    call msevb_check(ctx, msevb_get_forces(ctx, f), "msevb_get_forces")
    atom_data%force(:,1:n) = atom_data%force(:,1:n) + f
  end subroutine add_device_forces

  !*********** src/general_routines.f90:1339 (the list itself stays on the device; verlet_list_data only carries the flag)
  subroutine construct_verlet_list( verlet_list_data, atom_data, molecule_data, total_atoms, box, xyz_to_box_transform )
    type(verlet_list_data_type), intent(inout) :: verlet_list_data
    type(atom_data_type), intent(in) :: atom_data
    type(molecule_data_type), dimension(:), intent(in) :: molecule_data
    integer, intent(in) :: total_atoms
    real*8, dimension(:,:), intent(in) :: box, xyz_to_box_transform
    type(c_ptr) :: ctx
    ctx = need_ctx("construct_verlet_list")
    call msevb_check(ctx, msevb_set_positions(ctx, atom_data%xyz), "msevb_set_positions")
    call msevb_check(ctx, msevb_construct_verlet_list(ctx), "msevb_construct_verlet_list")
    verlet_list_data%flag_verlet_list = 0
    list_built = .true.
  end subroutine construct_verlet_list

  !*********** src/general_routines.f90:1259
  subroutine update_verlet_displacements( total_atoms, xyz, verlet_list_data, box, xyz_to_box_transform, flag_verlet_list, flag_initialize )
    integer, intent(in) :: total_atoms
    type(verlet_list_data_type), intent(inout) :: verlet_list_data
    real*8, dimension(:,:), intent(in) :: xyz, box, xyz_to_box_transform
    integer, intent(out) :: flag_verlet_list
    integer, intent(in), optional :: flag_initialize
    type(c_ptr) :: ctx
    integer(c_int) :: flag
    flag_verlet_list = 0
    if (present(flag_initialize)) return      ! the device list build resets its displacement store itself
    ctx = need_ctx("update_verlet_displacements")
    call msevb_check(ctx, msevb_set_positions(ctx, xyz), "msevb_set_positions")
    call msevb_check(ctx, msevb_update_verlet_displacements(ctx, flag), "msevb_update_verlet_displacements")
    flag_verlet_list = flag
  end subroutine update_verlet_displacements

  !*********** src/pair_int_real_space.f90:60
  subroutine real_space_energy_force( system_data, molecule_data, atom_data, verlet_list_data, PME_data )
    Type(system_data_type),intent(inout)             :: system_data
    Type(molecule_data_type),dimension(:),intent(in) :: molecule_data
    Type(atom_data_type),intent(inout)               :: atom_data
    Type(verlet_list_data_type),intent(in)           :: verlet_list_data
    Type(PME_data_type), intent(in)                  :: PME_data
    type(c_ptr) :: ctx
    real(c_double) :: e(4)
    call msevb_hip_prepare(system_data, molecule_data, atom_data, verlet_list_data, PME_data, ms_evb_simulation == "yes")
    ctx = msevb_hip_context()
    call msevb_check(ctx, msevb_set_positions(ctx, atom_data%xyz), "msevb_set_positions")
    if (.not. list_built .or. verlet_list_data%flag_verlet_list == 1) then
       call msevb_check(ctx, msevb_construct_verlet_list(ctx), "msevb_construct_verlet_list")
       list_built = .true.
    end if
    call msevb_check(ctx, msevb_zero_forces(ctx), "msevb_zero_forces")
    call msevb_check(ctx, msevb_real_space_energy_force(ctx, e), "msevb_real_space_energy_force")
    call add_device_forces(ctx, atom_data, system_data%total_atoms)
    system_data%E_elec = system_data%E_elec + e(1) + e(3)
    system_data%E_vdw  = system_data%E_vdw  + e(2) + e(4)
  end subroutine real_space_energy_force

  !*********** src/pme.f90:28 (also leaves Q_grid, theta_conv_Q, force_recip and E_recip in PME_data, as the reference does)
  subroutine pme_reciprocal_space_energy_force( system_data, atom_data, PME_data )
    type(system_data_type), intent(inout)   :: system_data
    type(atom_data_type) , intent(inout)    :: atom_data
    type(PME_data_type)  , intent(inout), target :: PME_data
    type(c_ptr) :: ctx, pq, pt, pf
    real(c_double) :: er
    ctx = need_ctx("pme_reciprocal_space_energy_force")
    call msevb_check(ctx, msevb_set_positions(ctx, atom_data%xyz), "msevb_set_positions")
    call msevb_check(ctx, msevb_zero_forces(ctx), "msevb_zero_forces")
    call msevb_check(ctx, msevb_pme_reciprocal_space_energy_force(ctx, er), "msevb_pme_reciprocal_space_energy_force")
    call add_device_forces(ctx, atom_data, system_data%total_atoms)
    pq = c_null_ptr; pt = c_null_ptr; pf = c_null_ptr
    if (allocated(PME_data%Q_grid)) pq = c_loc(PME_data%Q_grid)
    if (allocated(PME_data%theta_conv_Q)) pt = c_loc(PME_data%theta_conv_Q)
    if (allocated(PME_data%force_recip)) pf = c_loc(PME_data%force_recip)
    call msevb_check(ctx, msevb_get_pme_state(ctx, pq, pt, c_null_ptr, pf), "msevb_get_pme_state")
    PME_data%E_recip = er
    system_data%E_elec = system_data%E_elec + er
  end subroutine pme_reciprocal_space_energy_force

  !*********** src/pme.f90:184 (xyz = SCALED coordinates, as in the reference)
  subroutine grid_Q(Q,chg,xyz, PME_data)
    real*8, intent(in), dimension(:) :: chg
    real*8, intent(in), dimension(:,:) :: xyz
    type(PME_data_type), intent(in)    :: PME_data
    real*8,dimension(:,:,:),intent(out)::Q
    type(c_ptr) :: ctx
    ctx = need_ctx("grid_Q")
    call msevb_check(ctx, msevb_grid_Q(ctx, Q, chg, xyz, size(xyz,2)), "msevb_grid_Q")
  end subroutine grid_Q

  !*********** src/pme.f90:275
  subroutine modify_Q_grid( Q , chg, xyz, n_atom, PME_data, operation)
    integer, intent(in)              :: n_atom
    real*8, intent(in), dimension(:) :: chg
    real*8, intent(in), dimension(:,:) :: xyz
    type(PME_data_type) , intent(in)   :: PME_data
    real*8,dimension(:,:,:),intent(inout)::Q
    integer, intent(in) :: operation
    type(c_ptr) :: ctx
    ctx = need_ctx("modify_Q_grid")
    if (operation == 0) stop "modify_Q_grid: operation must be positive (add) or negative (subtract)"
    call msevb_check(ctx, msevb_modify_Q_grid(ctx, Q, chg, xyz, n_atom, operation), "msevb_modify_Q_grid")
  end subroutine modify_Q_grid

  !*********** src/pme.f90:346, one atom per call as in the reference (each call moves FQ to the GPU: prefer derivative_grid_Q_all)
  subroutine derivative_grid_Q(force,FQ,chg,xyz,i_atom, PME_data, kk, conv_e2A_kJmol, flag_update)
    real*8,dimension(3),intent(out)::force
    real*8,dimension(:,:,:),intent(in)::FQ
    type(PME_data_type), intent(inout)   :: PME_data
    integer, intent(in) :: i_atom
    real*8, intent(in), dimension(:) :: chg
    real*8,intent(in),dimension(3,3)::kk
    real*8, intent(in), dimension(:,:) :: xyz
    real*8, intent(in)                 :: conv_e2A_kJmol
    integer, intent(in), optional :: flag_update
    type(c_ptr) :: ctx
    real(c_double) :: f(3,1), q1(1), u1(3,1)
    ctx = need_ctx("derivative_grid_Q")
    q1(1) = chg(i_atom); u1(:,1) = xyz(:,i_atom)
    call msevb_check(ctx, msevb_derivative_grid_Q(ctx, f, FQ, q1, u1, 1, kk, conv_e2A_kJmol), "msevb_derivative_grid_Q")
    force = f(:,1)      ! the dQ_dr store of the reference (:470-479) has no counterpart: the GPU path recomputes the stencil weights
  end subroutine derivative_grid_Q

  !*********** the loop of src/pme.f90:156-165 in one call: force(3,n) for all atoms
  subroutine derivative_grid_Q_all(force,FQ,chg,xyz, PME_data, kk, conv_e2A_kJmol)
    real*8,dimension(:,:),intent(out)::force
    real*8,dimension(:,:,:),intent(in)::FQ
    type(PME_data_type), intent(in)   :: PME_data
    real*8, intent(in), dimension(:) :: chg
    real*8,intent(in),dimension(3,3)::kk
    real*8, intent(in), dimension(:,:) :: xyz
    real*8, intent(in)                 :: conv_e2A_kJmol
    type(c_ptr) :: ctx
    ctx = need_ctx("derivative_grid_Q_all")
    call msevb_check(ctx, msevb_derivative_grid_Q(ctx, force, FQ, chg, xyz, size(xyz,2), kk, conv_e2A_kJmol), "msevb_derivative_grid_Q")
  end subroutine derivative_grid_Q_all

  !*********** src/intra_bonded_interactions.f90:17
  subroutine intra_molecular_energy_force( system_data, molecule_data, atom_data )
    type(system_data_type), intent(inout) :: system_data
    type(molecule_data_type), dimension(:), intent(in) :: molecule_data
    type(atom_data_type), intent(inout) :: atom_data
    type(c_ptr) :: ctx
    real(c_double) :: e(3)
    ctx = need_ctx("intra_molecular_energy_force")
    call msevb_check(ctx, msevb_set_positions(ctx, atom_data%xyz), "msevb_set_positions")
    call msevb_check(ctx, msevb_zero_forces(ctx), "msevb_zero_forces")
    call msevb_check(ctx, msevb_intra_molecular_energy_force(ctx, e), "msevb_intra_molecular_energy_force")
    call add_device_forces(ctx, atom_data, system_data%total_atoms)
    system_data%E_bond = system_data%E_bond + e(1); system_data%E_angle = system_data%E_angle + e(2); system_data%E_dihedral = system_data%E_dihedral + e(3)
  end subroutine intra_molecular_energy_force

  !*********** src/md_integration.f90:438: ONE step, integrator kernels + force call on the device (positions, velocities and forces
  ! go up before and come back after the step, so the host may touch its arrays between steps).  Non-EVB builds; an EVB host keeps the
  ! reference's md_integrate_atomic, whose force call is the GPU one (module ms_evb in ms_evb_hip.f90 re-topologises the host arrays
  ! after a principal-diabat switch with the reference's own routine).
  subroutine md_integrate_atomic( system_data , molecule_data , atom_data, integrator_data, verlet_list_data, PME_data, file_io_data )
    type( system_data_type ), intent(inout)     :: system_data
    type( molecule_data_type ), dimension(:), intent(inout)   :: molecule_data
    type( atom_data_type ) , intent(inout)      :: atom_data
    type( integrator_data_type ), intent(in)    :: integrator_data
    type(verlet_list_data_type), intent(inout)  :: verlet_list_data
    type(PME_data_type)     , intent(inout)     :: PME_data
    type(file_io_data_type) , intent(in)        :: file_io_data
    type(c_ptr) :: ctx
    real(c_double) :: en(2,1)
    integer :: i_mole, k
    if (ms_evb_simulation == "yes") stop "msevb_hip_stages::md_integrate_atomic: use the reference's md_integrate_atomic with module ms_evb of ms_evb_hip.f90"
    call msevb_hip_prepare(system_data, molecule_data, atom_data, verlet_list_data, PME_data, .false.)
    ctx = msevb_hip_context()
    if (integrator_data%ensemble /= "NVE") then      ! langevin_integrator, src/md_integration.f90:186
       call msevb_check(ctx, msevb_set_option(ctx, "ensemble_nvt"//c_null_char, 1d0), "msevb_set_option")
       call msevb_check(ctx, msevb_set_option(ctx, "friction_coeff"//c_null_char, integrator_data%friction_coeff), "msevb_set_option")
       call msevb_check(ctx, msevb_set_option(ctx, "temperature"//c_null_char, system_data%temperature), "msevb_set_option")
    else
       call msevb_check(ctx, msevb_set_option(ctx, "ensemble_nvt"//c_null_char, 0d0), "msevb_set_option")
    end if
    call msevb_check(ctx, msevb_set_positions(ctx, atom_data%xyz), "msevb_set_positions")
    call msevb_check(ctx, msevb_set_velocities(ctx, atom_data%velocity), "msevb_set_velocities")
    call msevb_check(ctx, msevb_set_forces(ctx, atom_data%force), "msevb_set_forces")
    call msevb_check(ctx, msevb_md_integrate_atomic(ctx, 1, integrator_data%delta_t, 0, en), "msevb_md_integrate_atomic")
    call msevb_check(ctx, msevb_get_positions(ctx, atom_data%xyz), "msevb_get_positions")
    call msevb_check(ctx, msevb_get_velocities(ctx, atom_data%velocity), "msevb_get_velocities")
    call msevb_check(ctx, msevb_get_forces(ctx, atom_data%force), "msevb_get_forces")
    system_data%potential_energy = en(1,1)
    ! centres of mass as update_r_com leaves them (src/general_routines.f90:420; the device moved whole molecules back into the box)
    do i_mole = 1, system_data%n_mole
       molecule_data(i_mole)%r_com = 0d0
       do k = 1, molecule_data(i_mole)%n_atom
          molecule_data(i_mole)%r_com = molecule_data(i_mole)%r_com + atom_data%mass(molecule_data(i_mole)%atom_index(k)) * atom_data%xyz(:,molecule_data(i_mole)%atom_index(k))
       end do
       molecule_data(i_mole)%r_com = molecule_data(i_mole)%r_com / sum(atom_data%mass(molecule_data(i_mole)%atom_index(1:molecule_data(i_mole)%n_atom)))
    end do
    list_built = .true.
  end subroutine md_integrate_atomic

end module msevb_hip_stages
