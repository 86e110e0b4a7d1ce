!==============================================================================
! Drop-in replacement of the reference's module `total_energy_forces`
! (src/total_energy_forces.f90): the force call runs on an MI355X through
! libmsevb_hip.so.
!
! It is compiled AGAINST THE REFERENCE'S OWN, UNCHANGED glob_v.f90 /
! general_routines.f90 / pair_int_real_space.f90 / pme.f90 /
! intra_bonded_interactions.f90: the derived types, force-field tables and EVB
! tables are read from there, and - like the reference's module
! (src/total_energy_forces.f90:5-10) - it re-exports pair_int_real_space,
! pme_routines and bonded_interactions, through which md_integration.f90 reaches
! periodic_box_change (src/md_integration.f90:231-233,294,334).  main_ms_evb.f90,
! md_integration.f90 and initialize_routines.f90 stay untouched; see
! INTEGRATION.md and `make -C reactive_pb_nn_md_amd/fortran link_check`.
! Same procedure names, same argument lists:
!   calculate_total_force_energy(system_data, molecule_data, atom_data, verlet_list_data, PME_data)
!   calculate_kinetic_energy(KE, n_atom, mass, vel_atom, conv_fac)
! plus the body of ms_evb_calculate_total_force_energy (module ms_evb in ms_evb_hip.f90):
!   msevb_hip_ms_evb_force_energy(...)
!
! Environment (one process per GPU):
!   MSEVB_DEVICE     HIP device ordinal of this process (default: LOCAL_RANK, else 0)
!   MSEVB_RANK / MSEVB_NRANKS / MSEVB_COMM_FILE   join an RCCL communicator of MSEVB_NRANKS processes: rank 0 writes the
!                    128-byte unique id to MSEVB_COMM_FILE, the others wait for that file (the reference has no MPI to broadcast it)
!==============================================================================
module total_energy_forces
  use global_variables
  use routines
  use pair_int_real_space
  use pme_routines
  use bonded_interactions
  use msevb_hip_c
  use iso_c_binding
  implicit none
  type(c_ptr), save, private :: ctx = c_null_ptr
  logical, save, private :: uploaded = .false., uploaded_evb = .false.
  real*8, save, private :: box_uploaded(3,3) = 0d0

contains

  !-- the library context of this process (for callers that want the C entry points directly, e.g. msevb_hip_stages)
  function msevb_hip_context() result(p)
    type(c_ptr) :: p
    p = ctx
  end function msevb_hip_context

  !-- host arrays were re-ordered (principal-diabat switch): the next force call uploads everything again
  subroutine msevb_hip_invalidate()
    if (c_associated(ctx)) call msevb_check(ctx, msevb_destroy(ctx), "msevb_destroy")
    ctx = c_null_ptr
    uploaded = .false.
  end subroutine msevb_hip_invalidate

  !-- host arrays were re-ordered by the reference's evb_change_diabat_data_structure_topology (principal-diabat switch, src/ms_evb.f90:218-227): the
  !   library, which never moves atoms, is told where each of the host's atoms sits in ITS order; tables, FFT plans, the captured graph, the
  !   Verlet list and its capacity, the RCCL communicator all stay.  Molecule m is the same molecule on both sides and its atoms are in the same
  !   local order (the library re-topologises exactly as the reference does, msevb_get_topology).
  subroutine msevb_hip_remap(system_data, molecule_data)
    type(system_data_type), intent(in) :: system_data
    type(molecule_data_type), dimension(:), intent(in) :: molecule_data
    integer(c_int), allocatable :: at(:), mn(:), mt(:), ma(:,:), perm(:)
    real(c_double), allocatable :: chg(:)
    integer(c_int) :: hyd
    integer :: m, k, n
    integer(8) :: t0, t1, rate
    if (.not. c_associated(ctx)) return
    call system_clock(t0, rate)
    n = system_data%total_atoms
    allocate(at(n), chg(n), mn(system_data%n_mole), mt(system_data%n_mole), ma(8, system_data%n_mole), perm(n))
    call msevb_check(ctx, msevb_get_topology(ctx, at, chg, mn, mt, ma, hyd), "msevb_get_topology")
    perm = 0
    do m = 1, system_data%n_mole
       if (mn(m) /= molecule_data(m)%n_atom) stop "msevb_hip_remap: the library and the host disagree about a molecule's size after the proton hop"
       do k = 1, mn(m)
          perm(molecule_data(m)%atom_index(k)) = ma(k, m)
       end do
    end do
    call msevb_check(ctx, msevb_set_host_order(ctx, perm), "msevb_set_host_order")
    call system_clock(t1)
    write(*,'(A,F9.3,A,I0,A)') " msevb_hip_remap: host atom order handed to the GPU library in ", 1d3 * dble(t1 - t0) / dble(rate), " ms (", n, " atoms; context, tables, plans and list kept)"
  end subroutine msevb_hip_remap

  integer function env_int(name, default)
    character(*), intent(in) :: name
    integer, intent(in) :: default
    character(64) :: v
    integer :: l, st, ios
    env_int = default
    call get_environment_variable(name, v, l, st)
    if (st == 0 .and. l > 0) then
       read(v(1:l), *, iostat=ios) env_int
       if (ios /= 0) env_int = default
    end if
  end function env_int

  !-- MSEVB_RANK / MSEVB_NRANKS / MSEVB_COMM_FILE: one process per GPU, RCCL communicator over xGMI
  subroutine msevb_hip_join_communicator()
    integer :: rank, nranks, l, st, ios, tries
    character(512) :: fname
    integer(c_signed_char) :: id(128)
    logical :: there
    nranks = env_int("MSEVB_NRANKS", 1)
    if (nranks < 2) return
    rank = env_int("MSEVB_RANK", env_int("RANK", 0))
    call get_environment_variable("MSEVB_COMM_FILE", fname, l, st)
    if (st /= 0 .or. l == 0) stop "MSEVB_NRANKS > 1 needs MSEVB_COMM_FILE (a path all ranks can read)"
    if (rank == 0) then
       call msevb_check(ctx, msevb_comm_unique_id(id), "msevb_comm_unique_id")
       open(unit=97, file=fname(1:l)//".tmp", access='stream', form='unformatted', status='replace')
       write(97) id
       close(97)
       call rename(fname(1:l)//".tmp", fname(1:l))      ! the id appears under its name only when complete
    else
       do tries = 1, 6000
          inquire(file=fname(1:l), exist=there)
          if (there) exit
          call sleep(1)
       end do
       if (.not. there) stop "MSEVB_COMM_FILE did not appear"
       open(unit=97, file=fname(1:l), access='stream', form='unformatted', status='old')
       read(97, iostat=ios) id
       close(97)
       if (ios /= 0) stop "MSEVB_COMM_FILE unreadable"
    end if
    call msevb_check(ctx, msevb_comm_init(ctx, id, rank, nranks), "msevb_comm_init")
    ! the join is a collective: every rank has read the id by now.  The file goes, so that a later run (or a later join of this one) that names the
    ! same path cannot pick up this run's id
    if (rank == 0) then
       open(unit=97, file=fname(1:l), status='old', iostat=ios)
       if (ios == 0) close(97, status='delete')
    end if
  end subroutine msevb_hip_join_communicator

  !-- push every table the hot path reads to the GPU library (once; again after a box change)
  subroutine msevb_hip_upload(system_data, molecule_data, atom_data, verlet_list_data, PME_data, use_evb)
    type(system_data_type), intent(in) :: system_data
    type(molecule_data_type), dimension(:), intent(in) :: molecule_data
    type(atom_data_type), intent(in) :: atom_data
    type(verlet_list_data_type), intent(in) :: verlet_list_data
    type(PME_data_type), intent(in) :: PME_data
    logical, intent(in) :: use_evb
    integer(c_int), allocatable :: mfirst(:), mnat(:), mtype(:), idx4(:,:), idx5(:,:), bl(:,:), al(:,:), dl(:,:), rp(:), rb(:)
    real(c_double), allocatable :: par2(:,:), par6(:,:)
    integer :: i, j, k, l, n, it, na, nb, nang, nd, dev

    if (.not. c_associated(ctx)) then
       dev = env_int("MSEVB_DEVICE", env_int("LOCAL_RANK", 0))
       call msevb_check(ctx, msevb_create(ctx, dev), "msevb_create")
    end if
    call msevb_check(ctx, msevb_set_system(ctx, system_data%total_atoms, system_data%n_mole, system_data%box, real_space_cutoff, &
         verlet_list_data%verlet_cutoff, PME_data%alpha_sqrt, PME_data%pme_grid, PME_data%spline_order, PME_data%spline_grid, PME_data%erfc_grid), "msevb_set_system")
    n = system_data%n_mole
    allocate(mfirst(n), mnat(n), mtype(n))
    do i = 1, n
       mfirst(i) = molecule_data(i)%atom_index(1); mnat(i) = molecule_data(i)%n_atom; mtype(i) = molecule_data(i)%molecule_type_index
    end do
    call msevb_check(ctx, msevb_set_atoms(ctx, atom_data%atom_type_index, atom_data%charge, atom_data%mass, mfirst, mnat, mtype), "msevb_set_atoms")
    call msevb_check(ctx, msevb_set_forcefield(ctx, n_atom_type, MAX_N_ATOM_TYPE, atype_vdw_type, atype_vdw_parameter, atype_vdw_parameter_14, &
         atype_bond_type, atype_bond_parameter, atype_chg), "msevb_set_forcefield")
    ! sparse form of the 25^3 / 25^4 tensors
    n = count(atype_angle_type(1:n_atom_type,1:n_atom_type,1:n_atom_type) /= 0)
    allocate(idx4(4,max(n,1)), par2(2,max(n,1)))
    n = 0
    do i=1,n_atom_type; do j=1,n_atom_type; do k=1,n_atom_type
       if (atype_angle_type(i,j,k) /= 0) then
          n = n + 1; idx4(:,n) = (/i,j,k,atype_angle_type(i,j,k)/); par2(:,n) = atype_angle_parameter(i,j,k,:)
       end if
    end do; end do; end do
    call msevb_check(ctx, msevb_set_angle_types(ctx, n, idx4, par2), "msevb_set_angle_types")
    n = count(atype_dihedral_type(1:n_atom_type,1:n_atom_type,1:n_atom_type,1:n_atom_type) /= 0)
    allocate(idx5(5,max(n,1)), par6(6,max(n,1)))
    n = 0
    do i=1,n_atom_type; do j=1,n_atom_type; do k=1,n_atom_type; do l=1,n_atom_type
       if (atype_dihedral_type(i,j,k,l) /= 0) then
          n = n + 1; idx5(:,n) = (/i,j,k,l,atype_dihedral_type(i,j,k,l)/); par6(:,n) = atype_dihedral_parameter(i,j,k,l,:)
       end if
    end do; end do; end do; end do
    call msevb_check(ctx, msevb_set_dihedral_types(ctx, n, idx5, par6), "msevb_set_dihedral_types")
    do it = 1, n_molecule_type
       na = molecule_type_data(it)%n_atom
       nb = size(molecule_type_data(it)%bond_list); nang = size(molecule_type_data(it)%angle_list); nd = size(molecule_type_data(it)%dihedral_list)
       allocate(bl(2,max(nb,1)), al(3,max(nang,1)), dl(4,max(nd,1)), rp(na), rb(na))
       do i=1,nb; bl(:,i) = (/molecule_type_data(it)%bond_list(i)%i_atom, molecule_type_data(it)%bond_list(i)%j_atom/); end do
       do i=1,nang; al(:,i) = (/molecule_type_data(it)%angle_list(i)%i_atom, molecule_type_data(it)%angle_list(i)%j_atom, molecule_type_data(it)%angle_list(i)%k_atom/); end do
       do i=1,nd; dl(:,i) = (/molecule_type_data(it)%dihedral_list(i)%i_atom, molecule_type_data(it)%dihedral_list(i)%j_atom, &
            molecule_type_data(it)%dihedral_list(i)%k_atom, molecule_type_data(it)%dihedral_list(i)%l_atom/); end do
       rp = 0; rb = 0
       if (allocated(molecule_type_data(it)%evb_reactive_protons)) rp = molecule_type_data(it)%evb_reactive_protons
       if (allocated(molecule_type_data(it)%evb_reactive_basic_atoms)) rb = molecule_type_data(it)%evb_reactive_basic_atoms
       call msevb_check(ctx, msevb_set_molecule_type(ctx, it, na, molecule_type_data(it)%atom_type_index, nb, bl, nang, al, nd, dl, &
            molecule_type_data(it)%pair_exclusions, rp, rb), "msevb_set_molecule_type")
       deallocate(bl, al, dl, rp, rb)
    end do
    if (use_evb) then
       call msevb_check(ctx, msevb_set_evb(ctx, n_molecule_type, MAX_N_MOLE_TYPE, max_interaction_type, evb_acid_molecule, evb_basic_molecule, &
            evb_conjugate_pairs, evb_proton_index, evb_heavy_acid_index, evb_conjugate_atom_index, evb_reference_energy, evb_exchange_charge_atomic, &
            evb_exchange_charge_proton, evb_donor_acceptor_interaction, evb_donor_acceptor_parameters, evb_proton_acceptor_interaction, &
            evb_proton_acceptor_parameters, evb_diabat_coupling_interaction, evb_diabat_coupling_parameters, evb_diabat_coupling_type, &
            hydronium_molecule_index(1), evb_max_chain), "msevb_set_evb")
    end if
    call msevb_check(ctx, msevb_finalize(ctx), "msevb_finalize")
    call msevb_check(ctx, msevb_set_frozen_types(ctx, atype_freeze, n_atom_type), "msevb_set_frozen_types")
    call msevb_hip_join_communicator()
    uploaded = .true.
    uploaded_evb = use_evb
    box_uploaded = system_data%box
  end subroutine msevb_hip_upload

  !-- everything a force call needs before it starts: tables uploaded (again after msevb_hip_invalidate), and the box the host holds NOW.
  !   The Monte-Carlo barostat changes system_data%box and calls periodic_box_change (src/pme.f90:634) of the reference's own pme.f90 for
  !   the host-side CB array; the GPU side follows here: a changed box goes to msevb_set_box (CB, reciprocal lattice, Verlet list)
  subroutine msevb_hip_prepare(system_data, molecule_data, atom_data, verlet_list_data, PME_data, use_evb)
    type(system_data_type), intent(in) :: system_data
    type(molecule_data_type), dimension(:), intent(in) :: molecule_data
    type(atom_data_type), intent(in) :: atom_data
    type(verlet_list_data_type), intent(in) :: verlet_list_data
    type(PME_data_type), intent(in) :: PME_data
    logical, intent(in) :: use_evb
    real(c_double) :: b3(3)
    if (uploaded .and. (use_evb .neqv. uploaded_evb)) call msevb_hip_invalidate()
    if (.not. uploaded) call msevb_hip_upload(system_data, molecule_data, atom_data, verlet_list_data, PME_data, use_evb)
    if (any(system_data%box /= box_uploaded)) then
       b3 = (/ system_data%box(1,1), system_data%box(2,2), system_data%box(3,3) /)
       call msevb_check(ctx, msevb_set_box(ctx, b3), "msevb_set_box")
       box_uploaded = system_data%box
    end if
  end subroutine msevb_hip_prepare

  !***********************************************************
  ! same interface as the reference (src/total_energy_forces.f90:19)
  !***********************************************************
  subroutine calculate_total_force_energy(system_data, molecule_data, atom_data, verlet_list_data, PME_data )
    Type(system_data_type),intent(inout)                :: system_data
    Type(molecule_data_type),dimension(:),intent(in)    :: molecule_data
    Type(atom_data_type),intent(inout)                  :: atom_data
    Type(verlet_list_data_type),intent(inout)           :: verlet_list_data
    Type(PME_data_type), intent(inout)                  :: PME_data
    real(c_double) :: e(MSEVB_N_ENERGY)

    call msevb_hip_prepare(system_data, molecule_data, atom_data, verlet_list_data, PME_data, .false.)
    call msevb_check(ctx, msevb_set_positions(ctx, atom_data%xyz), "msevb_set_positions")
    ! the Verlet list (and its freshness logic, :30-39) lives on the device
    call msevb_check(ctx, msevb_calculate_total_force_energy(ctx, e, atom_data%force), "msevb_calculate_total_force_energy")
    system_data%potential_energy = e(1); system_data%E_elec = e(2); system_data%E_vdw = e(3)
    system_data%E_bond = e(4); system_data%E_angle = e(5); system_data%E_dihedral = e(6)
    PME_data%E_recip = e(7)
    verlet_list_data%flag_verlet_list = 0
  end subroutine calculate_total_force_energy

  !***********************************************************
  ! body of ms_evb_calculate_total_force_energy (src/ms_evb.f90:181-235): the whole EVB force call on the GPU.  Returns what the
  ! reference's diagonalize_evb_hamiltonian leaves behind for the host (src/ms_evb.f90:242-344): the number of diabats, the proton log
  ! and coupling matrix, the Hamiltonian, the ground-state vector, the principal diabat and the hydronium molecule it implies.
  ! proton_log is (n_diabat, max_chain, 5) as the reference's evb_diabat_proton_log(evb_max_states, evb_max_chain, 5).
  !***********************************************************
  subroutine msevb_hip_ms_evb_force_energy(system_data, molecule_data, atom_data, verlet_list_data, PME_data, n_diabat, proton_log, coupling_matrix, &
       hamiltonian, eigenvector, principle_diabat, new_hydronium)
    Type(system_data_type),intent(inout)                :: system_data
    Type(molecule_data_type),dimension(:),intent(in)    :: molecule_data
    Type(atom_data_type),intent(inout)                  :: atom_data
    Type(verlet_list_data_type),intent(inout)           :: verlet_list_data
    Type(PME_data_type), intent(inout)                  :: PME_data
    integer, intent(out) :: n_diabat, principle_diabat, new_hydronium
    integer, intent(inout) :: proton_log(:,:,:), coupling_matrix(:)
    real*8, intent(inout) :: hamiltonian(:,:), eigenvector(:)
    integer(c_int) :: nd, pr, nh
    integer(c_int), allocatable, target :: plog(:,:,:), par(:)
    real(c_double), allocatable, target :: hh(:,:), ev(:)
    integer :: d, h, q, mc

    call msevb_hip_prepare(system_data, molecule_data, atom_data, verlet_list_data, PME_data, .true.)
    call msevb_check(ctx, msevb_set_positions(ctx, atom_data%xyz), "msevb_set_positions")
    call msevb_check(ctx, msevb_ms_evb_calculate_total_force_energy(ctx, system_data%potential_energy, atom_data%force), "msevb_ms_evb_calculate_total_force_energy")
    call msevb_check(ctx, msevb_get_evb_state(ctx, nd, c_null_ptr, c_null_ptr, c_null_ptr, c_null_ptr, pr, nh), "msevb_get_evb_state")
    mc = evb_max_chain
    allocate(plog(5, mc, nd), par(nd), hh(nd, nd), ev(nd))      ! the library writes C order [d][hop][5], [i][j]
    call msevb_check(ctx, msevb_get_evb_state(ctx, nd, c_loc(plog), c_loc(par), c_loc(hh), c_loc(ev), pr, nh), "msevb_get_evb_state")
    n_diabat = nd; principle_diabat = pr; new_hydronium = nh
    if (nd > size(proton_log,1) .or. nd > size(hamiltonian,1)) stop "more diabats than evb_max_states"
    proton_log = -1
    do d = 1, nd; do h = 1, min(mc, size(proton_log,2)); do q = 1, 5
       proton_log(d, h, q) = plog(q, h, d)
    end do; end do; end do
    coupling_matrix(1:nd) = par(1:nd)
    hamiltonian = 0d0
    do d = 1, nd; do h = 1, nd
       hamiltonian(d, h) = hh(h, d)      ! C [i][j] -> Fortran (i,j)
    end do; end do
    eigenvector(1:nd) = ev(1:nd)
    verlet_list_data%flag_verlet_list = 0
  end subroutine msevb_hip_ms_evb_force_energy

  !**************************************
  ! same as the reference (src/total_energy_forces.f90:106); cheap, stays on the host
  !**************************************
  subroutine calculate_kinetic_energy( KE, n_atom, mass, vel_atom, conv_fac )
    real*8,intent(out) :: KE
    integer, intent(in) :: n_atom
    real*8,dimension(:),intent(in) :: mass
    real*8,dimension(:,:), intent(in) :: vel_atom
    real*8,intent(in) :: conv_fac
    integer :: i_atom
    KE=0d0
    do i_atom=1,n_atom
       KE = KE + 0.5d0 * mass(i_atom) * dot_product(vel_atom(:,i_atom),vel_atom(:,i_atom))/ conv_fac
    end do
  end subroutine calculate_kinetic_energy

end module total_energy_forces
