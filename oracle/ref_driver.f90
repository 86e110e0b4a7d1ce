!==============================================================================
! TEST INFRASTRUCTURE ONLY (oracle/): golden-vector dump driver.
!
! This program is build-authored code that links against the UNMODIFIED modules
! of the reference (compiled from /root/reference/src by oracle/build_ref.sh)
! and calls their public procedures to produce golden vectors:
!   - replays the reference's start-up sequence (src/main_ms_evb.f90:37-84),
!   - evaluates each stage of the hot path separately
!     (pairwise_real_space_verlet   src/pair_int_real_space.f90:135,
!      intra_molecular_pairwise_... src/pair_int_real_space.f90:386,
!      pme_reciprocal_space_...     src/pme.f90:28,
!      intra_molecular_energy_force src/intra_bonded_interactions.f90:17,
!      calculate_total_force_energy src/total_energy_forces.f90:19),
!   - in the EVB build, construct_evb_hamiltonian / diagonalize_evb_hamiltonian
!     (src/ms_evb.f90:375 / :242) with the per-element force store,
!   - optionally n velocity-Verlet steps of md_integrate_atomic
!     (src/md_integration.f90:438) from a seeded velocity file.
! The reference never writes forces (src/md_integration.f90:503-505 is commented
! out), hence this driver.  Output: a flat tagged binary stream read by
! oracle/refdump.py.
!
! usage: ref_dump gro ffpmt top simpmt traj_out log_out hop_out a_out DUMP [VELFILE NSTEPS [NBARO]]
!        NBARO > 0: after the NSTEPS steps, NBARO x (monte_carlo_barostat move, md_integrate_atomic step) as mc_sample orders them
!        (src/md_integration.f90:28-37), with the random numbers each move draws recorded next to its outcome
!==============================================================================
module dumpio
  implicit none
  integer, parameter :: du = 77
contains
  subroutine dump_open(fname)
    character(*), intent(in) :: fname
    open(unit=du, file=fname, access='stream', form='unformatted', status='replace')
  end subroutine
  subroutine dump_close()
    close(du)
  end subroutine
  subroutine dump_hdr(name, tcode, dims)
    character(*), intent(in) :: name
    integer, intent(in) :: tcode
    integer, intent(in) :: dims(:)
    character(32) :: nm
    integer :: d4(4), nd
    nm = name
    d4 = 1
    nd = size(dims)
    d4(1:nd) = dims
    write(du) nm, tcode, nd, d4
  end subroutine
  subroutine dump_r8(name, a, dims)
    character(*), intent(in) :: name
    real*8, intent(in) :: a(*)
    integer, intent(in) :: dims(:)
    integer :: n
    n = product(dims)
    call dump_hdr(name, 1, dims)
    write(du) a(1:n)
  end subroutine
  subroutine dump_i4(name, a, dims)
    character(*), intent(in) :: name
    integer, intent(in) :: a(*)
    integer, intent(in) :: dims(:)
    integer :: n
    n = product(dims)
    call dump_hdr(name, 0, dims)
    write(du) a(1:n)
  end subroutine
  subroutine dump_s(name, v)
    character(*), intent(in) :: name
    real*8, intent(in) :: v
    real*8 :: t(1)
    t(1) = v
    call dump_r8(name, t, (/1/))
  end subroutine
  subroutine dump_is(name, v)
    character(*), intent(in) :: name
    integer, intent(in) :: v
    integer :: t(1)
    t(1) = v
    call dump_i4(name, t, (/1/))
  end subroutine
end module dumpio

program ref_dump
  use routines
  use global_variables
  use simulation_parameters
  use total_energy_forces
  use md_integration
  use initialize_routines
  use pme_routines
  use pair_int_real_space
  use bonded_interactions
  use ms_evb
  use dumpio
  implicit none

  Type(system_data_type)                               :: system_data
  Type(molecule_data_type), dimension(:), allocatable  :: molecule_data
  Type(atom_data_type)                                 :: atom_data
  Type(integrator_data_type)                           :: integrator_data
  Type(file_io_data_type)                              :: file_io_data
  Type(verlet_list_data_type)                          :: verlet_list_data
  Type(PME_data_type)                                  :: PME_data
  type(single_molecule_data_type) :: smd

  character(300) :: dumpfile, velfile, arg
  integer :: nsteps, N, M, K, i, j, k2, l, nargs_cli, i_mole, n_nz, it, nt, na, nb, nang, ndih, D, ns, i_hyd_trace
  integer :: principle_diabat, new_h, i_step
  integer, allocatable :: itmp(:), itmp2(:,:)
  real*8, allocatable :: rtmp(:,:), vel(:,:), etraj(:,:)
  integer, allocatable :: vflag(:), ndia(:)
  integer :: nbaro, i_baro, nseed
  integer, allocatable :: rstate(:)
  real*8 :: r12(2)
  real*8, allocatable :: baro_rand(:,:), baro_box(:), baro_epot(:), baro_scale(:), baro_xyz(:,:,:), baro_epot_md(:)
  real*8 :: Epot

  nargs_cli = command_argument_count()
  if (nargs_cli < 9) stop "usage: ref_dump <8 reference args> DUMP [VELFILE NSTEPS]"
  call get_command_argument(9, dumpfile)
  nsteps = 0
  nbaro = 0
  if (nargs_cli >= 12) then
     call get_command_argument(12, arg)
     read(arg,*) nbaro
  end if
  if (nargs_cli >= 11) then
     call get_command_argument(10, velfile)
     call get_command_argument(11, arg)
     read(arg,*) nsteps
  end if

  !----- start-up, as src/main_ms_evb.f90:37-84 (restart logic bypassed)
  call initialize_constants( file_io_data , verlet_list_data , PME_data )
  call sort_input_files( file_io_data )
  restart_trajectory = "no"
  n_old_trajectory = 0
  trajectory_step = 0
  call read_simulation_parameters( file_io_data%ifile_simpmt_file_h, file_io_data%ifile_simpmt, system_data, integrator_data, verlet_list_data, PME_data )
  call initialize_simulation( system_data, molecule_data, atom_data, file_io_data, verlet_list_data, PME_data )
  open( file_io_data%ofile_traj_file_h, file=file_io_data%ofile_traj, status='replace' )
  open( file_io_data%ofile_log_file_h,  file=file_io_data%ofile_log, status='replace' )
  call initialize_energy_force( system_data, molecule_data, atom_data, verlet_list_data, PME_data, file_io_data, integrator_data )

  N = system_data%total_atoms
  M = system_data%n_mole
  K = PME_data%pme_grid
  call dump_open(trim(dumpfile))

  !----- configuration / static data
  call dump_is('evb_build', merge(1,0,ms_evb_simulation == "yes"))
  call dump_is('n_atom', N)
  call dump_is('n_mole', M)
  call dump_r8('box', system_data%box, (/3,3/))
  call dump_r8('xyz', atom_data%xyz, (/3,N/))
  call dump_r8('charge', atom_data%charge, (/N/))
  call dump_r8('mass', atom_data%mass, (/N/))
  call dump_i4('atom_type_index', atom_data%atom_type_index, (/N/))
  allocate(itmp(M))
  do i=1,M; itmp(i) = molecule_data(i)%n_atom; end do
  call dump_i4('mol_n_atom', itmp, (/M/))
  do i=1,M; itmp(i) = molecule_data(i)%atom_index(1); end do
  call dump_i4('mol_first_atom', itmp, (/M/))
  do i=1,M; itmp(i) = molecule_data(i)%molecule_type_index; end do
  call dump_i4('mol_type_index', itmp, (/M/))
  deallocate(itmp)
  allocate(rtmp(3,M))
  do i=1,M; rtmp(:,i) = molecule_data(i)%r_com; end do
  call dump_r8('mol_r_com', rtmp, (/3,M/))
  deallocate(rtmp)

  call dump_s('real_space_cutoff', real_space_cutoff)
  call dump_s('verlet_cutoff', verlet_list_data%verlet_cutoff)
  call dump_s('alpha_sqrt', PME_data%alpha_sqrt)
  call dump_is('pme_grid', K)
  call dump_is('spline_order', PME_data%spline_order)
  call dump_is('spline_grid', PME_data%spline_grid)
  call dump_is('erfc_grid', PME_data%erfc_grid)
  call dump_s('erfc_dx', PME_data%erfc_dx)
  call dump_s('delta_t', integrator_data%delta_t)
  call dump_s('conv_e2A_kJmol', constants%conv_e2A_kJmol)
  call dump_s('pi', constants%pi)
  call dump_s('pi_sqrt', constants%pi_sqrt)
  call dump_is('n_exclusions', n_exclusions)
  call dump_is('nslist', verlet_list_data%na_nslist)

  !----- force-field tables (src/glob_v.f90:320-337)
  call dump_is('n_atom_type', n_atom_type)
  call dump_is('n_molecule_type', n_molecule_type)
  call dump_r8('atype_chg', atype_chg, (/MAX_N_ATOM_TYPE/))
  call dump_r8('atype_mass', atype_mass, (/MAX_N_ATOM_TYPE/))
  call dump_i4('atype_freeze', atype_freeze, (/MAX_N_ATOM_TYPE/))
  call dump_r8('atype_vdw_parameter', atype_vdw_parameter, (/MAX_N_ATOM_TYPE,MAX_N_ATOM_TYPE,6/))
  call dump_r8('atype_vdw_parameter_14', atype_vdw_parameter_14, (/MAX_N_ATOM_TYPE,MAX_N_ATOM_TYPE,6/))
  call dump_i4('atype_vdw_type', atype_vdw_type, (/MAX_N_ATOM_TYPE,MAX_N_ATOM_TYPE/))
  call dump_i4('atype_bond_type', atype_bond_type, (/MAX_N_ATOM_TYPE,MAX_N_ATOM_TYPE/))
  call dump_r8('atype_bond_parameter', atype_bond_parameter, (/MAX_N_ATOM_TYPE,MAX_N_ATOM_TYPE,3/))
  ! sparse dumps of the 25^3 / 25^4 tensors
  nt = n_atom_type
  n_nz = 0
  do i=1,nt; do j=1,nt; do k2=1,nt
     if (atype_angle_type(i,j,k2) /= 0) n_nz = n_nz + 1
  end do; end do; end do
  allocate(itmp2(4,max(n_nz,1)), rtmp(2,max(n_nz,1)))
  n_nz = 0
  do i=1,nt; do j=1,nt; do k2=1,nt
     if (atype_angle_type(i,j,k2) /= 0) then
        n_nz = n_nz + 1
        itmp2(:,n_nz) = (/i,j,k2,atype_angle_type(i,j,k2)/)
        rtmp(:,n_nz) = atype_angle_parameter(i,j,k2,:)
     end if
  end do; end do; end do
  call dump_i4('angle_types_idx', itmp2, (/4,n_nz/))
  call dump_r8('angle_types_par', rtmp, (/2,n_nz/))
  deallocate(itmp2, rtmp)
  n_nz = 0
  do i=1,nt; do j=1,nt; do k2=1,nt; do l=1,nt
     if (atype_dihedral_type(i,j,k2,l) /= 0) n_nz = n_nz + 1
  end do; end do; end do; end do
  allocate(itmp2(5,max(n_nz,1)), rtmp(6,max(n_nz,1)))
  n_nz = 0
  do i=1,nt; do j=1,nt; do k2=1,nt; do l=1,nt
     if (atype_dihedral_type(i,j,k2,l) /= 0) then
        n_nz = n_nz + 1
        itmp2(:,n_nz) = (/i,j,k2,l,atype_dihedral_type(i,j,k2,l)/)
        rtmp(:,n_nz) = atype_dihedral_parameter(i,j,k2,l,:)
     end if
  end do; end do; end do; end do
  call dump_i4('dihedral_types_idx', itmp2, (/5,n_nz/))
  call dump_r8('dihedral_types_par', rtmp, (/6,n_nz/))
  deallocate(itmp2, rtmp)

  !----- molecule types (src/glob_v.f90:299-310)
  do it=1,n_molecule_type
     write(arg,'(A,I0,A)') 'mt', it, '_'
     na = molecule_type_data(it)%n_atom
     call dump_is(trim(arg)//'n_atom', na)
     call dump_i4(trim(arg)//'atom_type_index', molecule_type_data(it)%atom_type_index, (/na/))
     nb = size(molecule_type_data(it)%bond_list)
     allocate(itmp2(2,max(nb,1)))
     do i=1,nb
        itmp2(:,i) = (/molecule_type_data(it)%bond_list(i)%i_atom, molecule_type_data(it)%bond_list(i)%j_atom/)
     end do
     call dump_i4(trim(arg)//'bonds', itmp2, (/2,nb/))
     deallocate(itmp2)
     nang = size(molecule_type_data(it)%angle_list)
     allocate(itmp2(3,max(nang,1)))
     do i=1,nang
        itmp2(:,i) = (/molecule_type_data(it)%angle_list(i)%i_atom, molecule_type_data(it)%angle_list(i)%j_atom, molecule_type_data(it)%angle_list(i)%k_atom/)
     end do
     call dump_i4(trim(arg)//'angles', itmp2, (/3,nang/))
     deallocate(itmp2)
     ndih = size(molecule_type_data(it)%dihedral_list)
     allocate(itmp2(4,max(ndih,1)))
     do i=1,ndih
        itmp2(:,i) = (/molecule_type_data(it)%dihedral_list(i)%i_atom, molecule_type_data(it)%dihedral_list(i)%j_atom, &
             molecule_type_data(it)%dihedral_list(i)%k_atom, molecule_type_data(it)%dihedral_list(i)%l_atom/)
     end do
     call dump_i4(trim(arg)//'dihedrals', itmp2, (/4,ndih/))
     deallocate(itmp2)
     call dump_i4(trim(arg)//'pair_exclusions', molecule_type_data(it)%pair_exclusions, (/na,na/))
     if (allocated(molecule_type_data(it)%evb_reactive_protons)) then
        call dump_i4(trim(arg)//'reactive_protons', molecule_type_data(it)%evb_reactive_protons, (/na/))
     end if
     if (allocated(molecule_type_data(it)%evb_reactive_basic_atoms)) then
        call dump_i4(trim(arg)//'reactive_basic', molecule_type_data(it)%evb_reactive_basic_atoms, (/na/))
     end if
  end do

  !----- lookup tables built by initialize_energy_force (src/initialize_routines.f90:212-264)
  call dump_r8('B6_spline', PME_data%B6_spline, (/PME_data%spline_grid/))
  call dump_r8('B5_spline', PME_data%B5_spline, (/PME_data%spline_grid/))
  call dump_r8('erfc_table', PME_data%erfc_table, (/PME_data%erfc_grid+1/))
  call dump_r8('ewaldscale_table', PME_data%ewaldscale_table, (/PME_data%erfc_grid+1/))
  call dump_r8('Tang_Toennies_table', Tang_Toennies_table, (/4,Tang_Toennies_grid/))
  call dump_r8('dTang_Toennies_table', dTang_Toennies_table, (/4,Tang_Toennies_grid/))
  call dump_r8('CB', PME_data%CB, (/K,K,K/))
  call dump_s('Ewald_self', PME_data%Ewald_self)
  call dump_s('volume', system_data%volume)

  !----- Verlet list as built by construct_verlet_list_grid (src/general_routines.f90:1408)
  call dump_i4('verlet_point', verlet_list_data%verlet_point, (/N+1/))
  call dump_i4('neighbor_list', verlet_list_data%neighbor_list, (/verlet_list_data%verlet_point(N+1)-1/))

  !----- EVB parameter tables (src/glob_v.f90:54-120)
  if (ms_evb_simulation == "yes") then
     call dump_i4('evb_donor_acceptor_interaction', evb_donor_acceptor_interaction, (/max_interaction_type,3/))
     call dump_r8('evb_donor_acceptor_parameters', evb_donor_acceptor_parameters, (/max_interaction_type,6/))
     call dump_i4('evb_proton_acceptor_interaction', evb_proton_acceptor_interaction, (/max_interaction_type,2/))
     call dump_r8('evb_proton_acceptor_parameters', evb_proton_acceptor_parameters, (/max_interaction_type,5/))
     call dump_i4('evb_diabat_coupling_interaction', evb_diabat_coupling_interaction, (/max_interaction_type,3/))
     call dump_r8('evb_diabat_coupling_parameters', evb_diabat_coupling_parameters, (/max_interaction_type,10/))
     call dump_i4('evb_diabat_coupling_type', evb_diabat_coupling_type, (/max_interaction_type/))
     call dump_r8('evb_exchange_charge_atomic', evb_exchange_charge_atomic, (/MAX_N_ATOM_TYPE/))
     call dump_r8('evb_exchange_charge_proton', evb_exchange_charge_proton, (/MAX_N_MOLE_TYPE,MAX_N_MOLE_TYPE/))
     call dump_i4('evb_acid_molecule', evb_acid_molecule, (/MAX_N_MOLE_TYPE/))
     call dump_i4('evb_basic_molecule', evb_basic_molecule, (/MAX_N_MOLE_TYPE/))
     call dump_i4('evb_conjugate_pairs', evb_conjugate_pairs, (/MAX_N_MOLE_TYPE/))
     call dump_i4('evb_conjugate_atom_index', evb_conjugate_atom_index, (/MAX_N_ATOM_TYPE/))
     call dump_r8('evb_reference_energy', evb_reference_energy, (/MAX_N_MOLE_TYPE/))
     call dump_i4('evb_proton_index', evb_proton_index, (/MAX_N_MOLE_TYPE/))
     call dump_i4('evb_heavy_acid_index', evb_heavy_acid_index, (/MAX_N_MOLE_TYPE/))
     call dump_is('hydronium_molecule', hydronium_molecule_index(1))
     call dump_is('evb_max_chain', evb_max_chain)
  end if

  !----- stage 1: inter-molecular pair loop only
  call zero_all()
  call pairwise_real_space_verlet( system_data , atom_data , verlet_list_data , PME_data )
  call dump_r8('F_pair', atom_data%force, (/3,N/))
  call dump_s('E_elec_pair', system_data%E_elec)
  call dump_s('E_vdw_pair', system_data%E_vdw)

  !----- stage 2: intra-molecular non-bonded (loop of src/pair_int_real_space.f90:101-113)
  call zero_all()
  do i_mole=1,M
     if ( molecule_data(i_mole)%n_atom > 1) then
        call return_molecule_block( smd , molecule_data(i_mole)%n_atom, molecule_data(i_mole)%atom_index , atom_xyz=atom_data%xyz , atom_force=atom_data%force, atom_charge=atom_data%charge, atom_type_index=atom_data%atom_type_index )
        call intra_molecular_pairwise_energy_force( smd%force, system_data%E_elec, system_data%E_vdw,  smd ,  molecule_data(i_mole)%molecule_type_index , molecule_data(i_mole)%n_atom, PME_data )
        call dissociate_single_molecule_data(smd)
     end if
  end do
  call dump_r8('F_intra_nb', atom_data%force, (/3,N/))
  call dump_s('E_elec_intra_nb', system_data%E_elec)
  call dump_s('E_vdw_intra_nb', system_data%E_vdw)

  !----- stage 3: PME reciprocal space
  call zero_all()
  call pme_reciprocal_space_energy_force( system_data, atom_data, PME_data )
  call dump_r8('F_recip', PME_data%force_recip, (/3,N/))
  call dump_s('E_recip', system_data%E_elec)
  call dump_r8('Q_grid', PME_data%Q_grid, (/K,K,K/))
  call dump_r8('theta_conv_Q', PME_data%theta_conv_Q, (/K,K,K/))
  allocate(rtmp(3,N))
  block
    real*8 :: kk(3,3)
    call construct_reciprocal_lattice_vector(kk, system_data%box)
    call dump_r8('kk', kk, (/3,3/))
    call create_scaled_direct_coordinates(rtmp, atom_data%xyz, N, kk, K)
    call dump_r8('xyz_scale', rtmp, (/3,N/))
  end block
  deallocate(rtmp)

  !----- stage 4: bonded terms
  call zero_all()
  call intra_molecular_energy_force( system_data, molecule_data, atom_data )
  call dump_r8('F_bonded', atom_data%force, (/3,N/))
  call dump_s('E_bond', system_data%E_bond)
  call dump_s('E_angle', system_data%E_angle)
  call dump_s('E_dihedral', system_data%E_dihedral)

  !----- stage 5: the full force call
  call calculate_total_force_energy( system_data, molecule_data, atom_data, verlet_list_data, PME_data )
  call dump_r8('F_total', atom_data%force, (/3,N/))
  call dump_s('E_total', system_data%potential_energy)
  call dump_s('E_elec_total', system_data%E_elec)
  call dump_s('E_vdw_total', system_data%E_vdw)

  !----- stage 6: EVB Hamiltonian (src/ms_evb.f90:375, :242)
  if (ms_evb_simulation == "yes") then
     ! stand-alone MS-EVB3 repulsion procedures on the principal topology (src/ms_evb.f90:2295, :2405, :2484): energy and force arrays of each term
     block
       real*8, allocatable :: frep(:,:)
       real*8 :: erep, swv(2,61), rr
       integer :: ks
       allocate(frep(3,N))
       frep = 0d0; erep = 0d0
       call ms_evb_three_atom_repulsion( frep, erep, hydronium_molecule_index(1), system_data, molecule_data, atom_data )
       call dump_s('rep3_E', erep)
       call dump_r8('rep3_F', frep, (/3,N/))
       frep = 0d0; erep = 0d0
       call ms_evb_born_mayer( frep, erep, hydronium_molecule_index(1), system_data, molecule_data, atom_data )
       call dump_s('bm_E', erep)
       call dump_r8('bm_F', frep, (/3,N/))
       do ks = 1, 61      ! ms_evb_repulsive_switch on r = 2.00, 2.02, ... 3.20 with rs = 2.4, rc = 3.0
          rr = 2d0 + 0.02d0 * dble(ks - 1)
          call ms_evb_repulsive_switch( swv(1,ks), swv(2,ks), rr, 2.4d0, 3.0d0 )
       end do
       call dump_r8('rep_switch_table', swv, (/2,61/))
       deallocate(frep)
     end block
     call construct_evb_hamiltonian( hydronium_molecule_index(1),  system_data, molecule_data, atom_data, verlet_list_data, PME_data, file_io_data )
     D = diabat_index
     call dump_is('n_diabat', D)
     call dump_i4('evb_diabat_proton_log', evb_diabat_proton_log, (/evb_max_states, evb_max_chain, 5/))
     call dump_i4('evb_diabat_coupling_matrix', evb_diabat_coupling_matrix, (/evb_max_states/))
     call dump_r8('evb_hamiltonian', evb_hamiltonian, (/evb_max_states, evb_max_states/))
     call dump_i4('evb_forces_lookup_index', evb_forces_lookup_index, (/evb_max_states, evb_max_states/))
     ns = maxval(evb_forces_lookup_index(1:D,1:D))
     call dump_is('n_store', ns)
     call dump_r8('evb_forces_store', evb_forces_store, (/3,N,ns/))
     call dump_r8('Q_grid_diabats', Q_grid_diabats, (/K,K,K,D/))
     call dump_s('E_recip_principal', PME_data%E_recip)
     call dump_r8('F_principal', atom_data%force, (/3,N/))
     call dump_s('E_principal', system_data%potential_energy)
     call diagonalize_evb_hamiltonian( principle_diabat, new_h,  hydronium_molecule_index(1), system_data%potential_energy, atom_data%force, file_io_data, integrator_data%n_output )
     call dump_is('principle_diabat', principle_diabat)
     call dump_is('new_hydronium_molecule', new_h)
     call dump_s('E_adiabatic', system_data%potential_energy)
     call dump_r8('F_adiabatic', atom_data%force, (/3,N/))
     deallocate( evb_forces_store )
     deallocate( Q_grid_diabats )
  end if

  !----- optional: NSTEPS of velocity Verlet from a seeded velocity file (src/md_integration.f90:438)
  if (nsteps > 0) then
     allocate(vel(3,N), etraj(2,nsteps), vflag(nsteps), ndia(nsteps))
     open(unit=78, file=trim(velfile), access='stream', form='unformatted', status='old')
     read(78) vel
     close(78)
     atom_data%velocity = vel
     ! forces of the current configuration are needed by the first half-kick
     if (ms_evb_simulation == "yes") then
        call ms_evb_calculate_total_force_energy( system_data, molecule_data, atom_data, verlet_list_data, PME_data, file_io_data, integrator_data%n_output, integrator_data, trajectory_step )
     else
        call calculate_total_force_energy( system_data, molecule_data, atom_data, verlet_list_data, PME_data )
     end if
     call dump_r8('md_F0', atom_data%force, (/3,N/))
     ! one line per step, flushed at once: a run the reference itself ends (its 'stop' on a force that is too big, src/md_integration.f90:523) leaves its history behind
     open(unit=79, file='md_trace.txt', status='replace')
     do i_step=1,nsteps
        trajectory_step = i_step
        call md_integrate_atomic( system_data, molecule_data, atom_data, integrator_data, verlet_list_data, PME_data, file_io_data )
        call calculate_kinetic_energy( system_data%kinetic_energy, N, atom_data%mass, atom_data%velocity, constants%conv_kJmol_ang2ps2gmol )
        etraj(1,i_step) = system_data%potential_energy
        etraj(2,i_step) = system_data%kinetic_energy
        ! flag_verlet_list as the force call of this step left it (1 = the NEXT force call rebuilds the list,
        ! src/total_energy_forces.f90:30-41) and the diabat count of this step's EVB call (src/ms_evb.f90:451)
        vflag(i_step) = verlet_list_data%flag_verlet_list
        ndia(i_step) = 1
        if (ms_evb_simulation == "yes") ndia(i_step) = diabat_index
        i_hyd_trace = 0
        if (ms_evb_simulation == "yes") i_hyd_trace = hydronium_molecule_index(1)
        write(79,'(I7,2ES26.17E3,2I4,ES26.17E3,I8)') i_step, etraj(1,i_step), etraj(2,i_step), vflag(i_step), ndia(i_step), sqrt(maxval(sum(atom_data%force**2, dim=1))), i_hyd_trace
        flush(79)
     end do
     close(79)
     call dump_i4('md_verlet_flag', vflag, (/nsteps/))
     call dump_i4('md_n_diabat', ndia, (/nsteps/))
     call dump_is('md_nsteps', nsteps)
     call dump_r8('md_energy', etraj, (/2,nsteps/))
     call dump_r8('md_E_terms', (/ system_data%E_elec, system_data%E_vdw, system_data%E_bond, system_data%E_angle, system_data%E_dihedral /), (/5/))
     call dump_r8('md_xyz', atom_data%xyz, (/3,N/))
     call dump_r8('md_velocity', atom_data%velocity, (/3,N/))
     call dump_r8('md_force', atom_data%force, (/3,N/))
     call dump_i4('md_atom_type_index', atom_data%atom_type_index, (/N/))
     ! topology after the steps: a principal-diabat switch (src/ms_evb.f90:218-227) re-orders the atom arrays
     call dump_r8('md_charge', atom_data%charge, (/N/))
     allocate(itmp(M))
     do i_mole = 1, M
        itmp(i_mole) = molecule_data(i_mole)%n_atom
     end do
     call dump_i4('md_mol_n_atom', itmp, (/M/))
     do i_mole = 1, M
        itmp(i_mole) = molecule_data(i_mole)%molecule_type_index
     end do
     call dump_i4('md_mol_type_index', itmp, (/M/))
     deallocate(itmp)
     if (ms_evb_simulation == "yes") call dump_is('md_hydronium_molecule', hydronium_molecule_index(1))
     !----- the reference's own writers on the end state: trajectory frame + log block (print_step, src/general_routines.f90:870) and the
     !      velocity checkpoint (print_velocities_checkpoint, :997); read back as text by oracle/ref_inputs.py
     call print_step( file_io_data%ofile_traj_file_h, file_io_data%ofile_log_file_h, nsteps, system_data, integrator_data, molecule_data, atom_data )
     open(unit=file_io_data%ifile_velocity_file_h, file='vel_checkpoint.out', status='replace')
     call print_velocities_checkpoint( file_io_data%ifile_velocity_file_h, nsteps, integrator_data%delta_t, system_data, molecule_data, atom_data )
     close(file_io_data%ifile_velocity_file_h)
     flush(file_io_data%ofile_traj_file_h)
     flush(file_io_data%ofile_log_file_h)
  end if

  !----- optional: Monte-Carlo barostat moves (src/md_integration.f90:231-386), each followed by one MD step as in mc_sample (:28-37)
  if (nbaro > 0) then
     call random_seed(size=nseed)
     allocate(rstate(nseed), baro_rand(2,nbaro), baro_box(nbaro), baro_epot(nbaro), baro_scale(nbaro), baro_xyz(3,N,nbaro), baro_epot_md(nbaro))
     rstate = 20240901
     call random_seed(put=rstate)
     call dump_s('baro_pressure', system_data%pressure)
     call dump_s('baro_temperature', system_data%temperature)
     call dump_s('baro_scale0', system_data%baroscale)
     call dump_s('baro_box0', system_data%box(1,1))
     call dump_s('baro_epot0', system_data%potential_energy)
     do i_baro = 1, nbaro
        ! the two numbers the move will draw (trial length, Metropolis test): peek at the generator, put its state back
        call random_seed(get=rstate)
        call random_number(r12)
        call random_seed(put=rstate)
        baro_rand(:,i_baro) = r12
        call monte_carlo_barostat( system_data, molecule_data, atom_data, integrator_data, verlet_list_data, PME_data, file_io_data )
        baro_box(i_baro) = system_data%box(1,1)
        baro_epot(i_baro) = system_data%potential_energy
        baro_scale(i_baro) = system_data%baroscale
        baro_xyz(:,:,i_baro) = atom_data%xyz
        trajectory_step = nsteps + i_baro
        call md_integrate_atomic( system_data, molecule_data, atom_data, integrator_data, verlet_list_data, PME_data, file_io_data )
        baro_epot_md(i_baro) = system_data%potential_energy
     end do
     call dump_is('baro_n', nbaro)
     call dump_r8('baro_rand', baro_rand, (/2,nbaro/))
     call dump_r8('baro_box', baro_box, (/nbaro/))
     call dump_r8('baro_epot', baro_epot, (/nbaro/))
     call dump_r8('baro_scale', baro_scale, (/nbaro/))
     call dump_r8('baro_xyz', baro_xyz, (/3,N,nbaro/))
     call dump_r8('baro_epot_md', baro_epot_md, (/nbaro/))
     call dump_r8('baro_end_xyz', atom_data%xyz, (/3,N/))
     call dump_r8('baro_end_velocity', atom_data%velocity, (/3,N/))
  end if

  call dump_close()
  write(*,*) "ref_dump finished OK"

contains
  subroutine zero_all()
    atom_data%force = 0d0
    system_data%potential_energy = 0d0
    system_data%E_elec = 0d0; system_data%E_vdw = 0d0
    system_data%E_bond = 0d0; system_data%E_angle = 0d0; system_data%E_dihedral = 0d0
  end subroutine
end program ref_dump
