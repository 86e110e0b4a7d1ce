!==============================================================================
! Stand-alone Fortran driver (no reference module needed): reads a system
! written by tests/test_gpu_fortran.py, drives libmsevb_hip.so through the
! iso_c_binding interfaces (msevb_hip_c.f90) and prints what the test compares
! with the committed reference fixtures:
!   ENERGIES / FORCESUM      calculate_total_force_energy           (src/total_energy_forces.f90:19)
!   EVB / EVBFORCESUM        ms_evb_calculate_total_force_energy    (src/ms_evb.f90:181), when the file carries the EVB tables
!   HOSTORDER                the same EVB call with the host arrays in ANOTHER atom order (what the reference's arrays are after
!                            evb_change_diabat_data_structure_topology, src/ms_evb.f90:843): msevb_set_host_order
!   MD                       one block of md_integrate_atomic        (src/md_integration.f90:438), end state written to argument 2
! Everything the GPU tests need from the Fortran side of the boundary, with no reference-derived object in the executable.
!==============================================================================
program fortran_smoke
  use iso_c_binding
  use msevb_hip_c
  implicit none
  type(c_ptr) :: ctx
  integer :: u, N, M, nt, nmt, K, nang, ndih, it, na, nb, nan, nd, i, j, has_evb, has_md, nsteps, use_evb, hyd, max_chain, ldi
  integer(c_int) :: n_diabat, principal, new_hyd
  real(c_double) :: box(3,3), rc, rv, alpha, e(MSEVB_N_ENERGY), e_evb, e_evb2, dt, dmax
  integer(c_int), allocatable :: atype(:), mfirst(:), mnat(:), mtype(:), vt(:), bt(:), ai(:,:), di(:,:), mat(:), bl(:,:), al(:,:), dl(:,:), ex(:), rp(:), rb(:)
  integer(c_int), allocatable :: acid(:), base(:), conj(:), pidx(:), hidx(:), catom(:), da_i(:,:), pa_i(:,:), cp_i(:,:), cp_t(:), perm(:), ident(:)
  real(c_double), allocatable :: chg(:), mass(:), vp(:), vp14(:), bp(:), achg(:), ap(:,:), dp(:,:), xyz(:,:), force(:,:), eref(:), exa(:), exp_(:,:), da_p(:,:), pa_p(:,:), cp_p(:,:)
  real(c_double), allocatable :: xyz_p(:,:), force_p(:,:), vel(:,:), en(:,:)
  character(300) :: fname, oname

  call get_command_argument(1, fname)
  call get_command_argument(2, oname)
  u = 31
  open(unit=u, file=trim(fname), access='stream', form='unformatted', status='old')
  read(u) N, M, nt, nmt, K
  read(u) box, rc, rv, alpha
  allocate(atype(N), chg(N), mass(N), mfirst(M), mnat(M), mtype(M), xyz(3,N), force(3,N))
  read(u) atype, chg, mass, mfirst, mnat, mtype, xyz
  allocate(vt(nt*nt), vp(nt*nt*6), vp14(nt*nt*6), bt(nt*nt), bp(nt*nt*3), achg(nt))
  read(u) vt, vp, vp14, bt, bp, achg
  read(u) nang
  allocate(ai(4,nang), ap(2,nang)); read(u) ai, ap
  read(u) ndih
  allocate(di(5,ndih), dp(6,ndih)); read(u) di, dp
  call msevb_check(ctx, msevb_create(ctx, 0), "create")
  call msevb_check(ctx, msevb_set_system(ctx, N, M, box, rc, rv, alpha, K, 6, 100000, 100000), "set_system")
  call msevb_check(ctx, msevb_set_atoms(ctx, atype, chg, mass, mfirst, mnat, mtype), "set_atoms")
  call msevb_check(ctx, msevb_set_forcefield(ctx, nt, nt, vt, vp, vp14, bt, bp, achg), "set_forcefield")
  call msevb_check(ctx, msevb_set_angle_types(ctx, nang, ai, ap), "set_angle_types")
  call msevb_check(ctx, msevb_set_dihedral_types(ctx, ndih, di, dp), "set_dihedral_types")
  do it = 1, nmt
     read(u) na, nb, nan, nd
     allocate(mat(na), bl(2,max(nb,1)), al(3,max(nan,1)), dl(4,max(nd,1)), ex(na*na), rp(na), rb(na))
     read(u) mat
     if (nb > 0) read(u) bl(:,1:nb)
     if (nan > 0) read(u) al(:,1:nan)
     if (nd > 0) read(u) dl(:,1:nd)
     read(u) ex, rp, rb
     call msevb_check(ctx, msevb_set_molecule_type(ctx, it, na, mat, nb, bl, nan, al, nd, dl, ex, rp, rb), "set_molecule_type")
     deallocate(mat, bl, al, dl, ex, rp, rb)
  end do
  ! ---- optional blocks (absent in a file of the first kind: end of file)
  has_evb = 0; has_md = 0
  read(u, end=10) has_evb
  if (has_evb == 1) then
     read(u) ldi
     allocate(acid(nmt), base(nmt), conj(nmt), pidx(nmt), hidx(nmt), catom(nt), eref(nmt), exa(nt), exp_(nmt,nmt))
     allocate(da_i(ldi,3), da_p(ldi,6), pa_i(ldi,2), pa_p(ldi,5), cp_i(ldi,3), cp_p(ldi,10), cp_t(ldi))
     read(u) acid, base, conj, pidx, hidx, catom, eref, exa, exp_, da_i, da_p, pa_i, pa_p, cp_i, cp_p, cp_t, hyd, max_chain
     call msevb_check(ctx, msevb_set_evb(ctx, nmt, nmt, ldi, acid, base, conj, pidx, hidx, catom, eref, exa, exp_, da_i, da_p, pa_i, pa_p, cp_i, cp_p, cp_t, hyd, max_chain), "set_evb")
  end if
  read(u, end=10) has_md
  if (has_md == 1) then
     allocate(vel(3,N), perm(N))
     read(u) nsteps, dt, use_evb, vel, perm
  end if
10 continue
  close(u)
  call msevb_check(ctx, msevb_finalize(ctx), "finalize")
  call msevb_check(ctx, msevb_set_positions(ctx, xyz), "set_positions")
  call msevb_check(ctx, msevb_calculate_total_force_energy(ctx, e, force), "calculate_total_force_energy")
  write(*,'(A,6ES24.16)') "ENERGIES ", (e(i), i=1,6)
  write(*,'(A,3ES24.16)') "FORCESUM ", sum(abs(force(1,:))), sum(abs(force(2,:))), sum(abs(force(3,:)))

  if (has_evb == 1) then
     ! ---- the MS-EVB force call, then the same call with the caller's arrays permuted (a proton hop seen from the host side)
     call msevb_check(ctx, msevb_ms_evb_calculate_total_force_energy(ctx, e_evb, force), "ms_evb_calculate_total_force_energy")
     call msevb_check(ctx, msevb_get_evb_state(ctx, n_diabat, c_null_ptr, c_null_ptr, c_null_ptr, c_null_ptr, principal, new_hyd), "get_evb_state")
     write(*,'(A,ES24.16,3I6)') "EVB ", e_evb, n_diabat, principal, new_hyd
     write(*,'(A,3ES24.16)') "EVBFORCESUM ", sum(abs(force(1,:))), sum(abs(force(2,:))), sum(abs(force(3,:)))
     if (len_trim(oname) > 0) then
        open(unit=32, file=trim(oname)//".evb", access='stream', form='unformatted', status='replace')
        write(32) force
        close(32)
     end if
     if (has_md == 1) then
        allocate(xyz_p(3,N), force_p(3,N), ident(N))
        do i = 1, N
           xyz_p(:,i) = xyz(:,perm(i)); ident(i) = i
        end do
        call msevb_check(ctx, msevb_set_host_order(ctx, perm), "set_host_order")
        call msevb_check(ctx, msevb_set_positions(ctx, xyz_p), "set_positions (host order)")
        call msevb_check(ctx, msevb_ms_evb_calculate_total_force_energy(ctx, e_evb2, force_p), "ms_evb_calculate_total_force_energy (host order)")
        dmax = 0d0
        do i = 1, N
           do j = 1, 3
              dmax = max(dmax, abs(force_p(j,i) - force(j,perm(i))))
           end do
        end do
        write(*,'(A,2ES24.16)') "HOSTORDER ", dmax, abs(e_evb2 - e_evb)
        call msevb_check(ctx, msevb_get_positions(ctx, force_p), "get_positions (host order)")
        write(*,'(A,ES24.16)') "HOSTORDERXYZ ", maxval(abs(force_p - xyz_p))
        call msevb_check(ctx, msevb_set_host_order(ctx, ident), "set_host_order (identity)")
        call msevb_check(ctx, msevb_set_positions(ctx, xyz), "set_positions")
     end if
  end if

  if (has_md == 1 .and. nsteps > 0) then
     ! ---- one block of md_integrate_atomic from the fixture's velocities: first the force call of the start-up, then nsteps on the device
     allocate(en(2,nsteps))
     call msevb_check(ctx, msevb_set_velocities(ctx, vel), "set_velocities")
     if (use_evb == 1) then
        call msevb_check(ctx, msevb_ms_evb_calculate_total_force_energy(ctx, e_evb, force), "ms_evb_calculate_total_force_energy")
     else
        call msevb_check(ctx, msevb_calculate_total_force_energy(ctx, e, force), "calculate_total_force_energy")
     end if
     call msevb_check(ctx, msevb_md_integrate_atomic(ctx, nsteps, dt, use_evb, en), "md_integrate_atomic")
     do i = 1, nsteps
        write(*,'(A,I6,2ES24.16)') "MD ", i, en(1,i), en(2,i)
     end do
     call msevb_check(ctx, msevb_get_positions(ctx, xyz), "get_positions")
     call msevb_check(ctx, msevb_get_velocities(ctx, vel), "get_velocities")
     call msevb_check(ctx, msevb_get_forces(ctx, force), "get_forces")
     open(unit=32, file=trim(oname), access='stream', form='unformatted', status='replace')
     write(32) xyz, vel, force
     close(32)
  end if
  call msevb_check(ctx, msevb_destroy(ctx), "destroy")
  write(*,'(A)') "fortran_smoke done"
end program fortran_smoke
