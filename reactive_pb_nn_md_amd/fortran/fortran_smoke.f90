!==============================================================================
! Stand-alone Fortran driver (no reference module needed): reads a system
! written by tests/test_gpu_fortran.py, drives libmsevb_hip.so through the
! iso_c_binding interfaces and prints energies + a force checksum.
!==============================================================================
program fortran_smoke
  use iso_c_binding
  use msevb_hip_c
  implicit none
  type(c_ptr) :: ctx
  integer :: u, N, M, nt, nmt, K, nang, ndih, it, na, nb, nan, nd, i
  real(c_double) :: box(3,3), rc, rv, alpha, e(MSEVB_N_ENERGY)
  integer(c_int), allocatable :: atype(:), mfirst(:), mnat(:), mtype(:), vt(:), bt(:), ai(:,:), di(:,:), mat(:), bl(:,:), al(:,:), dl(:,:), ex(:), rp(:), rb(:)
  real(c_double), allocatable :: chg(:), mass(:), vp(:), vp14(:), bp(:), achg(:), ap(:,:), dp(:,:), xyz(:,:), force(:,:)
  character(300) :: fname

  call get_command_argument(1, fname)
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
  close(u)
  call msevb_check(ctx, msevb_finalize(ctx), "finalize")
  call msevb_check(ctx, msevb_set_positions(ctx, xyz), "set_positions")
  call msevb_check(ctx, msevb_calculate_total_force_energy(ctx, e, force), "calculate_total_force_energy")
  write(*,'(A,6ES24.16)') "ENERGIES ", (e(i), i=1,6)
  write(*,'(A,3ES24.16)') "FORCESUM ", sum(abs(force(1,:))), sum(abs(force(2,:))), sum(abs(force(3,:)))
  call msevb_check(ctx, msevb_destroy(ctx), "destroy")
end program fortran_smoke
