!> Fortran harness in the shape of the reference's driver (src/mg_testseamount.f90:44-213): sets up the analytic
!> seamount, u=v=0, w=-1, and calls nhydro_init / nhydro_matrices / nhydro_solve / nhydro_check_nondivergence /
!> nhydro_clean -- here resolved by fortran/nhydro.f90 + libmgx.so.  Prints sum(p**2), sum(b**2 after correction).
program mg_testseamount_gpu
  use iso_c_binding
  use nhydro
  implicit none
  integer(kind=4) :: nx, ny, nz, i, j, rc, narg
  real(kind=8) :: Lx, Ly, Htot, hc, theta_b, theta_s, x, y, x0, y0
  real(kind=8), dimension(:,:), pointer :: dx, dy, zeta, h, rmask
  real(kind=8), dimension(:,:,:), allocatable :: u, v, w, p, b
  character(len=32) :: arg

  nx = 64; ny = 64; nz = 16                  ! mg_testseamount.f90:44-49 hard-codes its sizes too
  narg = command_argument_count()
  if (narg >= 3) then
     call get_command_argument(1, arg); read(arg,*) nx
     call get_command_argument(2, arg); read(arg,*) ny
     call get_command_argument(3, arg); read(arg,*) nz
  endif

  call nhydro_init(nx, ny, nz, 1, 1)

  Lx = 1.e4_8; Ly = 1.e4_8; Htot = 4.e3_8    ! :76-82
  hc = 4.e3_8; theta_b = 0._8; theta_s = 0._8
  allocate(dx(0:ny+1,0:nx+1), dy(0:ny+1,0:nx+1), zeta(0:ny+1,0:nx+1), h(0:ny+1,0:nx+1), rmask(0:ny+1,0:nx+1))
  dx(:,:) = Lx/real(nx,kind=8); dy(:,:) = Ly/real(ny,kind=8); zeta(:,:) = 0._8; rmask(:,:) = 1._8
  x0 = Lx*0.5_8; y0 = Ly*0.5_8
  do i = 0, nx+1                              ! mg_setup_tests.f90:139-148
     do j = 0, ny+1
        x = (real(i,kind=8)-0.5_8)*dx(j,i)
        y = (real(j,kind=8)-0.5_8)*dy(j,i)
        h(j,i) = Htot*(1._8 - 0.5_8*exp(-(x-x0)**2._8/(Lx/5._8)**2._8 - (y-y0)**2._8/(Ly/5._8)**2._8))
     enddo
  enddo
  if (bmask) call fill_halo_2D_bmask(1, rmask)   ! mask the boundaries, as mg_testseamount.f90 does after allocate(rmask)
  call nhydro_matrices(dx, dy, zeta, h, rmask, hc, theta_b, theta_s)

  allocate(u(1:nx+1,0:ny+1,1:nz), v(0:nx+1,1:ny+1,1:nz), w(0:nx+1,0:ny+1,0:nz))
  u = 0._8; v = 0._8; w(:,:,0) = 0._8; w(:,:,1:nz) = -1._8   ! :119-123
  call nhydro_solve(nx, ny, nz, rmask, u, v, w)

  allocate(p(nz,0:ny+1,0:nx+1), b(nz,0:ny+1,0:nx+1))
  rc = mgx_get_field(1, 0, p)
  write(*,'(A,ES24.16)') 'sum_p2 = ', sum(p(1:nz,1:ny,1:nx)**2)
  call nhydro_check_nondivergence(nx, ny, nz, rmask, u, v, w)
  rc = mgx_get_field(1, 1, b)
  write(*,'(A,ES24.16)') 'sum_div2 = ', sum(b(1:nz,1:ny,1:nx)**2)
  call nhydro_clean()
end program mg_testseamount_gpu
