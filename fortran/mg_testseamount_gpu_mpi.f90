!> The reference's parallel driver shape (src/mg_testseamount.f90:44-213: MPI_Init, npxg x npyg ranks, local nx x ny x nz
!> block per rank) over module nhydro -> libmgx.so, with the MPI hooks of fortran/mgx_mpi_hooks.cpp.
!> Usage: mpiexec -n (npx*npy) testseamount_gpu_mpi npx npy nx ny nz [p2p|rccl]      (nx,ny = LOCAL sizes)
!>   p2p : the cycle's halos and gathers as device-side pushes between the ranks' GPUs (hipIpc), MPI hooks for the rest
!>   rccl: libmgx.so's native RCCL transport instead of the MPI hooks (one GPU per rank)
program mg_testseamount_gpu_mpi
  use iso_c_binding
  use nhydro
  implicit none
  include 'mpif.h'
  interface
     integer(c_int) function mgx_mpi_install(fcomm) bind(C, name='mgx_mpi_install')
       import :: c_int
       integer(c_int), value :: fcomm
     end function mgx_mpi_install
     integer(c_int) function mgx_mpi_connect_p2p() bind(C, name='mgx_mpi_connect_p2p')
       import :: c_int
     end function mgx_mpi_connect_p2p
     integer(c_int) function mgx_mpi_connect_rccl() bind(C, name='mgx_mpi_connect_rccl')
       import :: c_int
     end function mgx_mpi_connect_rccl
     integer(c_int) function mgx_rccl_selftest() bind(C, name='mgx_rccl_selftest')
       import :: c_int
     end function mgx_rccl_selftest
  end interface
  integer(kind=4) :: nx, ny, nz, npx, npy, i, j, rc, ierr, nprocs, pi, pj, use_p2p, use_rccl   ! myrank: module variable (mg_mpi, through nhydro)
  real(kind=8) :: Lx, Ly, Htot, hc, theta_b, theta_s, x, y, x0, y0, s_loc, s_glo
  real(kind=8), dimension(:,:), pointer :: dx, dy, zeta, h, rmask
  real(kind=8), dimension(:,:,:), allocatable :: u, v, w, p, b
  character(len=32) :: arg

  call MPI_Init(ierr)
  call MPI_Comm_rank(MPI_COMM_WORLD, myrank, ierr)
  call MPI_Comm_size(MPI_COMM_WORLD, nprocs, ierr)
  call get_command_argument(1, arg); read(arg,*) npx
  call get_command_argument(2, arg); read(arg,*) npy
  call get_command_argument(3, arg); read(arg,*) nx
  call get_command_argument(4, arg); read(arg,*) ny
  call get_command_argument(5, arg); read(arg,*) nz
  use_p2p = 0; use_rccl = 0
  if (command_argument_count() >= 6) then
     call get_command_argument(6, arg)
     if (trim(arg) == 'rccl') then
        use_rccl = 1
     else
        use_p2p = 1
     endif
  endif
  if (npx*npy /= nprocs) then
     write(*,*) 'Error: npx*npy /= number of MPI ranks'; stop -1
  endif
  pi = mod(myrank, npx); pj = myrank/npx       ! mg_grids.f90:593-594

  call mgx_check(mgx_mpi_install(MPI_COMM_WORLD), 'mgx_mpi_install')
  if (use_rccl == 1) then
     rc = mgx_mpi_connect_rccl()
     if (myrank == 0) write(*,'(A,I2)') 'rccl_connected = ', 1 - rc
  endif
  call nhydro_init(nx, ny, nz, npx, npy)
  if (use_rccl == 1) then
     rc = mgx_rccl_selftest()
     if (myrank == 0) write(*,'(A,I2)') 'rccl_selftest_ok = ', 1 - min(rc, 1)
  endif
  if (use_p2p == 1) then
     rc = mgx_mpi_connect_p2p()
     if (myrank == 0) write(*,'(A,I2)') 'p2p_connected = ', 1 - rc
  endif

  Lx = 1.e4_8; Ly = 1.e4_8; Htot = 4.e3_8
  hc = 4.e3_8; theta_b = 0._8; theta_s = 0._8
  allocate(dx(0:ny+1,0:nx+1), dy(0:ny+1,0:nx+1), zeta(0:ny+1,0:nx+1), h(0:ny+1,0:nx+1), rmask(0:ny+1,0:nx+1))
  dx(:,:) = Lx/real(nx*npx,kind=8); dy(:,:) = Ly/real(ny*npy,kind=8); zeta(:,:) = 0._8; rmask(:,:) = 1._8
  x0 = Lx*0.5_8; y0 = Ly*0.5_8
  do i = 0, nx+1                              ! mg_setup_tests.f90:139-148
     do j = 0, ny+1
        x = (real(i+pi*nx,kind=8)-0.5_8)*dx(j,i)
        y = (real(j+pj*ny,kind=8)-0.5_8)*dy(j,i)
        h(j,i) = Htot*(1._8 - 0.5_8*exp(-(x-x0)**2._8/(Lx/5._8)**2._8 - (y-y0)**2._8/(Ly/5._8)**2._8))
     enddo
  enddo
  if (bmask) call fill_halo_2D_bmask(1, rmask)   ! mask the physical boundaries (mg_testseamount.f90, fill_halo_2D_bmask)
  call nhydro_matrices(dx, dy, zeta, h, rmask, hc, theta_b, theta_s)

  allocate(u(1:nx+1,0:ny+1,1:nz), v(0:nx+1,1:ny+1,1:nz), w(0:nx+1,0:ny+1,0:nz))
  u = 0._8; v = 0._8; w(:,:,0) = 0._8; w(:,:,1:nz) = -1._8
  call nhydro_solve(nx, ny, nz, rmask, u, v, w)

  allocate(p(nz,0:ny+1,0:nx+1), b(nz,0:ny+1,0:nx+1))
  rc = mgx_get_field(1, 0, p)
  s_loc = sum(p(1:nz,1:ny,1:nx)**2)
  call MPI_Allreduce(s_loc, s_glo, 1, MPI_DOUBLE_PRECISION, MPI_SUM, MPI_COMM_WORLD, ierr)
  if (myrank == 0) write(*,'(A,ES24.16)') 'sum_p2 = ', s_glo
  call nhydro_check_nondivergence(nx, ny, nz, rmask, u, v, w)
  rc = mgx_get_field(1, 1, b)
  s_loc = sum(b(1:nz,1:ny,1:nx)**2)
  call MPI_Allreduce(s_loc, s_glo, 1, MPI_DOUBLE_PRECISION, MPI_SUM, MPI_COMM_WORLD, ierr)
  if (myrank == 0) write(*,'(A,ES24.16)') 'sum_div2 = ', s_glo
  call nhydro_clean()
  call MPI_Finalize(ierr)
end program mg_testseamount_gpu_mpi
