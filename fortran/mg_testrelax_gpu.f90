!> Solver-level harness in the shape of the reference's unit program src/old_tests/mg_testrelax.f90 (:97-160): after the set-up it
!> drives the operators one by one THROUGH `use nhydro` -- per level: set p and b, fill_halo, relax(lev,nsweeps),
!> compute_residual(lev,res); then, on level 1: fine2coarse / coarse2fine, Vcycle, Vcycle2, Fcycle, and solve_p(tol,maxite) -- all of
!> them resolved by fortran/mg_solvers.f90 + libmgx.so (HIP kernels).  tic / toc / print_tictoc bracket the program as the reference's
!> drivers do (mg_testseamount.f90:37,220-221).  Every number is printed with 17 digits for tests/test_gpu_parity.py, which repeats the
!> same sequence on the CPU oracle.
!> The fields are filled with a closed rational pattern (no libm): p(k,j,i) = mod(7k+3j+5i+lev,11)/11 - 1/2, b = mod(5k+7j+3i+2lev,13)/13 - 1/2.
program mg_testrelax_gpu
  use nhydro
  implicit none
  integer(kind=ip) :: nx, ny, nz, lev, nxl, nyl, nzl, i, j, k, narg, nsweeps
  real(kind=rp) :: Lx, Ly, Htot, hc, theta_b, theta_s, x, y, x0, y0, res
  real(kind=rp), dimension(:,:), pointer :: dx, dy, zeta, h, rmask
  real(kind=rp), dimension(:,:,:), allocatable :: u, v, w, p, b
  character(len=32) :: arg

  call tic(1, 'mg_testrelax')

  nx = 64; ny = 32; nz = 16; nsweeps = 2
  narg = command_argument_count()
  if (narg >= 3) then
     call get_command_argument(1, arg); read(arg,*) nx
     call get_command_argument(2, arg); read(arg,*) ny
     call get_command_argument(3, arg); read(arg,*) nz
  endif

  call nhydro_init(nx, ny, nz, 1, 1)
  write(*,'(A,I3,A,L2,A,I3)') 'nlevs = ', nlevs, ' netcdf_output = ', netcdf_output, ' myrank = ', myrank

  Lx = 1.e4_8; Ly = 1.e4_8; Htot = 4.e3_8
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
  call nhydro_matrices(dx, dy, zeta, h, rmask, hc, theta_b, theta_s)

  ! ---- one relax call and one residual per level (mg_testrelax.f90:97-141 runs its loop on the levels it is asked for)
  do lev = 1, nlevs
     call grid_dims(lev, nxl, nyl, nzl)
     allocate(p(nzl,0:nyl+1,0:nxl+1), b(nzl,0:nyl+1,0:nxl+1))
     p = 0._8; b = 0._8
     do i = 1, nxl
        do j = 1, nyl
           do k = 1, nzl
              p(k,j,i) = real(mod(7*k + 3*j + 5*i + lev, 11), kind=8)/11._8 - 0.5_8
              b(k,j,i) = real(mod(5*k + 7*j + 3*i + 2*lev, 13), kind=8)/13._8 - 0.5_8
           enddo
        enddo
     enddo
     call grid_set(lev, 'p', p)
     call grid_set(lev, 'b', b)
     call fill_halo(lev, 'p')
     call relax(lev, nsweeps)
     call compute_residual(lev, res)
     call grid_get(lev, 'p', p)
     write(*,'(A,I2,A,ES24.16,A,ES24.16)') 'lev=', lev, ' res=', res, ' sum_p2=', sum(p(1:nzl,1:nyl,1:nxl)**2)
     deallocate(p, b)
  enddo

  ! ---- transfers and cycles on the state the loop left (level 1 holds its relaxed p and the residual r)
  allocate(p(nz,0:ny+1,0:nx+1), b(nz/2,0:ny/2+1,0:nx/2+1))
  call fine2coarse(1)
  call grid_get(2, 'b', b)
  write(*,'(A,ES24.16)') 'f2c_sum_b2 = ', sum(b(1:nz/2,1:ny/2,1:nx/2)**2)
  call coarse2fine(1)
  call grid_get(1, 'p', p)
  write(*,'(A,ES24.16)') 'c2f_sum_p2 = ', sum(p(1:nz,1:ny,1:nx)**2)
  call Vcycle(1)
  call compute_residual(1, res)
  write(*,'(A,ES24.16)') 'vcycle_res = ', res
  if (nlevs >= 3) then
     call Vcycle2(1, 3)
     call compute_residual(1, res)
     write(*,'(A,ES24.16)') 'vcycle2_res = ', res
  endif
  call Fcycle()
  call compute_residual(1, res)
  write(*,'(A,ES24.16)') 'fcycle_res = ', res

  ! ---- solve_p on the resting column (b from compute_rhs, as nhydro_solve forms it: nhydro_check_nondivergence is that call alone)
  allocate(u(1:nx+1,0:ny+1,1:nz), v(0:nx+1,1:ny+1,1:nz), w(0:nx+1,0:ny+1,0:nz))
  u = 0._8; v = 0._8; w(:,:,0) = 0._8; w(:,:,1:nz) = -1._8
  call nhydro_check_nondivergence(nx, ny, nz, rmask, u, v, w)
  call solve_p(1.e-8_8, 5)
  call grid_get(1, 'p', p)
  write(*,'(A,I3,A,ES24.16,A,ES24.16)') 'solve_p_nite = ', solve_p_nite, ' res = ', solve_p_res, ' sum_p2 = ', sum(p(1:nz,1:ny,1:nx)**2)

  call nhydro_clean()
  call toc(1, 'mg_testrelax')
  if (myrank == 0) call print_tictoc(myrank)
end program mg_testrelax_gpu
