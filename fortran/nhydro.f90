!> Drop-in replacement of the reference's model-facing module (src/nhydro.f90): same module name, same five
!> procedures with the same argument lists, bodies forwarding through ISO_C_BINDING to libmgx.so (include/mgx.h),
!> whose operators are HIP kernels on the MI355X.  An ocean model (or the reference's own mg_testseamount driver)
!> that does `use nhydro` links against this object + libmgx.so instead of the reference's solver modules.
!>
!> Like the reference's module, it re-exports what it uses (fortran/mg_solvers.f90): the solver-level procedures solve_p, Fcycle,
!> Vcycle, Vcycle2, testgalerkin, relax, compute_residual, fine2coarse, coarse2fine, fill_halo, tic / toc / print_tictoc, and the
!> module variables myrank, nlevs, netcdf_output, bmask ... -- what the reference's drivers take through `use nhydro`
!> (mg_testseamount.f90:37,201,220-221; old_tests/mg_testrelax.f90).  grid(lev)%p becomes grid_get(lev,'p',array).
!>
!> The namelist ./nh_namelist is read by the library (same members, mg_namelist.f90:37-50); rank 0 prints the same parameter
!> block / level table / "ite = " lines.  Multi-rank runs need the three communication hooks of mgx_set_comm (INTEGRATION.md;
!> fortran/mgx_mpi_hooks.cpp serves them with MPI); set `myrank` before nhydro_init.
module nhydro
  use iso_c_binding
  use mgx_c
  use mg_tictoc
  use mg_mpi
  use mg_namelist
  use mg_grids
  use mg_mpi_exchange
  use mg_relax
  use mg_intergrids
  use mg_solvers
  implicit none

contains

  !--------------------------------------------------------------  (nhydro.f90:18-33)
  subroutine nhydro_init(nx, ny, nz, npxg, npyg)
    integer(kind=ip), intent(in) :: nx, ny, nz
    integer(kind=ip), intent(in) :: npxg, npyg
    call mgx_check(mgx_init(nx, ny, nz, npxg, npyg, myrank, c_null_ptr), 'nhydro_init')
    call mgx_namelist_readback()       ! netcdf_output, bmask, ns_pre ... as `use mg_namelist` gives the reference's drivers
    nlevs = mgx_nlevs()                ! mg_grids.f90:117
  end subroutine nhydro_init

  !--------------------------------------------------------------  (nhydro.f90:36-50)
  subroutine nhydro_matrices(dx, dy, zeta, h, rmask, hc, theta_b, theta_s)
    real(kind=rp), dimension(:,:)         , intent(in) :: dx, dy, zeta, h
    real(kind=rp), dimension(:,:), pointer, intent(in) :: rmask
    real(kind=rp)                         , intent(in) :: hc, theta_b, theta_s
    real(kind=rp), dimension(:,:), allocatable :: a, b, c, d   ! contiguous copies of the assumed-shape arguments
    real(kind=rp), dimension(:,:), allocatable, target :: m
    allocate(a, source=dx); allocate(b, source=dy); allocate(c, source=zeta); allocate(d, source=h)
    if (associated(rmask)) then   ! only read by the library when bmask=.true.
       allocate(m, source=rmask)
       call mgx_check(mgx_matrices(a, b, c, d, c_loc(m), hc, theta_b, theta_s), 'nhydro_matrices')
    else
       call mgx_check(mgx_matrices(a, b, c, d, c_null_ptr, hc, theta_b, theta_s), 'nhydro_matrices')
    endif
  end subroutine nhydro_matrices

  !--------------------------------------------------------------  (nhydro.f90:53-102)
  subroutine nhydro_solve(nx, ny, nz, rmaska, ua, va, wa)
    integer(kind=ip), intent(in) :: nx, ny, nz
    real(kind=rp), dimension(0:nx+1,0:ny+1)     , target, intent(inout) :: rmaska
    real(kind=rp), dimension(1:nx+1,0:ny+1,1:nz), target, intent(inout) :: ua
    real(kind=rp), dimension(0:nx+1,1:ny+1,1:nz), target, intent(inout) :: va
    real(kind=rp), dimension(0:nx+1,0:ny+1,0:nz), target, intent(inout) :: wa
    ! rmaska: the memory the caller allocated as rmask(0:ny+1,0:nx+1) and filled as rmask(j,i) (mg_testseamount.f90:97,185).  The
    ! library reads it in THAT layout (element (j,i) at j+(ny+2)*i).  The reference, after `rmask => rmaska` with the bounds declared
    ! above, reads rmask(j,i) at j+(nx+2)*i (nhydro.f90:72, mg_compute_rhs.f90:61,110): the same element for square blocks or an
    ! all-ones mask -- the cases it is run on -- and a different one otherwise.  Deliberate: the drivers' layout is the intent.
    call mgx_check(mgx_solve(ua, va, wa, c_loc(rmaska)), 'nhydro_solve')
  end subroutine nhydro_solve

  !--------------------------------------------------------------  (nhydro.f90:105-134)
  subroutine nhydro_check_nondivergence(nx, ny, nz, rmaska, ua, va, wa)
    integer(kind=ip), intent(in) :: nx, ny, nz
    real(kind=rp), dimension(0:nx+1,0:ny+1)     , target, intent(inout) :: rmaska
    real(kind=rp), dimension(1:nx+1,0:ny+1,1:nz), target, intent(inout) :: ua
    real(kind=rp), dimension(0:nx+1,1:ny+1,1:nz), target, intent(inout) :: va
    real(kind=rp), dimension(0:nx+1,0:ny+1,0:nz), target, intent(inout) :: wa
    call mgx_check(mgx_check_nondivergence(ua, va, wa, c_loc(rmaska)), 'nhydro_check_nondivergence')
  end subroutine nhydro_check_nondivergence

  !--------------------------------------------------------------  (nhydro.f90:137-141)
  subroutine nhydro_clean()
    call mgx_clean()
  end subroutine nhydro_clean

end module nhydro
