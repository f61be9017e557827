!> Drop-in replacement of the reference's model-facing module (src/nhydro.f90): same module name, same five
!> procedures with the same argument lists, bodies forwarding through ISO_C_BINDING to libmgx.so (include/mgx.h),
!> whose operators are HIP kernels on the MI355X.  An ocean model (or the reference's own mg_testseamount driver)
!> that does `use nhydro` links against this object + libmgx.so instead of the reference's solver modules.
!>
!> Differences a caller can observe: none in the interface.  The namelist ./nh_namelist is read by the library
!> (same members, mg_namelist.f90:37-50); rank 0 prints the same parameter block / level table / "ite = " lines.
!> Multi-rank runs need the three communication hooks of mgx_set_comm (INTEGRATION.md); this shim covers the
!> single-process case, nhydro_init(nx,ny,nz,1,1).
module nhydro
  use iso_c_binding
  implicit none
  integer(kind=4), parameter :: rp = 8, ip = 4
  integer(kind=4) :: nhydro_rank = 0   !< set before nhydro_init when the caller has an MPI rank
  logical :: bmask = .false.           !< namelist member, as `use mg_namelist` gives the reference's drivers; valid after nhydro_init

  interface
     integer(c_int) function mgx_init(nx, ny, nz, npx, npy, rank, par) bind(C, name='mgx_init')
       import :: c_int, c_ptr
       integer(c_int), value :: nx, ny, nz, npx, npy, rank
       type(c_ptr), value :: par
     end function mgx_init
     integer(c_int) function mgx_matrices(dx, dy, zeta, h, rmask, hc, theta_b, theta_s) bind(C, name='mgx_matrices')
       import :: c_int, c_double, c_ptr
       real(c_double), intent(in) :: dx(*), dy(*), zeta(*), h(*)
       type(c_ptr), value :: rmask
       real(c_double), value :: hc, theta_b, theta_s
     end function mgx_matrices
     integer(c_int) function mgx_solve(u, v, w, rmask) bind(C, name='mgx_solve')
       import :: c_int, c_double, c_ptr
       real(c_double), intent(inout) :: u(*), v(*), w(*)
       type(c_ptr), value :: rmask
     end function mgx_solve
     integer(c_int) function mgx_check_nondivergence(u, v, w, rmask) bind(C, name='mgx_check_nondivergence')
       import :: c_int, c_double, c_ptr
       real(c_double), intent(inout) :: u(*), v(*), w(*)
       type(c_ptr), value :: rmask
     end function mgx_check_nondivergence
     subroutine mgx_clean() bind(C, name='mgx_clean')
     end subroutine mgx_clean
     integer(c_int) function mgx_get_field(lev, field, host) bind(C, name='mgx_get_field')
       import :: c_int, c_double
       integer(c_int), value :: lev, field
       real(c_double), intent(out) :: host(*)
     end function mgx_get_field
     integer(c_int) function mgx_get_option(name, value) bind(C, name='mgx_get_option')
       import :: c_int, c_char
       character(kind=c_char), intent(in) :: name(*)
       integer(c_int), intent(out) :: value
     end function mgx_get_option
     integer(c_int) function mgx_level_info(lev, info) bind(C, name='mgx_level_info')
       import :: c_int
       integer(c_int), value :: lev
       integer(c_int), intent(out) :: info(18)
     end function mgx_level_info
     type(c_ptr) function mgx_last_error() bind(C, name='mgx_last_error')
       import :: c_ptr
     end function mgx_last_error
  end interface

contains

  subroutine mgx_check(rc, where)
    integer(c_int), intent(in) :: rc
    character(len=*), intent(in) :: where
    if (rc /= 0) then
       write(*,*) 'Error in ', where, ' (libmgx), see stderr'
       stop -1   ! the reference's error behaviour (mg_grids.f90:530,657)
    endif
  end subroutine mgx_check

  !--------------------------------------------------------------  (nhydro.f90:18-33)
  subroutine nhydro_init(nx, ny, nz, npxg, npyg)
    integer(kind=ip), intent(in) :: nx, ny, nz
    integer(kind=ip), intent(in) :: npxg, npyg
    integer(c_int) :: ib
    call mgx_check(mgx_init(nx, ny, nz, npxg, npyg, nhydro_rank, c_null_ptr), 'nhydro_init')
    call mgx_check(mgx_get_option('bmask'//c_null_char, ib), 'nhydro_init')
    bmask = ib /= 0
  end subroutine nhydro_init

  !--------------------------------------------------------------  (mg_mpi_exchange.f90:357-391)
  !> fill_halo_2D_bmask(1, a2D): zero the halo line of every side without a neighbour (what the reference's drivers call on rmask
  !> before nhydro_matrices when bmask, mg_testseamount.f90)
  subroutine fill_halo_2D_bmask(lev, a2D)
    integer(kind=ip), intent(in) :: lev
    real(kind=rp), dimension(:,:), pointer, intent(inout) :: a2D
    integer(c_int) :: info(18)
    integer(kind=ip) :: nx, ny, j0, i0
    call mgx_check(mgx_level_info(lev, info), 'fill_halo_2D_bmask')
    j0 = lbound(a2D, 1); i0 = lbound(a2D, 2)
    ny = size(a2D, dim=1) - 2; nx = size(a2D, dim=2) - 2
    if (info(11) < 0) a2D(j0, :) = 0._rp            ! south
    if (info(12) < 0) a2D(:, i0+nx+1) = 0._rp       ! east
    if (info(13) < 0) a2D(j0+ny+1, :) = 0._rp       ! north
    if (info(14) < 0) a2D(:, i0) = 0._rp            ! west
  end subroutine fill_halo_2D_bmask

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
