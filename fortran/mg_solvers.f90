!> The solver-level half of the reference's Fortran surface, over ISO_C_BINDING to libmgx.so (include/mgx.h): the procedures the
!> reference's drivers and unit programs reach THROUGH `use nhydro` (its `use` chain re-exports every module below it), under the
!> reference's own module and procedure names and argument lists:
!>   mg_tictoc       tic, toc, print_tictoc                         (src/mg_tictoc.f90:21,73,114)
!>   mg_mpi          myrank                                         (src/mg_mpi.f90:7)
!>   mg_namelist     netcdf_output, bmask and the other /nhparam/ members, valid after nhydro_init   (src/mg_namelist.f90:12-50)
!>   mg_grids        nlevs; grid_get / grid_set / grid_dims in place of the pointer components grid(lev)%p, %b, %r, %nx ... (src/mg_grids.f90:24-117)
!>   mg_mpi_exchange fill_halo(lev, 'p')                            (src/mg_mpi_exchange.f90:10-16)
!>   mg_relax        relax(lev,nsweeps), compute_residual(lev,res)  (src/mg_relax.f90:16,337)
!>   mg_intergrids   fine2coarse(lev), coarse2fine(lev)             (src/mg_intergrids.f90:16,167)
!>   mg_solvers      solve_p(tol,maxite), Fcycle(), Vcycle(lev), Vcycle2(lev1,lev2), testgalerkin(lev)   (src/mg_solvers.f90:17,104,129,155,203)
!> One difference is unavoidable: the level arrays live on the GPU, so `grid(lev)%p` cannot be a Fortran pointer.  grid_get(lev,'p',a)
!> copies the field into the caller's array a(nz,0:ny+1,0:nx+1) (the reference's shape), grid_set the other way.  A line such as
!>   call write_netcdf(grid(1)%b, ...)        becomes        call grid_get(1,'b',b) ; call write_netcdf(b, ...)
!> Errors: the reference stops (`stop -1`); so do these (mgx_check) after printing libmgx's message.
module mgx_c
  use iso_c_binding
  implicit none
  integer(kind=4), parameter :: rp = 8, ip = 4, st = 4, lg = 8

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
     integer(c_int) function mgx_set_field(lev, field, host) bind(C, name='mgx_set_field')
       import :: c_int, c_double
       integer(c_int), value :: lev, field
       real(c_double), intent(in) :: host(*)
     end function mgx_set_field
     integer(c_int) function mgx_get_option(name, value) bind(C, name='mgx_get_option')
       import :: c_int, c_char
       character(kind=c_char), intent(in) :: name(*)
       integer(c_int), intent(out) :: value
     end function mgx_get_option
     integer(c_int) function mgx_set_option(name, value) bind(C, name='mgx_set_option')
       import :: c_int, c_char
       character(kind=c_char), intent(in) :: name(*)
       integer(c_int), value :: value
     end function mgx_set_option
     integer(c_int) function mgx_level_info(lev, info) bind(C, name='mgx_level_info')
       import :: c_int
       integer(c_int), value :: lev
       integer(c_int), intent(out) :: info(18)
     end function mgx_level_info
     integer(c_int) function mgx_level_dims(lev, nx, ny, nz) bind(C, name='mgx_level_dims')
       import :: c_int
       integer(c_int), value :: lev
       integer(c_int), intent(out) :: nx, ny, nz
     end function mgx_level_dims
     integer(c_int) function mgx_nlevs() bind(C, name='mgx_nlevs')
       import :: c_int
     end function mgx_nlevs
     integer(c_int) function mgx_solve_p(tol, maxite, nite, res, hist) bind(C, name='mgx_solve_p')
       import :: c_int, c_double, c_ptr
       real(c_double), value :: tol
       integer(c_int), value :: maxite
       integer(c_int), intent(out) :: nite
       real(c_double), intent(out) :: res
       type(c_ptr), value :: hist
     end function mgx_solve_p
     integer(c_int) function mgx_fcycle() bind(C, name='mgx_fcycle')
       import :: c_int
     end function mgx_fcycle
     integer(c_int) function mgx_vcycle(lev) bind(C, name='mgx_vcycle')
       import :: c_int
       integer(c_int), value :: lev
     end function mgx_vcycle
     integer(c_int) function mgx_vcycle2(lev1, lev2) bind(C, name='mgx_vcycle2')
       import :: c_int
       integer(c_int), value :: lev1, lev2
     end function mgx_vcycle2
     integer(c_int) function mgx_testgalerkin(lev, norm_c, norm_f) bind(C, name='mgx_testgalerkin')
       import :: c_int, c_double
       integer(c_int), value :: lev
       real(c_double), intent(out) :: norm_c, norm_f
     end function mgx_testgalerkin
     integer(c_int) function mgx_relax(lev, nsweeps) bind(C, name='mgx_relax')
       import :: c_int
       integer(c_int), value :: lev, nsweeps
     end function mgx_relax
     integer(c_int) function mgx_residual(lev, res) bind(C, name='mgx_residual')
       import :: c_int, c_double
       integer(c_int), value :: lev
       real(c_double), intent(out) :: res
     end function mgx_residual
     integer(c_int) function mgx_fine2coarse(lev) bind(C, name='mgx_fine2coarse')
       import :: c_int
       integer(c_int), value :: lev
     end function mgx_fine2coarse
     integer(c_int) function mgx_coarse2fine(lev) bind(C, name='mgx_coarse2fine')
       import :: c_int
       integer(c_int), value :: lev
     end function mgx_coarse2fine
     integer(c_int) function mgx_fill_halo(lev, field) bind(C, name='mgx_fill_halo')
       import :: c_int
       integer(c_int), value :: lev, field
     end function mgx_fill_halo
     integer(c_int) function mgx_tic(lev, name) bind(C, name='mgx_tic')
       import :: c_int, c_char
       integer(c_int), value :: lev
       character(kind=c_char), intent(in) :: name(*)
     end function mgx_tic
     integer(c_int) function mgx_toc(lev, name) bind(C, name='mgx_toc')
       import :: c_int, c_char
       integer(c_int), value :: lev
       character(kind=c_char), intent(in) :: name(*)
     end function mgx_toc
     integer(c_int) function mgx_print_tictoc(path) bind(C, name='mgx_print_tictoc')
       import :: c_int, c_char
       character(kind=c_char), intent(in) :: path(*)
     end function mgx_print_tictoc
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

  !> field ids of include/mgx.h (MGX_P ...) for the component names of grid_type (mg_grids.f90:24-65)
  integer(c_int) function mgx_field_id(name) result(id)
    character(len=*), intent(in) :: name
    select case (trim(name))
    case ('p');     id = 0
    case ('b');     id = 1
    case ('r');     id = 2
    case ('cA');    id = 3
    case ('dx');    id = 4
    case ('dy');    id = 5
    case ('zeta');  id = 6
    case ('h');     id = 7
    case ('zr');    id = 8
    case ('zw');    id = 9
    case ('cw');    id = 10
    case ('rmask'); id = 14
    case default
       write(*,*) 'Error: grid_type has no array component named ', trim(name)
       stop -1
    end select
  end function mgx_field_id

end module mgx_c

!--------------------------------------------------------------------------------------------------  (src/mg_tictoc.f90)
module mg_tictoc
  use mgx_c
  implicit none
contains
  subroutine tic(lev, string)
    integer(kind=st), intent(in) :: lev
    character(len=*), intent(in) :: string
    call mgx_check(mgx_tic(lev, trim(string)//c_null_char), 'tic')
  end subroutine tic
  subroutine toc(lev, string)
    integer(kind=st), intent(in) :: lev
    character(len=*), intent(in) :: string
    if (mgx_toc(lev, trim(string)//c_null_char) /= 0) write(*,*) 'Error: tictoc'   ! mg_tictoc.f90:104-108 goes on as well
  end subroutine toc
  !> the table goes to Fortran's default file of unit myrank + 10, as the reference's write(lun, ...) without an open does (fort.10)
  subroutine print_tictoc(myrank)
    integer(kind=st), optional, intent(in) :: myrank
    character(len=16) :: fname
    integer(kind=st) :: lun
    lun = 10
    if (present(myrank)) lun = myrank + 10
    write(fname, '("fort.",I0)') lun
    call mgx_check(mgx_print_tictoc(trim(fname)//c_null_char), 'print_tictoc')
  end subroutine print_tictoc
end module mg_tictoc

!--------------------------------------------------------------------------------------------------  (src/mg_mpi.f90)
module mg_mpi
  use mgx_c
  implicit none
  integer(kind=4) :: myrank = 0    !< the caller's rank in its process grid; set it before nhydro_init (the reference: mpi_comm_rank in mg_mpi_init)
end module mg_mpi

!--------------------------------------------------------------------------------------------------  (src/mg_namelist.f90:12-50)
module mg_namelist
  use mgx_c
  implicit none
  !> members of /nhparam/ the drivers read as module variables; the library reads ./nh_namelist in nhydro_init, these are copies made then
  integer(kind=ip) :: solver_maxiter = 50, nsmall = 8, ns_coarsest = 40, ns_pre = 3, ns_post = 2
  logical :: aggressive = .false., netcdf_output = .false., bmask = .false.
contains
  subroutine mgx_namelist_readback()
    integer(c_int) :: v
    call mgx_check(mgx_get_option('solver_maxiter'//c_null_char, v), 'read_nhnamelist'); solver_maxiter = v
    call mgx_check(mgx_get_option('nsmall'//c_null_char, v), 'read_nhnamelist'); nsmall = v
    call mgx_check(mgx_get_option('ns_coarsest'//c_null_char, v), 'read_nhnamelist'); ns_coarsest = v
    call mgx_check(mgx_get_option('ns_pre'//c_null_char, v), 'read_nhnamelist'); ns_pre = v
    call mgx_check(mgx_get_option('ns_post'//c_null_char, v), 'read_nhnamelist'); ns_post = v
    call mgx_check(mgx_get_option('aggressive'//c_null_char, v), 'read_nhnamelist'); aggressive = v /= 0
    call mgx_check(mgx_get_option('netcdf_output'//c_null_char, v), 'read_nhnamelist'); netcdf_output = v /= 0
    call mgx_check(mgx_get_option('bmask'//c_null_char, v), 'read_nhnamelist'); bmask = v /= 0
  end subroutine mgx_namelist_readback
end module mg_namelist

!--------------------------------------------------------------------------------------------------  (src/mg_grids.f90)
module mg_grids
  use mgx_c
  implicit none
  integer(kind=ip) :: nlevs = 0    !< index of the coarsest level (mg_grids.f90:117); valid after nhydro_init
  interface grid_get
     module procedure grid_get_2D, grid_get_3D, grid_get_4D
  end interface grid_get
  interface grid_set
     module procedure grid_set_2D, grid_set_3D, grid_set_4D
  end interface grid_set
contains
  !> grid(lev)%nx, %ny, %nz
  subroutine grid_dims(lev, nx, ny, nz)
    integer(kind=ip), intent(in) :: lev
    integer(kind=ip), intent(out) :: nx, ny, nz
    call mgx_check(mgx_level_dims(lev, nx, ny, nz), 'grid_dims')
  end subroutine grid_dims
  !> a = grid(lev)%<name>: p, b, r (nz,0:ny+1,0:nx+1); zr (nz,-1:ny+2,-1:nx+2); zw (nz+1,-1:ny+2,-1:nx+2); cw (nz+1,0:ny+1,0:nx+1)
  subroutine grid_get_3D(lev, name, a)
    integer(kind=ip), intent(in) :: lev
    character(len=*), intent(in) :: name
    real(kind=rp), dimension(:,:,:), contiguous, intent(out) :: a
    call mgx_check(mgx_get_field(lev, mgx_field_id(name), a), 'grid_get')
  end subroutine grid_get_3D
  !> dx, dy, zeta, h, rmask (0:ny+1,0:nx+1)
  subroutine grid_get_2D(lev, name, a)
    integer(kind=ip), intent(in) :: lev
    character(len=*), intent(in) :: name
    real(kind=rp), dimension(:,:), contiguous, intent(out) :: a
    call mgx_check(mgx_get_field(lev, mgx_field_id(name), a), 'grid_get')
  end subroutine grid_get_2D
  !> cA (8,nz,0:ny+1,0:nx+1)
  subroutine grid_get_4D(lev, name, a)
    integer(kind=ip), intent(in) :: lev
    character(len=*), intent(in) :: name
    real(kind=rp), dimension(:,:,:,:), contiguous, intent(out) :: a
    call mgx_check(mgx_get_field(lev, mgx_field_id(name), a), 'grid_get')
  end subroutine grid_get_4D
  subroutine grid_set_3D(lev, name, a)
    integer(kind=ip), intent(in) :: lev
    character(len=*), intent(in) :: name
    real(kind=rp), dimension(:,:,:), contiguous, intent(in) :: a
    call mgx_check(mgx_set_field(lev, mgx_field_id(name), a), 'grid_set')
  end subroutine grid_set_3D
  subroutine grid_set_2D(lev, name, a)
    integer(kind=ip), intent(in) :: lev
    character(len=*), intent(in) :: name
    real(kind=rp), dimension(:,:), contiguous, intent(in) :: a
    call mgx_check(mgx_set_field(lev, mgx_field_id(name), a), 'grid_set')
  end subroutine grid_set_2D
  subroutine grid_set_4D(lev, name, a)
    integer(kind=ip), intent(in) :: lev
    character(len=*), intent(in) :: name
    real(kind=rp), dimension(:,:,:,:), contiguous, intent(in) :: a
    call mgx_check(mgx_set_field(lev, mgx_field_id(name), a), 'grid_set')
  end subroutine grid_set_4D
end module mg_grids

!--------------------------------------------------------------------------------------------------  (src/mg_mpi_exchange.f90)
module mg_mpi_exchange
  use mgx_c
  implicit none
contains
  !> fill_halo(lev, grid(lev)%<name>) (the generic of mg_mpi_exchange.f90:10-16): the array is named, not passed -- it lives on the GPU
  subroutine fill_halo(lev, name)
    integer(kind=ip), intent(in) :: lev
    character(len=*), intent(in) :: name
    call mgx_check(mgx_fill_halo(lev, mgx_field_id(name)), 'fill_halo')
  end subroutine fill_halo
  !> fill_halo_2D_bmask(lev, a2D) (mg_mpi_exchange.f90:357-391): zero the halo line of every side without a neighbour -- on the CALLER's
  !> array (what the reference's drivers do to rmask before nhydro_matrices when bmask, mg_testseamount.f90)
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
end module mg_mpi_exchange

!--------------------------------------------------------------------------------------------------  (src/mg_relax.f90)
module mg_relax
  use mgx_c
  implicit none
contains
  subroutine relax(lev, nsweeps)                    ! mg_relax.f90:16
    integer(kind=ip), intent(in) :: lev
    integer(kind=ip), intent(in) :: nsweeps
    call mgx_check(mgx_relax(lev, nsweeps), 'relax')
  end subroutine relax
  subroutine compute_residual(lev, res)             ! mg_relax.f90:337 (res = sqrt of the global sum of r**2, :377-379)
    integer(kind=ip), intent(in) :: lev
    real(kind=rp)   , intent(out):: res
    call mgx_check(mgx_residual(lev, res), 'compute_residual')
  end subroutine compute_residual
end module mg_relax

!--------------------------------------------------------------------------------------------------  (src/mg_intergrids.f90)
module mg_intergrids
  use mgx_c
  implicit none
contains
  subroutine fine2coarse(lev)                       ! mg_intergrids.f90:16
    integer(kind=ip), intent(in) :: lev
    call mgx_check(mgx_fine2coarse(lev), 'fine2coarse')
  end subroutine fine2coarse
  subroutine coarse2fine(lev)                       ! mg_intergrids.f90:167
    integer(kind=ip), intent(in) :: lev
    call mgx_check(mgx_coarse2fine(lev), 'coarse2fine')
  end subroutine coarse2fine
end module mg_intergrids

!--------------------------------------------------------------------------------------------------  (src/mg_solvers.f90)
module mg_solvers
  use mgx_c
  implicit none
  integer(kind=ip) :: solve_p_nite = 0       !< iterations of the last solve_p (the reference only prints them, mg_solvers.f90:99)
  real(kind=rp)    :: solve_p_res = 0._rp    !< its last ||r|| / ||b||
contains
  subroutine solve_p(tol, maxite)                   ! mg_solvers.f90:17
    real(kind=rp)   , intent(in) :: tol
    integer(kind=ip), intent(in) :: maxite
    call mgx_check(mgx_solve_p(tol, maxite, solve_p_nite, solve_p_res, c_null_ptr), 'solve_p')
  end subroutine solve_p
  subroutine Fcycle()                               ! mg_solvers.f90:104
    call mgx_check(mgx_fcycle(), 'Fcycle')
  end subroutine Fcycle
  subroutine Vcycle(lev1)                           ! mg_solvers.f90:129
    integer(kind=ip), intent(in) :: lev1
    call mgx_check(mgx_vcycle(lev1), 'Vcycle')
  end subroutine Vcycle
  subroutine Vcycle2(lev1, lev2)                    ! mg_solvers.f90:155
    integer(kind=ip), intent(in) :: lev1, lev2
    call mgx_check(mgx_vcycle2(lev1, lev2), 'Vcycle2')
  end subroutine Vcycle2
  !> mg_solvers.f90:203.  The reference fills grid(lev)%p with random_number itself; here the caller sets it first (grid_set(lev,'p',..)),
  !> everything after that is the reference's sequence and the library prints the reference's three lines.
  subroutine testgalerkin(lev)
    integer(kind=ip) :: lev
    real(kind=rp) :: norm_c, norm_f
    call mgx_check(mgx_testgalerkin(lev, norm_c, norm_f), 'testgalerkin')
  end subroutine testgalerkin
end module mg_solvers
