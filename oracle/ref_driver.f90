! Own driver (not reference code) over the three reference modules that compile unmodified, in place, with flang and
! need neither MPI nor NetCDF: mg_zr_zw (SURVEY 8 row a13), mg_namelist (the &nhparam surface of row b), mg_tictoc.
! Test infrastructure: built by `make -C oracle ref` into oracle/_ref/ (git-ignored), used by oracle/make_ref_golden.py to
! produce the fixtures under tests/golden/ref_*.  Three modes, chosen by the first command-line word:
!
!   zrzw <in.bin> <out.bin>   in : int32 nx,ny,nz ; real64 hlim,theta_b,theta_s ; h(0:ny+1,0:nx+1), zeta(0:ny+1,0:nx+1)
!                             out: zr(nz,-1:ny+2,-1:nx+2), zw(nz+1,-1:ny+2,-1:nx+2) as define_matrices allocates them
!                             (mg_grids.f90:216-217; setup_zr_zw fills the 0:n+1 ring, the rest stays at the -999 put there here)
!   namelist <file>           read_nhnamelist(file), then one "key = value" line per member of /nhparam/ and "end_of_members"
!   tictoc <out>              a fixed tic/toc sequence, print_tictoc into unit 10 = file <out>
program ref_driver
  use mg_zr_zw
  use mg_namelist
  implicit none
  character(len=256) :: mode, a1, a2
  call get_command_argument(1, mode)
  call get_command_argument(2, a1)
  call get_command_argument(3, a2)
  select case (trim(mode))
  case ('zrzw')
     call do_zrzw(trim(a1), trim(a2))
  case ('namelist')
     call do_namelist(trim(a1))
  case ('tictoc')
     call do_tictoc(trim(a1))
  case default
     write(*,*) 'usage: ref_driver zrzw in out | namelist file | tictoc out'
     stop 2
  end select
contains

  subroutine do_zrzw(fin, fout)
    character(len=*), intent(in) :: fin, fout
    integer(kind=4) :: nx, ny, nz
    real(kind=8) :: hlim, theta_b, theta_s
    real(kind=8), dimension(:,:), pointer :: h, zeta
    real(kind=8), dimension(:,:,:), pointer :: zr, zw
    open(unit=21, file=fin, access='stream', form='unformatted', action='read')
    read(21) nx, ny, nz
    read(21) hlim, theta_b, theta_s
    allocate(h(0:ny+1,0:nx+1), zeta(0:ny+1,0:nx+1))
    read(21) h
    read(21) zeta
    close(21)
    allocate(zr(nz,-1:ny+2,-1:nx+2), zw(nz+1,-1:ny+2,-1:nx+2))
    zr = -999._8
    zw = -999._8
    call setup_zr_zw(hlim, theta_b, theta_s, zeta, h, zr, zw, coord_type='new_s_coord')   ! the call of mg_define_matrix.f90:116-127
    open(unit=22, file=fout, access='stream', form='unformatted', action='write', status='replace')
    write(22) zr
    write(22) zw
    close(22)
  end subroutine do_zrzw

  subroutine do_namelist(fn)
    character(len=*), intent(in) :: fn
    call read_nhnamelist(filename=fn, verbose=.false.)
    write(*,'(A,ES24.16E3)') 'solver_prec = ', solver_prec
    write(*,'(A,I0)') 'solver_maxiter = ', solver_maxiter
    write(*,'(A,I0)') 'nsmall = ', nsmall
    write(*,'(A,I0)') 'ns_coarsest = ', ns_coarsest
    write(*,'(A,I0)') 'ns_pre = ', ns_pre
    write(*,'(A,I0)') 'ns_post = ', ns_post
    write(*,'(A,A)') 'cmatrix = ', trim(cmatrix)
    write(*,'(A,A)') 'relax_method = ', trim(relax_method)
    write(*,'(A,A)') 'interp_type = ', trim(interp_type)
    write(*,'(A,A)') 'restrict_type = ', trim(restrict_type)
    write(*,'(A,L1)') 'aggressive = ', aggressive
    write(*,'(A,L1)') 'netcdf_output = ', netcdf_output
    write(*,'(A,L1)') 'bmask = ', bmask
    write(*,'(A)') 'end_of_members'
  end subroutine do_namelist

  subroutine do_tictoc(fout)
    character(len=*), intent(in) :: fout
    integer :: it
    open(unit=10, file=fout, action='write', status='replace')
    do it = 1, 3
       call tic(1, 'solve')
       call tic(1, 'relax_3D_8_FC')
       call toc(1, 'relax_3D_8_FC')
       call tic(2, 'relax_3D_8_FC')
       call toc(2, 'relax_3D_8_FC')
       call tic(3, 'residual_3D_8')
       call toc(3, 'residual_3D_8')
       call toc(1, 'solve')
    enddo
    call print_tictoc()
    close(10)
  end subroutine do_tictoc

end program ref_driver
