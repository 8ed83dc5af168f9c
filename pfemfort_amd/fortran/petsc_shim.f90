! External procedures with the names the drivers call directly (SURVEY 8b, tier 1b): PETSc
! Mat/Vec entry points over the pfem_solver handle, a single-rank MPI (default flavour; with
! -DPFEM_WITH_MPI the real MPI library provides MPI_* and the drivers run under mpiexec), and a
! deterministic stand-in for METIS_PartMeshNodal.
! Implicit-interface externals on purpose: the drivers call them without explicit interfaces.

subroutine PetscInitialize(file, ierr)
  use petscvec
  implicit none
  character(len=*) :: file
  integer :: ierr, io, n
  character(len=256) :: line, key, val
  logical :: ex
#ifdef PFEM_WITH_MPI
  logical :: up
  call MPI_Initialized(up, ierr)
  if (.not. up) call MPI_Init(ierr)
#endif
  ierr = 0
  inquire(file=trim(file), exist=ex)
  if (ex) then
    open(97, file=trim(file), status="old", action="read")
    do
      read(97, '(A)', iostat=io) line
      if (io /= 0) exit
      line = adjustl(line)
      if (len_trim(line) == 0 .or. line(1:1) == '#') cycle
      n = index(trim(line), ' ')
      if (n == 0) then                      ! a flag without a value (-ksp_cg_single_reduction)
        key = trim(line); val = ""
      else
        key = line(1:n-1); val = adjustl(line(n+1:))
      end if
      select case (trim(key))
      case ("-ksp_rtol");   read(val, *, iostat=io) pfem_opt_rtol
      case ("-ksp_atol");   read(val, *, iostat=io) pfem_opt_atol
      case ("-ksp_divtol"); read(val, *, iostat=io) pfem_opt_dtol
      case ("-ksp_max_it"); read(val, *, iostat=io) pfem_opt_maxits
      case ("-pc_type");    call pfem_set_pc_type(trim(val))
      case ("-pc_mg_cycle_type")            ! PCMGSetCycleType behind PCGAMG
        select case (trim(val))
        case ("v", "V"); pfem_opt_cycle = 1
        case ("w", "W"); pfem_opt_cycle = 2
        case default
          write(*,*) "pfem_amd: -pc_mg_cycle_type ", trim(val), " is not available (v, w)"
          error stop " Aborting... unsupported -pc_mg_cycle_type"
        end select
      case ("-ksp_cg_single_reduction")     ! KSPCGUseSingleReduction: one all-reduce per iteration
        pfem_opt_single = 1
        if (trim(val) == "0" .or. trim(val) == "false" .or. trim(val) == "no") pfem_opt_single = 0
      case ("-ksp_type")
        if (trim(val) /= "cg") then        ! the reference hard-wires KSPCG (solverpetsc.F:187); nothing else is built
          write(*,*) "pfem_amd: -ksp_type ", trim(val), " is not available (only cg, as the reference sets)"
          error stop " Aborting... unsupported -ksp_type"
        end if
      case default
        write(*,*) "pfem_amd: option ", trim(key), " in ", trim(file), " is not understood and is IGNORED"
      end select
    end do
    close(97)
  end if
  call get_environment_variable("PFEM_KSP_RTOL", val, status=io)
  if (io == 0 .and. len_trim(val) > 0) read(val, *, iostat=io) pfem_opt_rtol
  call get_environment_variable("PFEM_KSP_MAX_IT", val, status=io)
  if (io == 0 .and. len_trim(val) > 0) read(val, *, iostat=io) pfem_opt_maxits
  call get_environment_variable("PFEM_PC_TYPE", val, status=io)
  if (io == 0 .and. len_trim(val) > 0) call pfem_set_pc_type(trim(val))
end subroutine PetscInitialize

! -pc_type: what this library has.  Anything else stops loudly instead of silently running another solve.
subroutine pfem_set_pc_type(val)
  use petscvec
  implicit none
  character(len=*) :: val
  select case (val)
  case ("jacobi");   pfem_opt_pc = 0
  case ("pbjacobi"); pfem_opt_pc = 1
  case ("gamg");     pfem_opt_pc = 2
  case default
    write(*,*) "pfem_amd: -pc_type ", val, " is not available (jacobi, pbjacobi, gamg)"
    error stop " Aborting... unsupported -pc_type"
  end select
end subroutine pfem_set_pc_type

subroutine PetscFinalize(ierr)
  implicit none
  integer :: ierr
  ierr = 0
#ifdef PFEM_WITH_MPI
  call MPI_Finalize(ierr)
#endif
end subroutine PetscFinalize

! PETSc passes the Fortran string to a C printf: the literal two characters "\n" become a newline
subroutine PetscPrintf(comm, str, ierr)
  implicit none
  integer :: comm, ierr, i, n
  character(len=*) :: str
#ifdef PFEM_WITH_MPI
  integer :: me
  call MPI_Comm_rank(comm, me, ierr)
  ierr = 0
  if (me /= 0) return                 ! PetscPrintf prints on the first rank of the communicator only
#endif
  ierr = 0
  n = len_trim(str)
  i = 1
  do while (i <= n)
    if (i < n .and. str(i:i) == achar(92) .and. str(i+1:i+1) == 'n') then
      write(*, *)
      i = i + 2
    else
      write(*, '(A)', advance='no') str(i:i)
      i = i + 1
    end if
  end do
  if (n == 0 .or. .not. (n >= 2 .and. str(max(n-1,1):n) == achar(92)//'n')) write(*, *)
end subroutine PetscPrintf

subroutine MatSetValues(mat, m, idxm, n, idxn, v, mode, ierr)
  use pfem_amd_c
  implicit none
  integer(kind=8) :: mat
  integer :: m, n, idxm(*), idxn(*), mode, ierr
  double precision :: v(*)
  ierr = pfem_mat_set_values(pfem_h2p(mat), m, idxm, n, idxn, v, mode)
  if (ierr /= 0) call pfem_chkerr(ierr)
end subroutine MatSetValues

subroutine MatSetValue(mat, row, col, v, mode, ierr)
  use pfem_amd_c
  implicit none
  integer(kind=8) :: mat
  integer :: row, col, mode, ierr, r(1), c(1)
  double precision :: v, vv(1)
  r(1) = row; c(1) = col; vv(1) = v
  ierr = pfem_mat_set_values(pfem_h2p(mat), 1, r, 1, c, vv, mode)
  if (ierr /= 0) call pfem_chkerr(ierr)
end subroutine MatSetValue

subroutine VecSetValues(vec, n, idx, v, mode, ierr)
  use pfem_amd_c
  implicit none
  integer(kind=8) :: vec
  integer :: n, idx(*), mode, ierr
  double precision :: v(*)
  ierr = pfem_vec_set_values(pfem_h2p(vec), n, idx, v, mode)
  if (ierr /= 0) call pfem_chkerr(ierr)
end subroutine VecSetValues

subroutine VecSetValue(vec, row, v, mode, ierr)
  use pfem_amd_c
  implicit none
  integer(kind=8) :: vec
  integer :: row, mode, ierr, r(1)
  double precision :: v, vv(1)
  r(1) = row; vv(1) = v
  ierr = pfem_vec_set_values(pfem_h2p(vec), 1, r, vv, mode)
  if (ierr /= 0) call pfem_chkerr(ierr)
end subroutine VecSetValue

! VecScatterCreateToAll + Begin/End: on one rank the gathered vector IS the solution vector
subroutine VecScatterCreateToAll(vec, ctx, vec_seq, ierr)
  implicit none
  integer(kind=8) :: vec, ctx, vec_seq
  integer :: ierr
  ctx = vec; vec_seq = vec; ierr = 0
end subroutine VecScatterCreateToAll

subroutine VecScatterBegin(ctx, vin, vout, mode, dir, ierr)
  implicit none
  integer(kind=8) :: ctx, vin, vout
  integer :: mode, dir, ierr
  ierr = 0
end subroutine VecScatterBegin

subroutine VecScatterEnd(ctx, vin, vout, mode, dir, ierr)
  implicit none
  integer(kind=8) :: ctx, vin, vout
  integer :: mode, dir, ierr
  ierr = 0
end subroutine VecScatterEnd

subroutine VecScatterDestroy(ctx, ierr)
  implicit none
  integer(kind=8) :: ctx
  integer :: ierr
  ctx = 0; ierr = 0
end subroutine VecScatterDestroy

! Legacy VecGetArray(vec, xx_v, xx_i, ierr): the caller indexes xx_v(xx_i + k), k = 1..n
! (tetrapoissonparallelimpl1.F:932-938), so xx_i is the distance in elements from xx_v(1) to
! one-before the first entry of the fetched solution.
subroutine VecGetArray(vec, xx_v, xx_i, ierr)
  use pfem_amd_c
#ifdef PFEM_WITH_MPI
  use petscvec, only: pfem_seq_soln, PETSC_COMM_WORLD, MPI_INTEGER, MPI_SUM
#else
  use petscvec, only: pfem_seq_soln
#endif
  implicit none
  integer(kind=8) :: vec, xx_i
  double precision, target :: xx_v(*)
  integer :: ierr
  integer(c_int64_t) :: nown, nloc, nnz, nst
  integer(kind=8) :: a0, a1
#ifdef PFEM_WITH_MPI
  integer :: nmine, ntot, ierr2
#endif
  interface
    integer(c_int) function pfem_matrix_info(s, a, b, c, d) bind(C, name="pfem_matrix_info")
      import
      type(c_ptr), value :: s
      integer(c_int64_t) :: a, b, c, d
    end function
  end interface
  ierr = pfem_matrix_info(pfem_h2p(vec), nown, nloc, nnz, nst)
  if (ierr /= 0) call pfem_chkerr(ierr)
  if (allocated(pfem_seq_soln)) deallocate(pfem_seq_soln)
#ifdef PFEM_WITH_MPI
  ! vec_SEQ of VecScatterCreateToAll: every rank gets all size_global entries, rank blocks in order
  nmine = int(nown)
  call MPI_Allreduce(nmine, ntot, 1, MPI_INTEGER, MPI_SUM, PETSC_COMM_WORLD, ierr2)
  allocate(pfem_seq_soln(max(ntot, 1)))
  ierr = pfem_mpi_gather_solution(pfem_h2p(vec), PETSC_COMM_WORLD, nown, pfem_seq_soln)
#else
  allocate(pfem_seq_soln(max(nown, 1_c_int64_t)))
  ierr = pfem_solver_get_solution(pfem_h2p(vec), pfem_seq_soln)
#endif
  if (ierr /= 0) call pfem_chkerr(ierr)
  a0 = transfer(c_loc(xx_v(1)), a0)
  a1 = transfer(c_loc(pfem_seq_soln(1)), a1)
  xx_i = (a1 - a0) / 8 - 0      ! xx_v(xx_i + 1) == pfem_seq_soln(1)
end subroutine VecGetArray

subroutine VecRestoreArray(vec, xx_v, xx_i, ierr)
  implicit none
  integer(kind=8) :: vec, xx_i
  double precision :: xx_v(*)
  integer :: ierr
  ierr = 0
end subroutine VecRestoreArray

! ---- MPI on one rank -------------------------------------------------------------------------
#ifndef PFEM_WITH_MPI
double precision function MPI_Wtime()
  implicit none
  integer(kind=8) :: c, r
  call system_clock(c, r)
  MPI_Wtime = dble(c) / dble(r)
end function MPI_Wtime

subroutine MPI_Comm_size(comm, n, ierr)
  implicit none
  integer :: comm, n, ierr
  n = 1; ierr = 0
end subroutine MPI_Comm_size

subroutine MPI_Comm_rank(comm, r, ierr)
  implicit none
  integer :: comm, r, ierr
  r = 0; ierr = 0
end subroutine MPI_Comm_rank

subroutine MPI_Barrier(comm, ierr)
  implicit none
  integer :: comm, ierr
  ierr = 0
end subroutine MPI_Barrier

subroutine MPI_Bcast(buf, n, dtype, root, comm, ierr)
  implicit none
  integer :: buf(*), n, dtype, root, comm, ierr
  ierr = 0
end subroutine MPI_Bcast

subroutine MPI_Allgather(sbuf, ns, stype, rbuf, nr, rtype, comm, ierr)
  implicit none
  integer :: sbuf(*), ns, stype, rbuf(*), nr, rtype, comm, ierr
  rbuf(1:ns) = sbuf(1:ns); ierr = 0
end subroutine MPI_Allgather

subroutine MPI_Allgatherv(sbuf, ns, stype, rbuf, nrs, displs, rtype, comm, ierr)
  implicit none
  integer :: sbuf(*), ns, stype, rbuf(*), nrs(*), displs(*), rtype, comm, ierr
  rbuf(displs(1)+1:displs(1)+ns) = sbuf(1:ns); ierr = 0
end subroutine MPI_Allgatherv

subroutine MPI_Allreduce(sbuf, rbuf, n, dtype, op, comm, ierr)
  implicit none
  integer :: sbuf(*), rbuf(*), n, dtype, op, comm, ierr
  rbuf(1:n) = sbuf(1:n); ierr = 0
end subroutine MPI_Allreduce

#endif

! ---- METIS: only reached when n_mpi_procs > 1 (tetrapoissonparallelimpl1.F:423-467) ----------------
! METIS 5 is a third-party library that is neither part of the reference tree nor installed here.
! Stand-in with the same interface and the same kind of result (0-based part of every element and
! node): nodes are cut into nparts contiguous index blocks of equal size -- z-slabs for the
! structured generator's numbering -- and an element goes to the lowest part among its nodes.
! Any valid partition works downstream: the drivers renumber from (epart, npart) alone.
! A partition computed offline by the real METIS is taken instead when PFEM_METIS_PREFIX is set:
! mpmetis writes <prefix>.epart.<nparts> and <prefix>.npart.<nparts>, one 0-based part per line.
subroutine METIS_SetDefaultOptions(options)
  implicit none
  integer :: options(*)
  options(1) = 0
end subroutine METIS_SetDefaultOptions

subroutine METIS_PartMeshNodal(ne, nn, eptr, eind, vwgt, vsize, nparts, tpwgts, options, objval, epart, npart)
  implicit none
  integer :: ne, nn, eptr(*), eind(*), vwgt, vsize, nparts, options(*), objval, epart(*), npart(*)
  double precision :: tpwgts
  integer :: e, k, p, io
  integer(kind=8) :: i
  character(len=1024) :: prefix
  character(len=16) :: sfx
  call get_environment_variable("PFEM_METIS_PREFIX", prefix, status=io)
  if (io == 0 .and. len_trim(prefix) > 0) then
    write(sfx, '(I0)') nparts
    open(98, file=trim(prefix)//'.epart.'//trim(sfx), status='old', action='read', iostat=io)
    if (io /= 0) stop "PFEM_METIS_PREFIX is set but <prefix>.epart.<nparts> cannot be opened"
    read(98, *, iostat=io) (epart(e), e = 1, ne)
    if (io /= 0) stop "short or malformed .epart file"
    close(98)
    open(98, file=trim(prefix)//'.npart.'//trim(sfx), status='old', action='read', iostat=io)
    if (io /= 0) stop "PFEM_METIS_PREFIX is set but <prefix>.npart.<nparts> cannot be opened"
    read(98, *, iostat=io) (npart(i), i = 1, nn)
    if (io /= 0) stop "short or malformed .npart file"
    close(98)
    if (minval(epart(1:ne)) < 0 .or. maxval(epart(1:ne)) >= nparts .or. &
        minval(npart(1:nn)) < 0 .or. maxval(npart(1:nn)) >= nparts) stop "part id out of range in the METIS files"
    objval = 0
    return
  end if
  do i = 1, nn
    npart(i) = int(((i - 1) * int(nparts, 8)) / int(nn, 8))
  end do
  do e = 1, ne
    p = nparts
    do k = eptr(e) + 1, eptr(e + 1)          ! eptr / eind are 0-based (C numbering)
      p = min(p, npart(eind(k) + 1))
    end do
    epart(e) = p
  end do
  objval = 0
end subroutine METIS_PartMeshNodal

! (iargc/getarg, GNU extensions used by the drivers, come from the flang runtime; petscvec only
!  declares iargc's type because the drivers are IMPLICIT NONE)
