! boundary_check.F90 -- the build's OWN small Fortran host program for the drop-in boundary (test infrastructure).
!
! It talks to the library exactly the way a PFEMFort driver does -- the PETSc finclude files and modules, TYPE
! PetscSolver of Module_SolverPetsc (initialise / setZero / factoriseAndSolve / free), the element modules
! ElementUtilitiesPoisson / ElementUtilitiesElasticity3D, MatSetValues / VecSetValues with INSERT_VALUES and
! ADD_VALUES, VecScatterCreateToAll + the legacy VecGetArray(xx_v, xx_i) -- but carries none of a driver's mesh
! bookkeeping: the prepared problem (coordinates and connectivity in the new numbering, ElemDofArray, element
! ownership, row blocks) is read from "problem.txt", written by tests/test_fortran_boundary.py.
! PFEM_CHECK_MODE=device: the element loop runs on the GPU instead (PetscSolver%uploadMeshToDevice / %assembleOnDevice, the
! build's extension of the type: one-thread-per-... kernels of the batched path driven from Fortran).
! Serial flavour: one process.  MPI flavour (-DPFEM_WITH_MPI, mpiexec -n P): every rank loops over the elements it
! owns and passes GLOBAL indices, as tetrapoissonparallelimpl1.F:828-884 does.  Rank 0 writes "solution.txt".
#include <petsc/finclude/petscsysdef.h>
#include <petsc/finclude/petscvecdef.h>
#include <petsc/finclude/petscmatdef.h>
#include <petsc/finclude/petsckspdef.h>
#include <petsc/finclude/petscpcdef.h>
program boundary_check
  use petscvec
  use petscmat
  use petscksp
  use petscpc
  use Module_SolverPetsc
  use ElementUtilitiesPoisson
  use ElementUtilitiesElasticity3D
  implicit none
  type(PetscSolver) :: solver
  PetscErrorCode :: ierr
  PetscInt :: me, nprocs
  Vec :: vec_seq
  VecScatter :: ctx
  PetscScalar :: xx_v(1)
  PetscOffset :: xx_i
  integer :: ndof, nNode, nElem, ntot, nranks_file, nsize, e, a, d, i, j, k, node, nloc
  integer, allocatable :: sizes(:), owner(:), conn(:,:), edof(:,:), rows(:), nnz_d(:), nnz_o(:), conn_loc(:,:), edof_loc(:,:)
  double precision, allocatable :: xyz(:,:), applied(:), Kl(:,:), Fl(:), zeroK(:,:), valC(:), valDotC(:), coords(:,:)
  character(len=32) :: mode
  logical :: have_bin
  integer(kind=8) :: tk0, tk1, tk2, tk3, tk4, tkr
  double precision :: xn(4), yn(4), zn(4), elemData(6), timeData(3), fact

  call PetscInitialize("petsc_options.dat", ierr)
  call MPI_Comm_rank(PETSC_COMM_WORLD, me, ierr)
  call MPI_Comm_size(PETSC_COMM_WORLD, nprocs, ierr)

  ! "problem.bin" (the same records as unformatted stream: bench.py --mode compat at config 2's size, where text would
  ! take minutes to write and read) or "problem.txt"
  inquire(file="problem.bin", exist=have_bin)
  if (have_bin) then
    open(11, file="problem.bin", status="old", action="read", access="stream", form="unformatted")
    read(11) ndof, nNode, nElem, ntot, nranks_file
  else
    open(11, file="problem.txt", status="old", action="read")
    read(11, *) ndof, nNode, nElem, ntot, nranks_file
  end if
  if (nranks_file /= nprocs) stop "the problem file was prepared for another number of ranks"
  nsize = 4 * ndof
  allocate(sizes(nprocs), owner(nElem), conn(4, nElem), edof(nsize, nElem), xyz(3, nNode), applied(nNode * ndof))
  if (have_bin) then
    read(11) sizes
    read(11) owner
    read(11) xyz
    read(11) conn
    read(11) edof
    read(11) applied
    read(11) elemData
  else
    read(11, *) sizes
    read(11, *) owner
    read(11, *) xyz
    read(11, *) conn
    read(11, *) edof
    read(11, *) applied
    read(11, *) elemData
  end if
  close(11)
  timeData = (/ 0.0d0, 1.0d0, 0.0d0 /)

  allocate(nnz_d(max(sizes(me + 1), 1)), nnz_o(max(sizes(me + 1), 1)))
  nnz_d = 50; nnz_o = 25
  call solver%initialise(sizes(me + 1), ntot, nnz_d, nnz_o)

  allocate(rows(nsize), Kl(nsize, nsize), Fl(nsize), zeroK(nsize, nsize), valC(nsize), valDotC(nsize))
  zeroK = 0.0d0; valC = 0.0d0; valDotC = 0.0d0

  call get_environment_variable("PFEM_CHECK_MODE", mode)
  if (trim(mode) == "device") then
    ! the element loop on the GPU: the rank's elements in the driver's own array layout (nElem_local, npElem) etc.
    nloc = count(owner == me)
    allocate(conn_loc(nloc, 4), edof_loc(nloc, nsize), coords(nNode, 3))
    k = 0
    do e = 1, nElem
      if (owner(e) /= me) cycle
      k = k + 1
      conn_loc(k, :) = conn(:, e)
      edof_loc(k, :) = edof(:, e)
    end do
    coords = transpose(xyz)
    call solver%uploadMeshToDevice(merge(2, 3, ndof == 1), conn_loc, coords, edof_loc, applied)
    call solver%assembleOnDevice(elemData, timeData)
  else

  ! the pattern: zeros with INSERT_VALUES for every element this rank owns
  call system_clock(tk0, tkr)
  do e = 1, nElem
    if (owner(e) /= me) cycle
    rows = edof(:, e)
    call MatSetValues(solver%mtx, nsize, rows, nsize, rows, zeroK, INSERT_VALUES, ierr)
  end do
  call system_clock(tk1)
  call solver%setZero()
  call system_clock(tk2)

  ! the element loop: element routine, matrix block, lifting of the prescribed values, vector
  ! (what the reference times as "assembly", tetrapoissonparallelimpl1.F:826-893)
  do e = 1, nElem
    if (owner(e) /= me) cycle
    do a = 1, 4
      xn(a) = xyz(1, conn(a, e)); yn(a) = xyz(2, conn(a, e)); zn(a) = xyz(3, conn(a, e))
    end do
    if (ndof == 1) then
      call StiffnessResidualPoissonLinearTetra(xn, yn, zn, elemData, timeData, valC, valDotC, Kl, Fl)
    else
      call StiffnessResidualElasticityLinearTetra(xn, yn, zn, elemData, timeData, valC, valDotC, Kl, Fl)
    end if
    rows = edof(:, e)
    call MatSetValues(solver%mtx, nsize, rows, nsize, rows, Kl, ADD_VALUES, ierr)
    do i = 1, nsize
      if (rows(i) /= -1) cycle
      node = conn((i - 1) / ndof + 1, e)
      d = mod(i - 1, ndof) + 1
      fact = applied((node - 1) * ndof + d)
      do j = 1, nsize
        if (rows(j) /= -1) Fl(j) = Fl(j) - Kl(j, i) * fact
      end do
    end do
    call VecSetValues(solver%rhsVec, nsize, rows, Fl, ADD_VALUES, ierr)
  end do
  call system_clock(tk3)
  end if

  call solver%factoriseAndSolve()
  call system_clock(tk4)
  if (trim(mode) /= "device" .and. me == 0) then
    ! seconds: INSERT_VALUES pass | setZero (pattern finalised on the device) | element loop with ADD_VALUES | factoriseAndSolve
    write(*, '(A,4(1X,ES14.6))') "TIMING", dble(tk1 - tk0) / dble(tkr), dble(tk2 - tk1) / dble(tkr), dble(tk3 - tk2) / dble(tkr), dble(tk4 - tk3) / dble(tkr)
  end if

  call VecScatterCreateToAll(solver%solnVec, ctx, vec_seq, ierr)
  call VecScatterBegin(ctx, solver%solnVec, vec_seq, INSERT_VALUES, SCATTER_FORWARD, ierr)
  call VecScatterEnd(ctx, solver%solnVec, vec_seq, INSERT_VALUES, SCATTER_FORWARD, ierr)
  call VecScatterDestroy(ctx, ierr)
  call VecGetArray(vec_seq, xx_v, xx_i, ierr)
  if (me == 0) then
    open(12, file="solution.txt", status="unknown", action="write")
    write(12, *) solver%its, solver%reason
    do k = 1, ntot
      write(12, '(ES25.17)') xx_v(xx_i + k)
    end do
    close(12)
  end if
  call VecRestoreArray(vec_seq, xx_v, xx_i, ierr)
  call solver%free()
  call PetscFinalize(ierr)
end program boundary_check
