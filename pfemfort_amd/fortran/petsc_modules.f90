! Build-owned stand-ins for the PETSc Fortran modules the drivers USE
! (tetrapoissonparallelimpl1.F:36-39): constants, the MPI_Wtime interface, and the options that
! PetscInitialize("petsc_options.dat") / KSPSetFromOptions would have read.
! Two flavours (fortran/Makefile): the default one carries a single-rank MPI (petsc_shim.f90);
! with -DPFEM_WITH_MPI the constants are the real MPI's (mpif.h) and PETSC_COMM_WORLD is
! MPI_COMM_WORLD, so the drivers run under mpiexec, one rank per GPU (pfem_mpi.cpp).
module petscvec
  implicit none
#ifdef PFEM_WITH_MPI
  include 'mpif.h'
  integer, parameter :: PETSC_COMM_WORLD = MPI_COMM_WORLD, PETSC_COMM_SELF = MPI_COMM_SELF
#else
  integer, parameter :: PETSC_COMM_WORLD = 0, PETSC_COMM_SELF = 1
  integer, parameter :: MPI_COMM_WORLD = 0
  integer, parameter :: MPI_INT = 1, MPI_INTEGER = 1, MPI_DOUBLE_PRECISION = 2, MPI_SUM = 1
#endif
  integer, parameter :: INSERT_VALUES = 1, ADD_VALUES = 2            ! PFEM_INSERT_VALUES / PFEM_ADD_VALUES
  integer, parameter :: SCATTER_FORWARD = 0, SCATTER_REVERSE = 1
  logical, parameter :: PETSC_TRUE = .true., PETSC_FALSE = .false.
  character(len=1), parameter :: PETSC_NULL_CHARACTER = ' '
  ! KSP options (PETSc defaults; overridden by petsc_options.dat or PFEM_KSP_* environment variables)
  double precision, save :: pfem_opt_rtol = 1.0d-5, pfem_opt_atol = 1.0d-50, pfem_opt_dtol = 1.0d5
  integer, save :: pfem_opt_maxits = 10000
  integer, save :: pfem_opt_pc = 0          ! 0 = -pc_type jacobi (default), 1 = pbjacobi (node blocks), 2 = gamg (aggregation multigrid)
  integer, save :: pfem_opt_cycle = 0       ! -pc_mg_cycle_type: 1 = v, 2 = w, 0 = not given (V, PETSc's default too)
  integer, save :: pfem_opt_single = -1     ! -ksp_cg_single_reduction: 1 / 0, -1 = not given (PFEM_CG_SINGLE_REDUCTION decides)
  ! gathered solution of VecScatterCreateToAll / VecGetArray
  double precision, allocatable, target, save :: pfem_seq_soln(:)
  interface
#ifndef PFEM_WITH_MPI
    double precision function MPI_Wtime()
    end function
#endif
    ! GNU extension the drivers rely on (tetrapoissonparallelimpl1.F:158); flang wants a type
    integer function iargc()
    end function
  end interface
end module petscvec

module petscmat
  use petscvec
end module petscmat

module petscksp
  use petscvec
end module petscksp

module petscpc
  use petscvec
end module petscpc
