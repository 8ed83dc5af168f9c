! Build-owned replacement of MODULE Module_SolverPetsc (solverpetsc.F): TYPE PetscSolver with the
! same public components and procedures; Mat/Vec/KSP components all alias the one pfem_solver
! handle, because the drivers only hand them back to MatSetValues / VecSetValues / VecGetArray.
module Module_SolverPetsc
  use pfem_amd_c
#ifdef PFEM_WITH_MPI
  use petscvec, only: pfem_opt_rtol, pfem_opt_atol, pfem_opt_dtol, pfem_opt_maxits, pfem_opt_pc, pfem_opt_single, pfem_opt_cycle, PETSC_COMM_WORLD, MPI_INTEGER, MPI_SUM
#else
  use petscvec, only: pfem_opt_rtol, pfem_opt_atol, pfem_opt_dtol, pfem_opt_maxits, pfem_opt_pc, pfem_opt_single, pfem_opt_cycle
#endif
  implicit none
  integer, parameter :: SOLVER_EMPTY=1, PATTERN_OK=2, INIT_OK=3, ASSEMBLY_OK=4, FACTORISE_OK=5   ! solverpetsc.F:64-68

  type PetscSolver
    integer(kind=8) :: rhsVec = 0, solnVec = 0, solnPrev = 0, mtx = 0, ksp = 0, pc = 0
    integer :: nRow = 0, nCol = 0, nnz = 0
    double precision :: norm = 0.0d0
    integer :: currentStatus = 0
    integer :: its = 0, reason = 0
    ! not in the reference type: the rank's row block, and whether the communicator is wired (MPI flavour)
    integer(c_int64_t) :: row_start = 0, size_local = 0
    logical :: attached = .false.
  contains
    procedure :: initialise
    procedure :: setZero
    procedure :: free
    procedure :: printInfo
    procedure :: assembleMatrix
    procedure :: assembleVector
    procedure :: assembleMatrixAndVector
    procedure :: factorise
    procedure :: solve
    procedure :: factoriseAndSolve
    ! not in the reference type: the driver's whole element loop (tetrapoissonparallelimpl1.F:786-884) as device calls
    procedure :: uploadMeshToDevice
    procedure :: assembleOnDevice
  end type PetscSolver

contains

  subroutine sync_status(this)
    class(PetscSolver) :: this
    integer(c_int) :: st
    if (pfem_solver_status(pfem_h2p(this%mtx), st) == 0) this%currentStatus = st
  end subroutine

  ! solverpetsc.F:116-214
  subroutine initialise(this, size_local, size_global, diag_nnz, offdiag_nnz)
    class(PetscSolver) :: this
    integer, intent(in) :: size_global, size_local
    integer, dimension(:) :: diag_nnz, offdiag_nnz
    type(c_ptr) :: h
    integer :: ierr, rs, me
    integer(c_int) :: dev
    this%nRow = size_global
    this%nCol = size_global
    rs = 0
    dev = -1
#ifdef PFEM_WITH_MPI
    ! PETSc's ownership ranges: rows of rank r start after the size_local of the lower ranks
    call MPI_Comm_rank(PETSC_COMM_WORLD, me, ierr)
    call MPI_Exscan(size_local, rs, 1, MPI_INTEGER, MPI_SUM, PETSC_COMM_WORLD, ierr)
    if (me == 0) rs = 0
    ierr = pfem_mpi_pick_device(PETSC_COMM_WORLD, dev)
    if (ierr /= 0) call pfem_chkerr(ierr)
#endif
    me = 0
#ifdef PFEM_WITH_MPI
    call MPI_Comm_rank(PETSC_COMM_WORLD, me, ierr)
#endif
    ! one loud line: the reference sets KSPCG + PCBJACOBI (per-rank ILU(0)) at solverpetsc.F:187,206; this library runs
    ! CG with the preconditioner named here, so iteration counts are not those of the reference's PETSc run
    if (me == 0) then
      if (pfem_opt_pc == 2) then
        write(*,*) " pfem_amd: KSP = cg, PC = gamg (plain-aggregation multigrid V-cycle) on the GPU; the reference's PCBJACOBI/ILU(0) is not reproduced"
      else if (pfem_opt_pc == 1) then
        write(*,*) " pfem_amd: KSP = cg, PC = pbjacobi (node-block Jacobi) on the GPU; the reference's PCBJACOBI/ILU(0) is not reproduced"
      else
        write(*,*) " pfem_amd: KSP = cg, PC = jacobi on the GPU; the reference's PCBJACOBI/ILU(0) is not reproduced"
      end if
    end if
    this%row_start = rs
    this%size_local = size_local
    this%attached = .false.
    ierr = pfem_solver_create(h, int(size_local, c_int64_t), int(size_global, c_int64_t), int(rs, c_int64_t), &
                              diag_nnz, offdiag_nnz, dev)
    if (ierr /= 0) call pfem_chkerr(ierr)
    ierr = pfem_solver_set_tolerances(h, pfem_opt_rtol, pfem_opt_atol, pfem_opt_dtol, int(pfem_opt_maxits, c_int))
    if (ierr /= 0) call pfem_chkerr(ierr)
    ierr = pfem_solver_set_preconditioner(h, int(pfem_opt_pc, c_int))       ! KSPSetFromOptions: -pc_type
    if (ierr == 0) ierr = pfem_solver_set_cg_single_reduction(h, int(pfem_opt_single, c_int))   ! -ksp_cg_single_reduction
    if (ierr == 0 .and. pfem_opt_cycle /= 0) ierr = pfem_solver_set_amg_cycle(h, int(pfem_opt_cycle, c_int))   ! -pc_mg_cycle_type
    if (ierr /= 0) call pfem_chkerr(ierr)
    this%mtx = pfem_p2h(h)
    this%rhsVec = this%mtx
    this%solnVec = this%mtx
    this%ksp = this%mtx
    this%pc = this%mtx
    this%currentStatus = SOLVER_EMPTY
  end subroutine initialise

  ! solverpetsc.F:222-246
  subroutine setZero(this)
    class(PetscSolver) :: this
    integer :: ierr
    ierr = pfem_solver_set_zero(pfem_h2p(this%mtx))
    if (ierr /= 0) call pfem_chkerr(ierr)
#ifdef PFEM_WITH_MPI
    ! the first setZero finalises the pattern (MatAssembly of the INSERT_VALUES pass), and with it the
    ! local numbering: exchange the ghost lists, install the neighbour plan and the communication backend
    if (.not. this%attached) then
      ierr = pfem_mpi_attach(pfem_h2p(this%mtx), PETSC_COMM_WORLD, this%row_start, this%size_local)
      if (ierr /= 0) call pfem_chkerr(ierr)
      this%attached = .true.
    end if
#endif
    call sync_status(this)
  end subroutine setZero

  ! EXTENSION (no counterpart in solverpetsc.F): hand the rank's elements to the GPU once -- the driver's own arrays in
  ! their own layout: elemNodeConn(nElem,npElem) with NEW 1-based node ids of the rank's elements, coords(nNode,ndim)
  ! gathered through node_map_get_old, ElemDofArray(nElem,nsize) (0-based, -1 = Dirichlet), solnApplied -- and build the
  ! pattern on the device.  Replaces the INSERT_VALUES loop (:786-802) and the first setZero (:817).
  ! kind: 1 tria Poisson, 2 tet Poisson, 3 tet elasticity, 5 tria elasticity (include/pfem_amd.h).
  subroutine uploadMeshToDevice(this, kind, elemNodeConn, coords, ElemDofArray, solnApplied)
    class(PetscSolver) :: this
    integer, intent(in) :: kind
    integer, dimension(:,:), intent(in) :: elemNodeConn, ElemDofArray
    double precision, dimension(:,:), intent(in) :: coords
    double precision, dimension(:), intent(in) :: solnApplied
    integer(c_int), allocatable :: conn0(:,:), edof(:,:)
    double precision, allocatable :: xyz(:,:), sa(:)
    integer :: ierr
    conn0 = elemNodeConn - 1                      ! 0-based copy; column-major (nElem,npElem) IS the SoA layout of the ABI
    edof = ElemDofArray
    xyz = coords
    sa = solnApplied
    ierr = pfem_mesh_upload(pfem_h2p(this%mtx), int(kind, c_int), int(size(conn0, 1), c_int64_t), conn0, &
                            int(size(xyz, 1), c_int64_t), xyz, edof, sa)
    if (ierr /= 0) call pfem_chkerr(ierr)
#ifdef PFEM_WITH_MPI
    if (.not. this%attached) then                 ! the local numbering exists now: neighbour plan + communication backend
      ierr = pfem_mpi_attach(pfem_h2p(this%mtx), PETSC_COMM_WORLD, this%row_start, this%size_local)
      if (ierr /= 0) call pfem_chkerr(ierr)
      this%attached = .true.
    end if
#endif
    ierr = pfem_pattern_build(pfem_h2p(this%mtx))
    if (ierr /= 0) call pfem_chkerr(ierr)
    call sync_status(this)
  end subroutine uploadMeshToDevice

  ! EXTENSION: setZero + the element loop (:817-884: element routine, Dirichlet lifting, ADD_VALUES) as ONE device call
  subroutine assembleOnDevice(this, elemData, timeData)
    class(PetscSolver) :: this
    double precision, dimension(:), intent(in) :: elemData, timeData
    double precision :: ed(8), td(8)
    integer :: ierr
    ed = 0.0d0; td = 0.0d0
    ed(1:min(size(elemData), 8)) = elemData(1:min(size(elemData), 8))
    td(1:min(size(timeData), 8)) = timeData(1:min(size(timeData), 8))
    ierr = pfem_assemble(pfem_h2p(this%mtx), ed, td)
    if (ierr /= 0) call pfem_chkerr(ierr)
    call sync_status(this)
  end subroutine assembleOnDevice

  ! solverpetsc.F:254-278
  subroutine free(this)
    class(PetscSolver) :: this
    integer :: ierr
    if (this%mtx /= 0) ierr = pfem_solver_destroy(pfem_h2p(this%mtx))
    this%mtx = 0; this%rhsVec = 0; this%solnVec = 0; this%ksp = 0; this%pc = 0
  end subroutine free

  ! solverpetsc.F:286-320
  subroutine printInfo(this)
    class(PetscSolver) :: this
    integer :: ierr
    ierr = pfem_solver_print_info(pfem_h2p(this%mtx))
  end subroutine printInfo

  ! solverpetsc.F:328-401
  subroutine assembleMatrix(this, RINDICES, CINDICES, KLOCAL)
    class(PetscSolver) :: this
    integer, dimension(:) :: RINDICES, CINDICES
    double precision, dimension(:,:), target :: KLOCAL
    double precision, allocatable, target :: K(:,:)
    integer :: ierr
    K = KLOCAL
    ierr = pfem_solver_assemble_matrix_and_vector(pfem_h2p(this%mtx), size(RINDICES), RINDICES, CINDICES, c_loc(K), c_null_ptr)
    if (ierr /= 0) call pfem_chkerr(ierr)
  end subroutine assembleMatrix

  subroutine assembleVector(this, RINDICES, FLOCAL)
    class(PetscSolver) :: this
    integer, dimension(:) :: RINDICES
    double precision, dimension(:), target :: FLOCAL
    double precision, allocatable, target :: F(:)
    integer :: ierr
    F = FLOCAL
    ierr = pfem_solver_assemble_matrix_and_vector(pfem_h2p(this%mtx), size(RINDICES), RINDICES, RINDICES, c_null_ptr, c_loc(F))
    if (ierr /= 0) call pfem_chkerr(ierr)
  end subroutine assembleVector

  subroutine assembleMatrixAndVector(this, RINDICES, CINDICES, KLOCAL, FLOCAL)
    class(PetscSolver) :: this
    integer, dimension(:) :: RINDICES, CINDICES
    double precision, dimension(:,:) :: KLOCAL
    double precision, dimension(:) :: FLOCAL
    double precision, allocatable, target :: K(:,:), F(:)
    integer :: ierr
    K = KLOCAL; F = FLOCAL
    ierr = pfem_solver_assemble_matrix_and_vector(pfem_h2p(this%mtx), size(RINDICES), RINDICES, CINDICES, c_loc(K), c_loc(F))
    if (ierr /= 0) call pfem_chkerr(ierr)
  end subroutine assembleMatrixAndVector

  ! solverpetsc.F:409-423
  subroutine factorise(this)
    class(PetscSolver) :: this
    if (this%currentStatus /= ASSEMBLY_OK) then
      write(*,*) "Assemble matrix first before solving it! "
      stop " Aborting... in 'solverpetsc->factorise' "
    end if
    this%currentStatus = FACTORISE_OK          ! status check only, as in the reference
  end subroutine factorise

  ! solverpetsc.F:431-490
  subroutine solve(this)
    class(PetscSolver) :: this
    integer(c_int) :: its, reason
    real(c_double) :: rn
    integer :: ierr
    if (this%currentStatus /= FACTORISE_OK) then
      write(*,*) "Factorise matrix first before solving it! "
      stop " Aborting... in 'solverpetsc->solve' "
    end if
    write(*,*) " Solving the matrix system "
    ! MatAssembly/VecAssembly (push host-staged values) + KSPSolve on the GPU
    ierr = pfem_solver_factorise_and_solve(pfem_h2p(this%mtx), its, reason, rn)
    if (ierr /= 0) call pfem_chkerr(ierr)
    this%its = its; this%reason = reason; this%norm = rn
    if (reason < 0) then
      write(*,*) "Divergence."
    else
      write(*,*) "Convergence in", its, " iterations."
    end if
  end subroutine solve

  ! solverpetsc.F:498-509
  subroutine factoriseAndSolve(this)
    class(PetscSolver) :: this
    this%currentStatus = ASSEMBLY_OK
    call this%factorise()
    call this%solve()
  end subroutine factoriseAndSolve

end module Module_SolverPetsc
