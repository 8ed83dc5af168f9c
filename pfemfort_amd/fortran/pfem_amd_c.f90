! pfem_amd_c.f90 -- BIND(C) interfaces to libpfem_amd.so (include/pfem_amd.h).
! Build-owned Fortran side of the drop-in boundary: the modules in this directory carry the
! reference's own module / procedure / type names so that tetrapoissonparallelimpl1.F and
! tetraelasticityparallelimpl1.F compile UNCHANGED against them (see INTEGRATION.md).
module pfem_amd_c
  use iso_c_binding
  implicit none
  interface
    integer(c_int) function pfem_poisson_tria_ke(x, y, ed, td, vc, K, F) bind(C, name="pfem_poisson_tria_ke")
      import
      real(c_double) :: x(3), y(3), ed(*), td(*), vc(3), K(3,3), F(3)
    end function
    integer(c_int) function pfem_poisson_tet_ke(x, y, z, ed, td, vc, K, F) bind(C, name="pfem_poisson_tet_ke")
      import
      real(c_double) :: x(4), y(4), z(4), ed(*), td(*), vc(4), K(4,4), F(4)
    end function
    integer(c_int) function pfem_elast_tet_ke(x, y, z, ed, td, vc, K, F) bind(C, name="pfem_elast_tet_ke")
      import
      real(c_double) :: x(4), y(4), z(4), ed(*), td(*), vc(12), K(12,12), F(12)
    end function
    integer(c_int) function pfem_elast_tria_ke(x, y, ed, td, vc, K, F) bind(C, name="pfem_elast_tria_ke")
      import
      real(c_double) :: x(3), y(3), ed(*), td(*), vc(6), K(6,6), F(6)
    end function
    integer(c_int) function pfem_solver_create(s, size_local, size_global, row_start, diag_nnz, offdiag_nnz, device) &
        bind(C, name="pfem_solver_create")
      import
      type(c_ptr) :: s
      integer(c_int64_t), value :: size_local, size_global, row_start
      integer(c_int) :: diag_nnz(*), offdiag_nnz(*)
      integer(c_int), value :: device
    end function
    integer(c_int) function pfem_solver_destroy(s) bind(C, name="pfem_solver_destroy")
      import
      type(c_ptr), value :: s
    end function
    integer(c_int) function pfem_solver_set_tolerances(s, rtol, abstol, dtol, maxits) bind(C, name="pfem_solver_set_tolerances")
      import
      type(c_ptr), value :: s
      real(c_double), value :: rtol, abstol, dtol
      integer(c_int), value :: maxits
    end function
    integer(c_int) function pfem_solver_set_preconditioner(s, pc) bind(C, name="pfem_solver_set_preconditioner")
      import
      type(c_ptr), value :: s
      integer(c_int), value :: pc
    end function
    integer(c_int) function pfem_solver_set_cg_single_reduction(s, on) bind(C, name="pfem_solver_set_cg_single_reduction")
      import
      type(c_ptr), value :: s
      integer(c_int), value :: on
    end function
    integer(c_int) function pfem_solver_set_amg_cycle(s, cycle) bind(C, name="pfem_solver_set_amg_cycle")
      import
      type(c_ptr), value :: s
      integer(c_int), value :: cycle
    end function
    integer(c_int) function pfem_solver_set_zero(s) bind(C, name="pfem_solver_set_zero")
      import
      type(c_ptr), value :: s
    end function
    integer(c_int) function pfem_solver_print_info(s) bind(C, name="pfem_solver_print_info")
      import
      type(c_ptr), value :: s
    end function
    integer(c_int) function pfem_mat_set_values(s, m, idxm, n, idxn, v, mode) bind(C, name="pfem_mat_set_values")
      import
      type(c_ptr), value :: s
      integer(c_int), value :: m, n, mode
      integer(c_int) :: idxm(*), idxn(*)
      real(c_double) :: v(*)
    end function
    integer(c_int) function pfem_vec_set_values(s, n, idx, v, mode) bind(C, name="pfem_vec_set_values")
      import
      type(c_ptr), value :: s
      integer(c_int), value :: n, mode
      integer(c_int) :: idx(*)
      real(c_double) :: v(*)
    end function
    integer(c_int) function pfem_solver_assemble_matrix_and_vector(s, n, rows, cols, K, F) &
        bind(C, name="pfem_solver_assemble_matrix_and_vector")
      import
      type(c_ptr), value :: s
      integer(c_int), value :: n
      integer(c_int) :: rows(*), cols(*)
      type(c_ptr), value :: K, F
    end function
    integer(c_int) function pfem_solver_factorise(s) bind(C, name="pfem_solver_factorise")
      import
      type(c_ptr), value :: s
    end function
    integer(c_int) function pfem_solver_solve(s, its, reason, rnorm) bind(C, name="pfem_solver_solve")
      import
      type(c_ptr), value :: s
      integer(c_int) :: its, reason
      real(c_double) :: rnorm
    end function
    integer(c_int) function pfem_solver_factorise_and_solve(s, its, reason, rnorm) bind(C, name="pfem_solver_factorise_and_solve")
      import
      type(c_ptr), value :: s
      integer(c_int) :: its, reason
      real(c_double) :: rnorm
    end function
    integer(c_int) function pfem_solver_status(s, st) bind(C, name="pfem_solver_status")
      import
      type(c_ptr), value :: s
      integer(c_int) :: st
    end function
    integer(c_int) function pfem_solver_get_solution(s, x) bind(C, name="pfem_solver_get_solution")
      import
      type(c_ptr), value :: s
      real(c_double) :: x(*)
    end function
    integer(c_int) function pfem_write_vtk(path, ndim, nElem, nNode, npElem, ndof, coords, conn, procid, soln) &
        bind(C, name="pfem_write_vtk")
      import
      character(kind=c_char) :: path(*)
      integer(c_int), value :: ndim, npElem, ndof
      integer(c_int64_t), value :: nElem, nNode
      real(c_double) :: coords(*), soln(*)
      integer(c_int) :: conn(*), procid(*)
    end function
    ! the element loop as one device call (the fast path of include/pfem_amd.h, section 4)
    integer(c_int) function pfem_mesh_upload(s, kind, nElem, conn, nNode, xyz, edof, solnApplied) bind(C, name="pfem_mesh_upload")
      import
      type(c_ptr), value :: s
      integer(c_int), value :: kind
      integer(c_int64_t), value :: nElem, nNode
      integer(c_int) :: conn(*), edof(*)          ! SoA = the column-major layout of elemNodeConn(nElem,npElem) / ElemDofArray(nElem,nsize); 0-based
      real(c_double) :: xyz(*), solnApplied(*)    ! coords(nNode,ndim) column-major; solnApplied((node-1)*ndof+dof)
    end function
    integer(c_int) function pfem_pattern_build(s) bind(C, name="pfem_pattern_build")
      import
      type(c_ptr), value :: s
    end function
    integer(c_int) function pfem_assemble(s, elemData, timeData) bind(C, name="pfem_assemble")
      import
      type(c_ptr), value :: s
      real(c_double) :: elemData(*), timeData(*)
    end function
    function pfem_last_error_string() bind(C, name="pfem_last_error_string") result(p)
      import
      type(c_ptr) :: p
    end function
    function pfem_strerror(code) bind(C, name="pfem_strerror") result(p)
      import
      integer(c_int), value :: code
      type(c_ptr) :: p
    end function
    ! pfem_mpi.cpp (MPI flavour of the shim only)
    integer(c_int) function pfem_mpi_attach(s, fcomm, row_start, size_local) bind(C, name="pfem_mpi_attach")
      import
      type(c_ptr), value :: s
      integer(c_int), value :: fcomm
      integer(c_int64_t), value :: row_start, size_local
    end function
    integer(c_int) function pfem_mpi_gather_solution(s, fcomm, size_local, out) bind(C, name="pfem_mpi_gather_solution")
      import
      type(c_ptr), value :: s
      integer(c_int), value :: fcomm
      integer(c_int64_t), value :: size_local
      real(c_double) :: out(*)
    end function
    integer(c_int) function pfem_mpi_pick_device(fcomm, device) bind(C, name="pfem_mpi_pick_device")
      import
      integer(c_int), value :: fcomm
      integer(c_int) :: device
    end function
  end interface
contains
  ! 8-byte PETSc-style handle <-> C pointer
  function pfem_h2p(h) result(p)
    integer(kind=8), intent(in) :: h
    type(c_ptr) :: p
    p = transfer(h, p)
  end function
  function pfem_p2h(p) result(h)
    type(c_ptr), intent(in) :: p
    integer(kind=8) :: h
    h = transfer(p, h)
  end function
  subroutine pfem_print_cstr(p)
    type(c_ptr), intent(in) :: p
    character(kind=c_char), pointer :: s(:)
    integer :: i
    if (.not. c_associated(p)) return
    call c_f_pointer(p, s, [4096])
    do i = 1, 4096
      if (s(i) == c_null_char) exit
      write(*, '(A)', advance='no') s(i)
    end do
    write(*, *)
  end subroutine
end module pfem_amd_c

! CHKERRQ(n) of the include files lands here (PETSc: MPI_Abort)
subroutine pfem_chkerr(n)
  use pfem_amd_c
  implicit none
  integer, intent(in) :: n
  write(*, '(A,I0,A)', advance='no') " pfem_amd error ", n, ": "
  call pfem_print_cstr(pfem_strerror(int(n, c_int)))
  call pfem_print_cstr(pfem_last_error_string())
  stop 1
end subroutine pfem_chkerr
