! Build-owned replacement of MODULE WriterVTK (writervtk.F:33-201): same name and argument list;
! the file is written by libpfem_amd's pfem_write_vtk, byte-identical to the reference writer
! (tests/test_host.py compares against files written by the reference's own writervtk.F).
module WriterVTK
  use pfem_amd_c
  implicit none
contains
  subroutine writeoutputvtk(ndim, nElem, nNode, npElem, ndof, coords, elemNodeConn, elem_procid, soln, fileName)
    integer, intent(in) :: ndim, nElem, nNode, npElem, ndof
    integer, dimension(:,:), intent(in) :: elemNodeConn          ! (nElem, npElem), 1-based
    integer, dimension(:), intent(in) :: elem_procid
    double precision, dimension(:,:), intent(in) :: coords       ! (nNode, ndim)
    double precision, dimension(:), intent(in) :: soln
    character(len=*) :: fileName
    integer(c_int), allocatable :: conn0(:,:), pid(:)
    double precision, allocatable :: xyz(:,:), sol(:)
    character(kind=c_char), allocatable :: cpath(:)
    integer :: i, n, ierr
    conn0 = elemNodeConn(1:nElem, 1:npElem) - 1                  ! column-major == SoA, 0-based
    pid = elem_procid(1:nElem)
    xyz = coords(1:nNode, 1:ndim)
    sol = soln(1:nNode*ndof)
    n = len_trim(fileName)
    allocate(cpath(n + 1))
    do i = 1, n
      cpath(i) = fileName(i:i)
    end do
    cpath(n + 1) = c_null_char
    ierr = pfem_write_vtk(cpath, int(ndim, c_int), int(nElem, c_int64_t), int(nNode, c_int64_t), int(npElem, c_int), &
                          int(ndof, c_int), xyz, conn0, pid, sol)
    if (ierr /= 0) call pfem_chkerr(ierr)
  end subroutine writeoutputvtk
end module WriterVTK
