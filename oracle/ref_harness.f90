! ref_harness.f90 -- TEST INFRASTRUCTURE (build-owned, not reference code).
!
! bind(C) entry points that forward to the PFEMFort reference element
! routines, which oracle/Makefile compiles IN PLACE from /root/reference/src
! with AMD flang into oracle/_ref/libpfem_ref.so.  Used only to pin the C
! restatement (oracle/pfem_oracle.c) and to generate tests/golden/*.npz.
!
! The explicit interfaces come from the reference's own modules
! (ElementUtilitiesPoisson, ElementUtilitiesElasticity3D); assumed-shape
! dummies (elemData, timeData, valC, valDotC) are satisfied by local copies.

subroutine ref_poisson_tet(x, y, z, ed, td, vc, K, F) bind(C, name="ref_poisson_tet")
  use iso_c_binding
  use ElementUtilitiesPoisson, only: StiffnessResidualPoissonLinearTetra
  implicit none
  real(c_double), intent(in)  :: x(4), y(4), z(4), ed(3), td(3), vc(4)
  real(c_double), intent(out) :: K(4,4), F(4)
  double precision :: xx(4), yy(4), zz(4), edd(3), tdd(3), vcc(4), vdd(4)
  xx = x; yy = y; zz = z; edd = ed; tdd = td; vcc = vc; vdd = 0.0d0
  call StiffnessResidualPoissonLinearTetra(xx, yy, zz, edd, tdd, vcc, vdd, K, F)
end subroutine ref_poisson_tet

subroutine ref_poisson_tria(x, y, ed, td, vc, K, F) bind(C, name="ref_poisson_tria")
  use iso_c_binding
  use ElementUtilitiesPoisson, only: StiffnessResidualPoissonLinearTria
  implicit none
  real(c_double), intent(in)  :: x(3), y(3), ed(2), td(3), vc(3)
  real(c_double), intent(out) :: K(3,3), F(3)
  double precision :: xx(3), yy(3), edd(2), tdd(3), vcc(3), vdd(3)
  xx = x; yy = y; edd = ed; tdd = td; vcc = vc; vdd = 0.0d0
  call StiffnessResidualPoissonLinearTria(xx, yy, edd, tdd, vcc, vdd, K, F)
end subroutine ref_poisson_tria

! Elasticity: linked against the reference routine with the documented
! 2-token patch (nGP=8 -> 1, ETYPE 1 -> 4; SURVEY finding 5) applied to a
! temporary copy at build time; the unpatched routine STOPs.
subroutine ref_elast_tet(x, y, z, ed, td, vc, K, F) bind(C, name="ref_elast_tet")
  use iso_c_binding
  use ElementUtilitiesElasticity3D, only: StiffnessResidualElasticityLinearTetra
  implicit none
  real(c_double), intent(in)  :: x(4), y(4), z(4), ed(6), td(3), vc(12)
  real(c_double), intent(out) :: K(12,12), F(12)
  double precision :: xx(4), yy(4), zz(4), edd(6), tdd(3), vcc(12), vdd(12)
  xx = x; yy = y; zz = z; edd = ed; tdd = td; vcc = vc; vdd = 0.0d0
  call StiffnessResidualElasticityLinearTetra(xx, yy, zz, edd, tdd, vcc, vdd, K, F)
end subroutine ref_elast_tet

! Batch drivers (loop in Fortran so 6000-element meshes are one call).
subroutine ref_poisson_tet_batch(n, x, y, z, ed, td, K, F) bind(C, name="ref_poisson_tet_batch")
  use iso_c_binding
  use ElementUtilitiesPoisson, only: StiffnessResidualPoissonLinearTetra
  implicit none
  integer(c_int64_t), value :: n
  real(c_double), intent(in)  :: x(4,n), y(4,n), z(4,n), ed(3), td(3)
  real(c_double), intent(out) :: K(4,4,n), F(4,n)
  double precision :: xx(4), yy(4), zz(4), edd(3), tdd(3), vcc(4), vdd(4)
  integer(c_int64_t) :: e
  edd = ed; tdd = td; vcc = 0.0d0; vdd = 0.0d0
  do e = 1, n
    xx = x(:,e); yy = y(:,e); zz = z(:,e)
    call StiffnessResidualPoissonLinearTetra(xx, yy, zz, edd, tdd, vcc, vdd, K(:,:,e), F(:,e))
  end do
end subroutine ref_poisson_tet_batch

subroutine ref_elast_tet_batch(n, x, y, z, ed, td, K, F) bind(C, name="ref_elast_tet_batch")
  use iso_c_binding
  use ElementUtilitiesElasticity3D, only: StiffnessResidualElasticityLinearTetra
  implicit none
  integer(c_int64_t), value :: n
  real(c_double), intent(in)  :: x(4,n), y(4,n), z(4,n), ed(6), td(3)
  real(c_double), intent(out) :: K(12,12,n), F(12,n)
  double precision :: xx(4), yy(4), zz(4), edd(6), tdd(3), vcc(12), vdd(12)
  integer(c_int64_t) :: e
  edd = ed; tdd = td; vcc = 0.0d0; vdd = 0.0d0
  do e = 1, n
    xx = x(:,e); yy = y(:,e); zz = z(:,e)
    call StiffnessResidualElasticityLinearTetra(xx, yy, zz, edd, tdd, vcc, vdd, K(:,:,e), F(:,e))
  end do
end subroutine ref_elast_tet_batch

subroutine ref_poisson_tria_batch(n, x, y, ed, td, K, F) bind(C, name="ref_poisson_tria_batch")
  use iso_c_binding
  use ElementUtilitiesPoisson, only: StiffnessResidualPoissonLinearTria
  implicit none
  integer(c_int64_t), value :: n
  real(c_double), intent(in)  :: x(3,n), y(3,n), ed(2), td(3)
  real(c_double), intent(out) :: K(3,3,n), F(3,n)
  double precision :: xx(3), yy(3), edd(2), tdd(3), vcc(3), vdd(3)
  integer(c_int64_t) :: e
  edd = ed; tdd = td; vcc = 0.0d0; vdd = 0.0d0
  do e = 1, n
    xx = x(:,e); yy = y(:,e)
    call StiffnessResidualPoissonLinearTria(xx, yy, edd, tdd, vcc, vdd, K(:,:,e), F(:,e))
  end do
end subroutine ref_poisson_tria_batch

! The reference's own VTK writer (writervtk.F), for the golden output file of the "output step".
subroutine ref_write_vtk(ndim, nElem, nNode, npElem, ndof, coords, conn1, procid, soln, fname, flen) bind(C, name="ref_write_vtk")
  use iso_c_binding
  use WriterVTK, only: writeoutputvtk
  implicit none
  integer(c_int), value :: ndim, nElem, nNode, npElem, ndof, flen
  real(c_double), intent(in) :: coords(nNode, ndim), soln(nNode*ndof)
  integer(c_int), intent(in) :: conn1(nElem, npElem), procid(nElem)
  character(kind=c_char), intent(in) :: fname(flen)
  character(len=flen) :: fn
  double precision, allocatable :: cc(:,:), ss(:)
  integer, allocatable :: ee(:,:), pp(:)
  integer :: i
  do i = 1, flen
    fn(i:i) = fname(i)
  end do
  cc = coords; ss = soln; ee = conn1; pp = procid
  call writeoutputvtk(ndim, nElem, nNode, npElem, ndof, cc, ee, pp, ss, fn)
end subroutine ref_write_vtk

subroutine ref_elast_tria_batch(n, x, y, ed, td, K, F) bind(C, name="ref_elast_tria_batch")
  use iso_c_binding
  use ElementUtilitiesElasticity2D, only: StiffnessResidualElasticityLinearTria
  implicit none
  integer(c_int64_t), value :: n
  real(c_double), intent(in)  :: x(3,n), y(3,n), ed(5), td(3)
  real(c_double), intent(out) :: K(6,6,n), F(6,n)
  double precision :: xx(3), yy(3), edd(5), tdd(3), vcc(6), vdd(6)
  integer(c_int64_t) :: e
  edd = ed; tdd = td; vcc = 0.0d0; vdd = 0.0d0
  do e = 1, n
    xx = x(:,e); yy = y(:,e)
    call StiffnessResidualElasticityLinearTria(xx, yy, edd, tdd, vcc, vdd, K(:,:,e), F(:,e))
  end do
end subroutine ref_elast_tria_batch
