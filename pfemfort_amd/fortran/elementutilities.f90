! Build-owned replacements of MODULE ElementUtilitiesPoisson (elementutilitiespoisson.F) and
! MODULE ElementUtilitiesElasticity3D (elementutilitieselasticity3D.F:248-393): same names,
! argument order, shapes and STOP messages; the arithmetic lives in libpfem_amd
! (csrc/pfem_elem.hpp, shared with the gfx950 kernels).
module ElementUtilitiesPoisson
  use pfem_amd_c
  implicit none
contains
  subroutine StiffnessResidualPoissonLinearTria(xNode, yNode, elemData, timeData, valC, valDotC, Klocal, Flocal)
    double precision, dimension(:) :: elemData, timeData
    double precision :: xNode(3), yNode(3), valC(3), valDotC(3), Klocal(3,3), Flocal(3)
    if (pfem_poisson_tria_ke(xNode, yNode, elemData, timeData, valC, Klocal, Flocal) /= 0) then
      stop " Negative Jacobian for the Tria element in Poisson"
    end if
  end subroutine
  subroutine StiffnessResidualPoissonLinearTetra(xNode, yNode, zNode, elemData, timeData, valC, valDotC, Klocal, Flocal)
    double precision, dimension(:) :: elemData, timeData
    double precision :: xNode(4), yNode(4), zNode(4), valC(4), valDotC(4), Klocal(4,4), Flocal(4)
    if (pfem_poisson_tet_ke(xNode, yNode, zNode, elemData, timeData, valC, Klocal, Flocal) /= 0) then
      stop " Negative Jacobian for the Tet element in Poisson"
    end if
  end subroutine
end module ElementUtilitiesPoisson

module ElementUtilitiesElasticity3D
  use pfem_amd_c
  implicit none
contains
  subroutine StiffnessResidualElasticityLinearTetra(xNode, yNode, zNode, elemData, timeData, valC, valDotC, Klocal, Flocal)
    double precision, dimension(4) :: xNode, yNode, zNode
    double precision, dimension(:) :: elemData, timeData, valC, valDotC
    double precision, dimension(12,12) :: Klocal
    double precision, dimension(12) :: Flocal
    if (pfem_elast_tet_ke(xNode, yNode, zNode, elemData, timeData, valC, Klocal, Flocal) /= 0) then
      stop " Negative Jacobian for the Tet element in Elasticity"
    end if
  end subroutine
end module ElementUtilitiesElasticity3D

! 2-D sibling (SURVEY 8f.1): MODULE ElementUtilitiesElasticity2D, the one routine on the implicit path
module ElementUtilitiesElasticity2D
  use pfem_amd_c
  implicit none
contains
  subroutine StiffnessResidualElasticityLinearTria(xNode, yNode, elemData, timeData, valC, valDotC, Klocal, Flocal)
    double precision, dimension(3) :: xNode, yNode
    double precision, dimension(:) :: elemData, timeData, valC, valDotC
    double precision, dimension(6,6) :: Klocal
    double precision, dimension(6) :: Flocal
    if (pfem_elast_tria_ke(xNode, yNode, elemData, timeData, valC, Klocal, Flocal) /= 0) then
      stop " Negative Jacobian for the Tria element in Elasticity"
    end if
  end subroutine
end module ElementUtilitiesElasticity2D
