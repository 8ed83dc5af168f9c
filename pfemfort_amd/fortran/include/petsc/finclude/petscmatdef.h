!  petscmatdef.h -- see petscsysdef.h
#if !defined(PFEM_PETSCMATDEF_H)
#define PFEM_PETSCMATDEF_H
#include "petsc/finclude/petscvecdef.h"
#define Mat integer(kind=8)
#define MatInfo double precision
#endif
