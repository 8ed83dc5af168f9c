!  petscpcdef.h -- see petscsysdef.h
#if !defined(PFEM_PETSCPCDEF_H)
#define PFEM_PETSCPCDEF_H
#include "petsc/finclude/petscmatdef.h"
#define PC integer(kind=8)
#endif
