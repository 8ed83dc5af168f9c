!  petscvecdef.h -- see petscsysdef.h
#if !defined(PFEM_PETSCVECDEF_H)
#define PFEM_PETSCVECDEF_H
#include "petsc/finclude/petscsysdef.h"
#define Vec integer(kind=8)
#define VecScatter integer(kind=8)
#define IS integer(kind=8)
#endif
