!  petsckspdef.h -- see petscsysdef.h
#if !defined(PFEM_PETSCKSPDEF_H)
#define PFEM_PETSCKSPDEF_H
#include "petsc/finclude/petscmatdef.h"
#define KSP integer(kind=8)
#define KSPConvergedReason integer
#endif
