!  petscsysdef.h -- build-owned stand-in for PETSc's Fortran include of the same name, so that
!  the PFEMFort drivers (tetrapoissonparallelimpl1.F:28-41) preprocess unchanged against
!  libpfem_amd.  PETSc objects become 8-byte handles (PETSc itself uses PetscFortranAddr).
#if !defined(PFEM_PETSCSYSDEF_H)
#define PFEM_PETSCSYSDEF_H
#define PetscErrorCode integer
#define PetscInt integer
#define PetscMPIInt integer
#define PetscBool logical
#define PetscScalar double precision
#define PetscReal double precision
#define PetscOffset integer(kind=8)
#define PetscFortranAddr integer(kind=8)
#define PetscLogStage integer
#define PetscViewer integer(kind=8)
#define CHKERRQ(n) if (n .ne. 0) then; call pfem_chkerr(n); endif
#endif
