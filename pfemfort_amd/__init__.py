"""pfemfort_amd -- MI355X-native implicit-FEM hot path behind PFEMFort's call surface.

Element stiffness (tria/tet Poisson, tet linear elasticity) -> sparse assembly -> Jacobi-PCG,
as hand-written HIP kernels for gfx950 behind a C ABI (include/pfem_amd.h); this package is the
thin host-side mirror of the reference's Fortran interface.  No CPU fallback exists.
"""
from . import _lib, host  # noqa: F401
from ._lib import (ELAST_TET, ELAST_TRIA, POISSON_TET, POISSON_TRIA, POISSON_TRIA_INLINE, PfemError,  # noqa: F401
                   device_count, device_info, device_memory)
from .drivers import (tetraelasticityparallelimpl1, tetrapoissonparallelimpl1,  # noqa: F401
                      triaelasticityparallelimpl1, triapoissonparallelimpl1, triapoissonserialimpl1)
from .solver import PetscSolver  # noqa: F401
