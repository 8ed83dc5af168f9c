// pfem_internal.hpp -- declarations shared by the translation units of libpfem_amd.so
#pragma once

#include "../../include/pfem_amd.h"
#include "pfem_elem.hpp"

#include <cstdint>
#include <string>
#include <vector>

namespace pfem {

// per-kind geometry of the element families on the hot path
inline int kind_npelem(int kind) { return (kind == PFEM_POISSON_TET || kind == PFEM_ELAST_TET) ? 4 : 3; }
inline int kind_ndof(int kind) { return kind == PFEM_ELAST_TET ? 3 : (kind == PFEM_ELAST_TRIA ? 2 : 1); }
inline int kind_ndim(int kind) { return (kind == PFEM_POISSON_TET || kind == PFEM_ELAST_TET) ? 3 : 2; }
inline bool kind_valid(int kind) { return kind >= PFEM_POISSON_TRIA && kind <= PFEM_ELAST_TRIA; }

void set_last_error(const std::string &msg);

// structured box of genTetra.cpp (pfem_host.cpp): the three axis tables, as accumulated (raw) and as read back from the
// "%.8f" node file (rounded), and the Dirichlet value u = x^2+y^2+z^2 on float-rounded coordinates, "%.8f"-rounded
struct BoxAxes {
    std::vector<double> raw[3], rounded[3];
};
BoxAxes box_axes(double x0, double x1, int nEx, double y0, double y1, int nEy, double z0, double z1, int nEz);
double box_dirichlet_value(double x_raw, double y_raw, double z_raw);
// hex layers [k0,k1) of slab `part` of `nparts` (the split of pfem_partition_box_slabs)
inline void box_slab_layers(int nEz, int nparts, int part, int *k0, int *k1)
{
    *k0 = static_cast<int>(static_cast<int64_t>(part) * nEz / nparts);
    *k1 = static_cast<int>(static_cast<int64_t>(part + 1) * nEz / nparts);
}
// free dofs of node plane k (bc_mode 0: all six faces constrained; 1: the plane y = y0 clamped)
inline int64_t box_free_per_plane(int nNx, int nNy, int nNz, int bc_mode, int ndof, int k)
{
    if (bc_mode == 0) return (k == 0 || k == nNz - 1) ? 0 : static_cast<int64_t>(nNx - 2) * (nNy - 2) * ndof;
    return static_cast<int64_t>(nNx) * (nNy - 1) * ndof;
}

}  // namespace pfem
