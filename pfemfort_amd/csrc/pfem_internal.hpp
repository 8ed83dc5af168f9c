// pfem_internal.hpp -- declarations shared by the translation units of libpfem_amd.so
#pragma once

#include "../../include/pfem_amd.h"
#include "../../include/pfem_amd_diag.h"      // introspection / measurement entry points (not the boundary)
#include "pfem_elem.hpp"

#include <algorithm>
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
inline void box_slab_layers(int nE, int nparts, int part, int *k0, int *k1)
{
    *k0 = static_cast<int>(static_cast<int64_t>(part) * nE / nparts);
    *k1 = static_cast<int>(static_cast<int64_t>(part + 1) * nE / nparts);
}

// The axis a box is cut along when the caller leaves it open (axis < 0): the one with the most hex layers -- faces are
// then the smallest cross-section, which is what a graph partitioner (METIS_PartMeshNodal, tetrapoissonparallelimpl1.F:464)
// finds for a slender body; ties go to the highest axis (a cube is cut along z, where the reference's numbering makes
// the renumbering the identity).
inline int box_slab_axis(int nEx, int nEy, int nEz, int axis)
{
    if (axis >= 0) return axis;
    int a = 2, best = nEz;
    if (nEy > best) { a = 1; best = nEy; }
    if (nEx > best) { a = 0; }
    return a;
}

// Slab `part` of a box cut into `nparts` slabs of hex layers along `axis`, in the reference's renumbering
// (tetrapoissonparallelimpl1.F:541-612: ranks concatenated, ascending OLD node id inside a rank; free dofs counted
// scanning the NEW node order).  A rank owns the node planes above its lowest hex layer (plane 0 goes to rank 0) --
// "the lowest part among the elements that touch a node".  Inside a rank the old ids ascend lexicographically in
// (k, j, i), so the rank's free dofs form a box lo[d] .. lo[d]+cnt[d]-1 numbered x-fastest from `start`.
struct BoxOwner {
    int64_t start = 0;          // global id of the first free dof of the rank
    int lo[3] = {0, 0, 0}, cnt[3] = {0, 0, 0};
    int64_t dofs(int ndof) const { return static_cast<int64_t>(cnt[0]) * cnt[1] * cnt[2] * ndof; }
};
struct BoxSlab {
    int N[3];                   // nodes per side of the whole box
    int axis, l0, l1;           // hex layers [l0,l1) along `axis`; local node planes l0..l1
    int own_lo;                 // first OWNED node plane along `axis` (l0 + 1, or 0 for part 0)
    int bc_mode, ndof;
    BoxOwner own, prev;         // numbering of this rank's free dofs and of the rank below (owner of plane l0)
    int64_t size_global = 0;
    int Ln(int d) const { return d == axis ? l1 - l0 + 1 : N[d]; }          // local node box
    int Le(int d) const { return d == axis ? l1 - l0 : N[d] - 1; }          // local hex box
    int64_t nNode() const { return static_cast<int64_t>(Ln(0)) * Ln(1) * Ln(2); }
    int64_t nElem() const { return 6LL * Le(0) * Le(1) * Le(2); }
};

// free node range [lo, hi] of axis d (bc_mode 0: all six faces constrained; 1: the plane y = y0 clamped)
inline void box_free_range(const int N[3], int bc_mode, int d, int *lo, int *hi)
{
    if (bc_mode == 0) { *lo = 1; *hi = N[d] - 2; return; }
    *lo = d == 1 ? 1 : 0;
    *hi = N[d] - 1;
}

inline BoxOwner box_owner(const int N[3], int bc_mode, int axis, int nparts, int part, int64_t start)
{
    BoxOwner o;
    o.start = start;
    for (int d = 0; d < 3; ++d) {
        int lo, hi;
        box_free_range(N, bc_mode, d, &lo, &hi);
        if (d == axis) {
            int l0, l1;
            box_slab_layers(N[d] - 1, nparts, part, &l0, &l1);
            lo = std::max(lo, part == 0 ? 0 : l0 + 1);
            hi = std::min(hi, l1);
        }
        o.lo[d] = lo;
        o.cnt[d] = std::max(0, hi - lo + 1);
    }
    return o;
}

inline int box_slab(int nEx, int nEy, int nEz, int bc_mode, int ndof, int axis, int nparts, int part, BoxSlab *out)
{
    if (nEx < 1 || nEy < 1 || nEz < 1 || ndof < 1 || nparts < 1 || part < 0 || part >= nparts || axis > 2 ||
        (bc_mode != 0 && bc_mode != 1))
        return PFEM_ERR_ARG;
    BoxSlab b;
    b.N[0] = nEx + 1; b.N[1] = nEy + 1; b.N[2] = nEz + 1;
    b.axis = box_slab_axis(nEx, nEy, nEz, axis);
    if (nparts > b.N[b.axis] - 1) return PFEM_ERR_ARG;        // every slab needs a hex layer
    b.bc_mode = bc_mode;
    b.ndof = ndof;
    box_slab_layers(b.N[b.axis] - 1, nparts, part, &b.l0, &b.l1);
    b.own_lo = part == 0 ? 0 : b.l0 + 1;
    int64_t start = 0;
    for (int q = 0; q < nparts; ++q) {
        const BoxOwner o = box_owner(b.N, bc_mode, b.axis, nparts, q, start);
        if (q == part) b.own = o;
        if (q + 1 == part) b.prev = o;
        start += o.dofs(ndof);
    }
    b.size_global = start;
    *out = b;
    return PFEM_OK;
}

}  // namespace pfem
