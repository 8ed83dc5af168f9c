// pfem_host.cpp -- host side of the C ABI: per-element compat entry points and the
// integer bookkeeping of the PFEMFort drivers (mesh generation, Dirichlet/DOF
// numbering, partition renumbering).  No device code here; these functions work
// without a GPU and are exercised by the CPU test-suite against the oracle.
#include "pfem_internal.hpp"

#ifdef _OPENMP
#include <omp.h>
#endif
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace pfem;

#if defined(_OPENMP) && defined(__clang__)       // (the product build: hipcc / LLVM libomp; g++ builds of this file -- the sanitizer test -- have no kmp_* API)
// The bookkeeping loops below run as OpenMP loops; the GPU work that follows them is often launch-bound (a small problem's CG
// iterations).  Idle OpenMP workers that spin before they sleep take the cores the HIP runtime's launch path needs: measured
// 3-5x on the iterations of a 10^6-dof problem for ~0.2 s after a parallel region (profiles/LAB_NOTES.md).  The workers of
// THIS library's regions sleep at once: kmp_set_blocktime(0) on the calling thread around the region -- an API of the
// library's own runtime (LLVM libomp, what hipcc -fopenmp links).  Nothing touches the process environment: round 4 set
// OMP_WAIT_POLICY / KMP_BLOCKTIME from a static initializer, which also put every other OpenMP user of the process -- the
// oracle's libgomp in bench.py's cpu_baseline leg, 2.1x slower for it -- on a passive policy, and raced with getenv.
namespace {
struct OmpQuiet {
    int old;
    OmpQuiet() : old(kmp_get_blocktime()) { kmp_set_blocktime(0); }
    ~OmpQuiet() { kmp_set_blocktime(old); }
};
}  // namespace
#define PFEM_OMP_CAT2(a, b) a##b
#define PFEM_OMP_CAT(a, b) PFEM_OMP_CAT2(a, b)
#define PFEM_OMP_QUIET() OmpQuiet PFEM_OMP_CAT(omp_quiet_, __LINE__)
#else
#define PFEM_OMP_QUIET() ((void)0)
#endif

// ---------------------------------------------------------------------------
// 1. per-element compat surface
// ---------------------------------------------------------------------------
extern "C" int pfem_poisson_tria_ke(const double xNode[3], const double yNode[3],
                                    const double *elemData, const double *timeData,
                                    const double valC[3], double K[9], double F[3])
{
    if (!xNode || !yNode || !elemData || !timeData || !valC || !K || !F) return PFEM_ERR_ARG;
    return poisson_tria(xNode, yNode, elemData[0], elemData[1], timeData[1], valC, K, F)
               ? PFEM_OK : PFEM_ERR_NEG_JAC;
}

extern "C" int pfem_poisson_tet_ke(const double xNode[4], const double yNode[4],
                                   const double zNode[4], const double *elemData,
                                   const double *timeData, const double valC[4], double K[16],
                                   double F[4])
{
    if (!xNode || !yNode || !zNode || !elemData || !timeData || !valC || !K || !F) return PFEM_ERR_ARG;
    return poisson_tet(xNode, yNode, zNode, elemData[0], elemData[1], elemData[2], timeData[1],
                       valC, K, F) ? PFEM_OK : PFEM_ERR_NEG_JAC;
}

extern "C" int pfem_elast_tet_ke(const double xNode[4], const double yNode[4],
                                 const double zNode[4], const double *elemData,
                                 const double *timeData, const double valC[12], double K[144],
                                 double F[12])
{
    (void)timeData; (void)valC;  // strain/stress from valC never reach Ke/Fe (:327-355)
    if (!xNode || !yNode || !zNode || !elemData || !K || !F) return PFEM_ERR_ARG;
    const double bf[3] = {elemData[3], elemData[4], elemData[5]};
    return elast_tet(xNode, yNode, zNode, elemData[0], elemData[1], bf, K, F) ? PFEM_OK
                                                                            : PFEM_ERR_NEG_JAC;
}

extern "C" int pfem_elast_tria_ke(const double xNode[3], const double yNode[3], const double *elemData,
                                  const double *timeData, const double valC[6], double K[36], double F[6])
{
    (void)timeData; (void)valC;  // strain/stress from valC never reach Ke/Fe (elementutilitieselasticity2D.F:96-117)
    if (!xNode || !yNode || !elemData || !K || !F) return PFEM_ERR_ARG;
    const double bf[2] = {elemData[3], elemData[4]};
    return elast_tria(xNode, yNode, elemData[0], elemData[1], elemData[2], bf, K, F) ? PFEM_OK : PFEM_ERR_NEG_JAC;
}

// ---------------------------------------------------------------------------
// 2. structured box mesh (genTetra.cpp)
// ---------------------------------------------------------------------------
namespace {

// The solver only ever sees what it reads back from the "%.8f" text files
// (genTetra.cpp:187-189, 514-516), so coordinates and BC values go through the
// same decimal round trip.
double text_round8(double v)
{
    char buf[64];
    std::snprintf(buf, sizeof buf, "%.8f", v);
    return std::strtod(buf, nullptr);
}

// xx = x0; repeat: use xx; xx += dx   (genTetra.cpp:194-216)
std::vector<double> axis_accumulate(double a0, double a1, int nE)
{
    std::vector<double> t(static_cast<size_t>(nE) + 1);
    const double d = (a1 - a0) / nE;
    double v = a0;
    for (int i = 0; i <= nE; ++i) { t[i] = v; v += d; }
    return t;
}

}  // namespace

pfem::BoxAxes pfem::box_axes(double x0, double x1, int nEx, double y0, double y1, int nEy, double z0, double z1, int nEz)
{
    BoxAxes a;
    a.raw[0] = axis_accumulate(x0, x1, nEx);
    a.raw[1] = axis_accumulate(y0, y1, nEy);
    a.raw[2] = axis_accumulate(z0, z1, nEz);
    for (int d = 0; d < 3; ++d) {
        a.rounded[d].resize(a.raw[d].size());
        for (size_t i = 0; i < a.raw[d].size(); ++i) a.rounded[d][i] = text_round8(a.raw[d][i]);
    }
    return a;
}

double pfem::box_dirichlet_value(double x_raw, double y_raw, double z_raw)
{
    // vtkPoints stores float; GetPoint widens back (genTetra.cpp:512-520)
    const double cx = static_cast<double>(static_cast<float>(x_raw));
    const double cy = static_cast<double>(static_cast<float>(y_raw));
    const double cz = static_cast<double>(static_cast<float>(z_raw));
    return text_round8(cx * cx + cy * cy + cz * cz);
}

// Sizes of slab `part` of `nparts` slabs of the generated box, cut along `axis` (0 x, 1 y, 2 z, -1: the axis with the
// most hex layers), in the reference's renumbering (tetrapoissonparallelimpl1.F:541-612: ranks concatenated, ascending
// old id inside a rank; rank r owns the node planes above its lowest hex layer -- plane 0 goes to rank 0; for z-slabs
// that renumbering is the identity): closed forms, so that a rank can create its solver and generate its share of the
// mesh on the device without ever holding the whole grid.
extern "C" int pfem_box_slab_sizes_axis(int nEx, int nEy, int nEz, int bc_mode, int ndof, int axis, int nparts, int part,
                                        int64_t *size_global, int64_t *row_start, int64_t *size_local,
                                        int64_t *nNode_local, int64_t *nElem_local, int *axis_used, int *layer0, int *layer1)
{
    BoxSlab b;
    const int rc = box_slab(nEx, nEy, nEz, bc_mode, ndof, axis, nparts, part, &b);
    if (rc != PFEM_OK) return rc;
    if (size_global) *size_global = b.size_global;
    if (row_start) *row_start = b.own.start;
    if (size_local) *size_local = b.own.dofs(ndof);
    if (nNode_local) *nNode_local = b.nNode();
    if (nElem_local) *nElem_local = b.nElem();
    if (axis_used) *axis_used = b.axis;
    if (layer0) *layer0 = b.l0;
    if (layer1) *layer1 = b.l1;
    return PFEM_OK;
}

extern "C" int pfem_box_slab_sizes(int nEx, int nEy, int nEz, int bc_mode, int ndof, int nparts, int part,
                                   int64_t *size_global, int64_t *row_start, int64_t *size_local,
                                   int64_t *nNode_local, int64_t *nElem_local)
{
    return pfem_box_slab_sizes_axis(nEx, nEy, nEz, bc_mode, ndof, 2, nparts, part, size_global, row_start, size_local,
                                    nNode_local, nElem_local, nullptr, nullptr, nullptr);
}

extern "C" int pfem_gen_box_tets(double x0, double x1, int nEx, double y0, double y1, int nEy,
                                 double z0, double z1, int nEz, int kz0, int kz1, int bc_mode,
                                 int ndof, double *xyz, int32_t *conn, int64_t *nDBC,
                                 int32_t *bc_node, int32_t *bc_dof, double *bc_val)
{
    if (nEx < 1 || nEy < 1 || nEz < 1 || ndof < 1 || !nDBC) return PFEM_ERR_ARG;
    if (kz0 < 0 || kz1 > nEz || kz0 > kz1) return PFEM_ERR_ARG;
    const int nNx = nEx + 1, nNy = nEy + 1, nNz = nEz + 1;
    const int64_t plane = static_cast<int64_t>(nNx) * nNy;
    const int64_t nNode = plane * nNz;
    if (nNode > INT32_MAX) return PFEM_ERR_ARG;
    const std::vector<double> xs = axis_accumulate(x0, x1, nEx), ys = axis_accumulate(y0, y1, nEy),
                              zs = axis_accumulate(z0, z1, nEz);

    if (xyz) {
        std::vector<double> xr(nNx), yr(nNy), zr(nNz);
        for (int i = 0; i < nNx; ++i) xr[i] = text_round8(xs[i]);
        for (int j = 0; j < nNy; ++j) yr[j] = text_round8(ys[j]);
        for (int k = 0; k < nNz; ++k) zr[k] = text_round8(zs[k]);
        double *X = xyz, *Y = xyz + nNode, *Z = xyz + 2 * nNode;
    PFEM_OMP_QUIET();
#pragma omp parallel for schedule(static)
        for (int k = 0; k < nNz; ++k)
            for (int j = 0; j < nNy; ++j) {
                const int64_t base = plane * k + static_cast<int64_t>(nNx) * j;
                for (int i = 0; i < nNx; ++i) {
                    X[base + i] = xr[i];
                    Y[base + i] = yr[j];
                    Z[base + i] = zr[k];
                }
            }
    }

    if (conn) {
        // hex corner ids pts[0..7] and the fixed 6-tet split of genTetra.cpp:263-322
        static const int split[6][4] = {{0, 1, 3, 5}, {0, 3, 2, 5}, {2, 3, 7, 5},
                                        {4, 6, 7, 2}, {4, 7, 5, 2}, {0, 4, 5, 2}};
        const int64_t nElem = 6LL * nEx * nEy * (kz1 - kz0);
    PFEM_OMP_QUIET();
#pragma omp parallel for schedule(static)
        for (int k = kz0; k < kz1; ++k)
            for (int j = 0; j < nEy; ++j)
                for (int i = 0; i < nEx; ++i) {
                    const int64_t hex = (static_cast<int64_t>(k - kz0) * nEy + j) * nEx + i;
                    const int64_t lo = plane * k + static_cast<int64_t>(nNx) * j + i;
                    const int64_t c[8] = {lo, lo + 1, lo + nNx, lo + nNx + 1,
                                          lo + plane, lo + plane + 1, lo + plane + nNx,
                                          lo + plane + nNx + 1};
                    for (int t = 0; t < 6; ++t)
                        for (int a = 0; a < 4; ++a)
                            conn[a * nElem + 6 * hex + t] = static_cast<int32_t>(c[split[t][a]]);
                }
    }

    // Dirichlet list: ascending unique node ids (sort+unique, genTetra.cpp:505-506)
    int64_t cnt = 0;
    for (int k = 0; k < nNz; ++k)
        for (int j = 0; j < nNy; ++j) {
            const bool edge_row = (k == 0 || k == nNz - 1 || j == 0 || j == nNy - 1);
            for (int i = 0; i < nNx; ++i) {
                const bool on = bc_mode == 0 ? (edge_row || i == 0 || i == nNx - 1) : (j == 0);
                if (!on) continue;
                if (bc_node) {
                    const int64_t node = plane * k + static_cast<int64_t>(nNx) * j + i;
                    double val = 0.0;
                    if (bc_mode == 0) {
                        // vtkPoints stores float; GetPoint widens back (genTetra.cpp:512-520)
                        const double cx = static_cast<double>(static_cast<float>(xs[i]));
                        const double cy = static_cast<double>(static_cast<float>(ys[j]));
                        const double cz = static_cast<double>(static_cast<float>(zs[k]));
                        val = text_round8(cx * cx + cy * cy + cz * cz);
                    }
                    for (int d = 0; d < ndof; ++d) {
                        bc_node[cnt + d] = static_cast<int32_t>(node);
                        bc_dof[cnt + d] = d;
                        bc_val[cnt + d] = val;
                    }
                }
                cnt += ndof;
            }
        }
    *nDBC = cnt;
    return PFEM_OK;
}

extern "C" int pfem_partition_box_slabs_axis(int nEx, int nEy, int nEz, int axis, int nParts,
                                             int32_t *elem_proc_id, int32_t *node_proc_id)
{
    if (nEx < 1 || nEy < 1 || nEz < 1 || nParts < 1 || axis > 2) return PFEM_ERR_ARG;
    const int a = box_slab_axis(nEx, nEy, nEz, axis);
    const int E[3] = {nEx, nEy, nEz};
    if (nParts > E[a]) return PFEM_ERR_ARG;
    std::vector<int32_t> layer_part(E[a]);
    for (int p = 0; p < nParts; ++p) {
        int lo, hi;
        box_slab_layers(E[a], nParts, p, &lo, &hi);
        for (int k = lo; k < hi; ++k) layer_part[k] = p;
    }
    if (elem_proc_id) {
    PFEM_OMP_QUIET();
#pragma omp parallel for schedule(static)
        for (int k = 0; k < nEz; ++k)
            for (int j = 0; j < nEy; ++j)
                for (int i = 0; i < nEx; ++i) {
                    const int c[3] = {i, j, k};
                    const int64_t hex = (static_cast<int64_t>(k) * nEy + j) * nEx + i;
                    std::fill(elem_proc_id + 6 * hex, elem_proc_id + 6 * hex + 6, layer_part[c[a]]);
                }
    }
    if (node_proc_id) {
        // node plane c is touched by hex layers c-1 and c: the lowest part wins
    PFEM_OMP_QUIET();
#pragma omp parallel for schedule(static)
        for (int k = 0; k <= nEz; ++k)
            for (int j = 0; j <= nEy; ++j)
                for (int i = 0; i <= nEx; ++i) {
                    const int c[3] = {i, j, k};
                    node_proc_id[(static_cast<int64_t>(k) * (nEy + 1) + j) * (nEx + 1) + i] = layer_part[c[a] > 0 ? c[a] - 1 : 0];
                }
    }
    return PFEM_OK;
}

extern "C" int pfem_partition_box_slabs(int nEx, int nEy, int nEz, int nParts,
                                        int32_t *elem_proc_id, int32_t *node_proc_id)
{
    return pfem_partition_box_slabs_axis(nEx, nEy, nEz, 2, nParts, elem_proc_id, node_proc_id);
}

// Recursive coordinate bisection: a geometric stand-in for METIS_PartMeshNodal (:464) on ANY mesh with coordinates
// (METIS itself is a third-party library absent here; its output is taken through the file hook when available).
// Elements are split by the median of their centroids along the longest axis of the current box, parts sized in
// proportion (nparts need not be a power of two); a node goes to the lowest part among the elements that touch it, so
// every owner assembles at least one element at its nodes.  Deterministic: ties are broken by element id.
namespace {
void rcb_split(std::vector<int64_t> &ids, int64_t lo, int64_t hi, int p0, int np, const double *cx, const double *cy,
               const double *cz, int ndim, int32_t *epart)
{
    if (np == 1) {
        for (int64_t i = lo; i < hi; ++i) epart[ids[i]] = p0;
        return;
    }
    const double *c[3] = {cx, cy, cz};
    int axis = 0;
    double best = -1.0;
    for (int d = 0; d < ndim; ++d) {
        double mn = 1e300, mx = -1e300;
        for (int64_t i = lo; i < hi; ++i) { mn = std::min(mn, c[d][ids[i]]); mx = std::max(mx, c[d][ids[i]]); }
        if (mx - mn > best) { best = mx - mn; axis = d; }
    }
    const int npl = np / 2;
    const int64_t mid = lo + (hi - lo) * npl / np;
    const double *ca = c[axis];
    std::nth_element(ids.begin() + lo, ids.begin() + mid, ids.begin() + hi, [ca](int64_t a, int64_t b) {
        return ca[a] < ca[b] || (ca[a] == ca[b] && a < b);
    });
    rcb_split(ids, lo, mid, p0, npl, cx, cy, cz, ndim, epart);
    rcb_split(ids, mid, hi, p0 + npl, np - npl, cx, cy, cz, ndim, epart);
}
}  // namespace

extern "C" int pfem_partition_rcb(int64_t nNode, int ndim, const double *xyz, int64_t nElem, int npElem, const int32_t *conn,
                                  int nParts, int32_t *elem_proc_id, int32_t *node_proc_id)
{
    if (nNode < 1 || (ndim != 2 && ndim != 3) || !xyz || nElem < 0 || npElem < 1 || (nElem && !conn) || nParts < 1 ||
        !elem_proc_id || !node_proc_id)
        return PFEM_ERR_ARG;
    std::vector<double> cen(static_cast<size_t>(3) * std::max<int64_t>(nElem, 1), 0.0);
    for (int64_t e = 0; e < nElem; ++e)
        for (int a = 0; a < npElem; ++a) {
            const int32_t n = conn[static_cast<int64_t>(a) * nElem + e];
            if (n < 0 || n >= nNode) return PFEM_ERR_ARG;
            for (int d = 0; d < ndim; ++d) cen[static_cast<size_t>(d) * nElem + e] += xyz[static_cast<int64_t>(d) * nNode + n] / npElem;
        }
    std::vector<int64_t> ids(static_cast<size_t>(nElem));
    for (int64_t e = 0; e < nElem; ++e) ids[e] = e;
    if (nElem > 0) rcb_split(ids, 0, nElem, 0, std::min<int64_t>(nParts, nElem), cen.data(), cen.data() + nElem, cen.data() + 2 * nElem, ndim, elem_proc_id);
    for (int64_t n = 0; n < nNode; ++n) node_proc_id[n] = nParts;           // lowest adjacent part wins
    for (int64_t e = 0; e < nElem; ++e)
        for (int a = 0; a < npElem; ++a) {
            int32_t &p = node_proc_id[conn[static_cast<int64_t>(a) * nElem + e]];
            p = std::min(p, elem_proc_id[e]);
        }
    for (int64_t n = 0; n < nNode; ++n)
        if (node_proc_id[n] == nParts) node_proc_id[n] = 0;                  // a node no element touches
    return PFEM_OK;
}

// ---------------------------------------------------------------------------
// 3. Dirichlet bookkeeping and (re)numbering (tetrapoissonparallelimpl1.F)
// ---------------------------------------------------------------------------
extern "C" int pfem_dof_numbering(int64_t nNode, int ndof, int64_t nDBC, const int32_t *dbc_node,
                                  const int32_t *dbc_dof, const double *dbc_val, int nParts,
                                  const int32_t *node_proc_id, int32_t *node_map_get_old,
                                  int32_t *node_map_get_new, int32_t *NodeDofArrayNew,
                                  double *solnApplied, int64_t *node_start, int64_t *node_end,
                                  int64_t *row_start, int64_t *row_end, int64_t *size_global)
{
    if (nNode < 1 || ndof < 1 || nParts < 1 || !node_map_get_old || !node_map_get_new ||
        !NodeDofArrayNew || !solnApplied || !node_start || !node_end || !row_start || !row_end ||
        !size_global || (nDBC > 0 && (!dbc_node || !dbc_dof || !dbc_val)) ||
        (nParts > 1 && !node_proc_id))
        return PFEM_ERR_ARG;
    const int64_t nd = nNode * ndof;

    // NodeTypeOld / solnApplied at OLD ids (:341-355)
    std::vector<uint8_t> constrained_old(nd, 0);
    std::fill(solnApplied, solnApplied + nd, 0.0);
    for (int64_t b = 0; b < nDBC; ++b) {
        if (dbc_node[b] < 0 || dbc_node[b] >= nNode || dbc_dof[b] < 0 || dbc_dof[b] >= ndof)
            return PFEM_ERR_ARG;
        const int64_t slot = static_cast<int64_t>(dbc_node[b]) * ndof + dbc_dof[b];
        constrained_old[slot] = 1;
        solnApplied[slot] = dbc_val[b];
    }

    if (nParts == 1) {  // :402-421
        for (int64_t n = 0; n < nNode; ++n) node_map_get_old[n] = node_map_get_new[n] = static_cast<int32_t>(n);
        node_start[0] = 0;
        node_end[0] = nNode;
    } else {
        // NEW numbering = ranks concatenated, ascending OLD id inside a rank
        // (locally_owned_nodes + MPI_Allgatherv, :541-566): a stable counting sort.
        std::vector<int64_t> first(static_cast<size_t>(nParts) + 1, 0);
        for (int64_t n = 0; n < nNode; ++n) {
            const int32_t p = node_proc_id[n];
            if (p < 0 || p >= nParts) return PFEM_ERR_ARG;
            ++first[p + 1];
        }
        for (int p = 0; p < nParts; ++p) first[p + 1] += first[p];
        for (int p = 0; p < nParts; ++p) { node_start[p] = first[p]; node_end[p] = first[p + 1]; }
        std::vector<int64_t> cursor(first.begin(), first.end() - 1);
        for (int64_t n = 0; n < nNode; ++n) {
            const int64_t nn = cursor[node_proc_id[n]]++;
            node_map_get_old[nn] = static_cast<int32_t>(n);
            node_map_get_new[n] = static_cast<int32_t>(nn);     // :588-595
        }
        // BC values re-entered at NEW ids; the OLD-id entries are not cleared (:668-677)
        for (int64_t b = 0; b < nDBC; ++b)
            solnApplied[static_cast<int64_t>(node_map_get_new[dbc_node[b]]) * ndof + dbc_dof[b]] = dbc_val[b];
    }

    // free dofs numbered in NEW node order (:357-367 / :601-612)
    int64_t next = 0;
    for (int64_t nn = 0; nn < nNode; ++nn) {
        const int64_t old = node_map_get_old[nn];
        for (int d = 0; d < ndof; ++d)
            NodeDofArrayNew[nn * ndof + d] = constrained_old[old * ndof + d] ? -1 : static_cast<int32_t>(next++);
    }
    if (next > INT32_MAX) return PFEM_ERR_ARG;
    *size_global = next;

    // contiguous row block per rank (:622-636); an empty block sits at the previous end
    int64_t prev_end = 0;
    for (int p = 0; p < nParts; ++p) {
        int64_t lo = -1, hi = -1;
        for (int64_t nn = node_start[p]; nn < node_end[p] && lo < 0; ++nn)
            for (int d = 0; d < ndof; ++d)
                if (NodeDofArrayNew[nn * ndof + d] >= 0) { lo = NodeDofArrayNew[nn * ndof + d]; break; }
        for (int64_t nn = node_end[p] - 1; nn >= node_start[p] && hi < 0; --nn)
            for (int d = ndof - 1; d >= 0; --d)
                if (NodeDofArrayNew[nn * ndof + d] >= 0) { hi = NodeDofArrayNew[nn * ndof + d] + 1; break; }
        if (lo < 0) { lo = prev_end; hi = prev_end; }
        row_start[p] = lo;
        row_end[p] = hi;
        prev_end = hi;
    }
    return PFEM_OK;
}

extern "C" int pfem_renumber_mesh(int64_t nNode, int ndim, int64_t nElem, int npElem, const int32_t *conn_old,
                                  const double *xyz_old, const int32_t *node_map_get_new,
                                  const int32_t *node_map_get_old, int32_t *conn_new, double *xyz_new)
{
    if (nNode < 0 || nElem < 0 || ndim < 1 || npElem < 1 || !node_map_get_new || !node_map_get_old) return PFEM_ERR_ARG;
    if ((nElem > 0 && (!conn_old || !conn_new)) || (nNode > 0 && (!xyz_old || !xyz_new))) return PFEM_ERR_ARG;
    int bad = 0;
    const int64_t nc = static_cast<int64_t>(npElem) * nElem;
    PFEM_OMP_QUIET();
#pragma omp parallel for schedule(static) reduction(| : bad) if (nc > 100000)
    for (int64_t i = 0; i < nc; ++i) {
        const int32_t n = conn_old[i];
        if (n < 0 || n >= nNode) { bad = 1; conn_new[i] = -1; } else conn_new[i] = node_map_get_new[n];
    }
    for (int d = 0; d < ndim; ++d) {
        const double *src = xyz_old + static_cast<int64_t>(d) * nNode;
        double *dst = xyz_new + static_cast<int64_t>(d) * nNode;
    PFEM_OMP_QUIET();
#pragma omp parallel for schedule(static) reduction(| : bad) if (nNode > 100000)
        for (int64_t i = 0; i < nNode; ++i) {
            const int32_t o = node_map_get_old[i];
            if (o < 0 || o >= nNode) { bad = 1; dst[i] = 0.0; } else dst[i] = src[o];
        }
    }
    if (bad) return PFEM_ERR_ARG;              // a node id out of range
    return PFEM_OK;
}

extern "C" int pfem_elem_dof_array(int64_t nElem, int npElem, int ndof, const int32_t *conn_new,
                                   const int32_t *NodeDofArrayNew, int32_t *edof)
{
    if (nElem < 0 || npElem < 1 || ndof < 1 || !conn_new || !NodeDofArrayNew || !edof) return PFEM_ERR_ARG;
    for (int i = 0; i < npElem; ++i)
        for (int d = 0; d < ndof; ++d) {
            const int32_t *c = conn_new + static_cast<int64_t>(i) * nElem;
            int32_t *o = edof + static_cast<int64_t>(i * ndof + d) * nElem;
    PFEM_OMP_QUIET();
#pragma omp parallel for schedule(static)
            for (int64_t e = 0; e < nElem; ++e) o[e] = NodeDofArrayNew[static_cast<int64_t>(c[e]) * ndof + d];
        }
    return PFEM_OK;
}

extern "C" int pfem_assy_for_soln(int64_t nNode, int ndof, const int32_t *NodeDofArrayNew,
                                  int32_t *assyForSoln)
{
    if (nNode < 0 || ndof < 1 || !NodeDofArrayNew || !assyForSoln) return PFEM_ERR_ARG;
    int64_t k = 0;
    for (int64_t s = 0; s < nNode * ndof; ++s)
        if (NodeDofArrayNew[s] >= 0) assyForSoln[k++] = static_cast<int32_t>(s);
    return PFEM_OK;
}

// ---------------------------------------------------------------------------
// 4. local numbering helper for the sub-assembled multi-rank layout
// ---------------------------------------------------------------------------
extern "C" int pfem_find_ghosts(int64_t count, const int32_t *edof, int64_t row_start, int64_t n_owned,
                                int64_t *n_ghost, int64_t *ghost_gid)
{
    if (count < 0 || (count && !edof) || row_start < 0 || n_owned < 0 || !n_ghost) return PFEM_ERR_ARG;
    std::vector<int64_t> g;
    const int64_t lo = row_start, hi = row_start + n_owned;
    for (int64_t i = 0; i < count; ++i) {
        const int64_t d = edof[i];
        if (d >= 0 && (d < lo || d >= hi)) g.push_back(d);
    }
    std::sort(g.begin(), g.end());
    g.erase(std::unique(g.begin(), g.end()), g.end());
    *n_ghost = static_cast<int64_t>(g.size());
    if (ghost_gid) std::copy(g.begin(), g.end(), ghost_gid);
    return PFEM_OK;
}

// Neighbour plan (include/pfem_amd.h, section 5).  local(r) = owned block of r + ghost list of r; rank q is a peer of
// `rank` when local(rank) and local(q) intersect, and the shared list is that intersection, ascending:
//   { g in ghost(q)    : g owned by rank or g in ghost(rank) }  +  { g in ghost(rank) : g owned by q }
// (the owned blocks are disjoint, so these are all the cases).  Both sides of a pair arrive at the same list.
// Replaces the VecScatter setup PETSc derives from the matrix's off-diagonal columns (solverpetsc.F:447-450).
extern "C" int pfem_neighbour_plan(int nranks, int rank, const int64_t *row_start, const int64_t *row_end,
                                   const int64_t *ghost_off, const int64_t *ghost_gid, int *n_peers, int64_t *n_total,
                                   int *peers, int64_t *peer_off, int64_t *shared_gid)
{
    if (nranks < 1 || rank < 0 || rank >= nranks || !row_start || !row_end || !ghost_off || !n_peers || !n_total ||
        (ghost_off[nranks] > 0 && !ghost_gid))
        return PFEM_ERR_ARG;
    for (int r = 0; r < nranks; ++r) {
        if (row_start[r] > row_end[r] || ghost_off[r] > ghost_off[r + 1]) return PFEM_ERR_ARG;
        for (int64_t k = ghost_off[r] + 1; k < ghost_off[r + 1]; ++k)
            if (ghost_gid[k - 1] >= ghost_gid[k]) return PFEM_ERR_ARG;           // ascending, unique
    }
    const int64_t *mine = ghost_gid + ghost_off[rank], *mine_end = ghost_gid + ghost_off[rank + 1];
    const int64_t lo = row_start[rank], hi = row_end[rank];
    int np = 0;
    int64_t total = 0;
    if (peer_off) peer_off[0] = 0;
    std::vector<int64_t> sh;
    for (int q = 0; q < nranks; ++q) {
        if (q == rank) continue;
        sh.clear();
        for (int64_t k = ghost_off[q]; k < ghost_off[q + 1]; ++k) {
            const int64_t g = ghost_gid[k];
            if ((g >= lo && g < hi) || std::binary_search(mine, mine_end, g)) sh.push_back(g);
        }
        for (const int64_t *g = mine; g != mine_end; ++g)
            if (*g >= row_start[q] && *g < row_end[q]) sh.push_back(*g);
        if (sh.empty()) continue;
        std::sort(sh.begin(), sh.end());        // the two parts are disjoint (a ghost of q is not owned by q)
        if (peers) {
            peers[np] = q;
            std::copy(sh.begin(), sh.end(), shared_gid + total);
            peer_off[np + 1] = total + static_cast<int64_t>(sh.size());
        }
        total += static_cast<int64_t>(sh.size());
        ++np;
    }
    *n_peers = np;
    *n_total = total;
    return PFEM_OK;
}

// ---------------------------------------------------------------------------
// 5. output step (after the path): legacy ASCII VTK, byte-compatible with writervtk.F:33-201
// ---------------------------------------------------------------------------
namespace {
// "%12.6f" of v into dst (at least 400 bytes), returns the byte count.  Values of ordinary size take a short path that
// is byte-identical to printf: r = v*1e6 rounded to an integer, accepted only when the exact residual v*1e6 - r (one
// fma) is clearly inside (-1/2, 1/2) -- near a tie, for large or non-finite values, snprintf decides.
inline int fmt_f12_6(char *dst, double v)
{
    const double a = std::fabs(v);
    if (a < 9999.0) {
        const double r = std::nearbyint(a * 1e6);
        const double res = std::fma(a, 1e6, -r);
        if (std::fabs(res) < 0.4999999) {
            uint64_t q = static_cast<uint64_t>(r);
            char tmp[12];
            for (int k = 11; k >= 0; --k) {
                if (k == 5) { tmp[k] = '.'; continue; }
                tmp[k] = static_cast<char>('0' + q % 10);
                q /= 10;
            }
            int lead = 0;                                   // blanks instead of leading zeros, one digit before the point
            while (lead < 4 && tmp[lead] == '0') { tmp[lead] = ' '; ++lead; }
            if (std::signbit(v)) tmp[lead - 1 >= 0 ? lead - 1 : 0] = '-';   // a < 9999: at least one blank is left
            std::memcpy(dst, tmp, 12);
            return 12;
        }
    }
    return std::snprintf(dst, 400, "%12.6f", v);
}

// "%<width>d" for width <= 11
inline int fmt_int(char *dst, int width, int32_t v)
{
    char tmp[12];
    int n = 0;
    uint32_t u = v < 0 ? 0u - static_cast<uint32_t>(v) : static_cast<uint32_t>(v);
    do { tmp[n++] = static_cast<char>('0' + u % 10); u /= 10; } while (u);
    if (v < 0) tmp[n++] = '-';
    int out = 0;
    for (int k = n; k < width; ++k) dst[out++] = ' ';
    while (n) dst[out++] = tmp[--n];
    return out;
}

// Formats records [0,n) on all threads, block by block, and writes the blocks in order.  `fmt(dst, i)` returns the bytes
// it wrote (never more than max_rec).
template <class F>
void emit_records(std::FILE *f, int64_t n, int max_rec, F fmt)
{
    int nt = 1;
#ifdef _OPENMP
    nt = std::max(1, omp_get_max_threads());
#endif
    const int64_t block = std::max<int64_t>(64, std::min<int64_t>(4096, (1 << 20) / max_rec));   // ~1 MiB per thread
    nt = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(nt, (n + block - 1) / block)));
    std::vector<std::vector<char>> bufs(static_cast<size_t>(nt));
    std::vector<size_t> used(static_cast<size_t>(nt), 0);
    for (auto &b : bufs) b.resize(static_cast<size_t>(block) * max_rec);
    for (int64_t base = 0; base < n; base += block * nt) {
    PFEM_OMP_QUIET();
#pragma omp parallel for schedule(static, 1) num_threads(nt)
        for (int t = 0; t < nt; ++t) {
            const int64_t b = base + t * block, e = std::min(n, b + block);
            char *p = bufs[t].data();
            for (int64_t i = b; i < e; ++i) p += fmt(p, i);
            used[t] = b < e ? static_cast<size_t>(p - bufs[t].data()) : 0;
        }
        for (int t = 0; t < nt; ++t)
            if (used[t]) std::fwrite(bufs[t].data(), 1, used[t], f);
    }
}
}  // namespace

extern "C" int pfem_write_vtk(const char *path, int ndim, int64_t nElem, int64_t nNode, int npElem, int ndof,
                              const double *coords /*SoA [d*nNode+n]*/, const int32_t *conn /*SoA, 0-based*/,
                              const int32_t *elem_procid, const double *soln /*[n*ndof+d]*/)
{
    if (!path || (ndim != 2 && ndim != 3) || nElem < 0 || nNode < 0 || npElem < 3 || ndof < 1 || !coords || !conn ||
        !elem_procid || !soln)
        return PFEM_ERR_ARG;
    int cell_type;                                               // writervtk.F:85-140
    if (ndim == 2) cell_type = npElem == 3 ? 5 : (npElem == 6 ? 22 : 9);
    else cell_type = npElem == 4 ? 10 : (npElem == 6 ? 13 : 12);
    const int ncol = ndim == 2 ? (npElem == 3 ? 3 : (npElem == 6 ? 6 : 4)) : (npElem == 4 ? 4 : (npElem == 6 ? 4 : 0));
    if (ncol == 0 || ncol > npElem) return PFEM_ERR_ARG;
    std::FILE *f = std::fopen(path, "w");
    if (!f) return PFEM_ERR_ARG;
    std::fputs("# vtk DataFile Version 4.0\n", f);               // '(A)'
    std::fputs("PoissonTwoD example\n", f);
    std::fputs(" ASCII\n", f);                                   // list-directed write(1,*): leading blank
    std::fputs("DATASET UNSTRUCTURED_GRID\n", f);
    std::fprintf(f, "POINTS %10lld float\n", static_cast<long long>(nNode));     // '(A,I10,A)'
    const auto triple = [](char *p, double a, double b, double c) {             // '(F12.6,F12.6,F12.6)'
        char *q = p;
        q += fmt_f12_6(q, a);
        q += fmt_f12_6(q, b);
        q += fmt_f12_6(q, c);
        *q++ = '\n';
        return static_cast<int>(q - p);
    };
    emit_records(f, nNode, 1204, [&](char *p, int64_t n) {
        return triple(p, coords[n], coords[nNode + n], ndim == 3 ? coords[2 * nNode + n] : 0.0);
    });
    std::fprintf(f, "CELLS %10lld%10lld\n", static_cast<long long>(nElem), static_cast<long long>(nElem * (npElem + 1)));
    emit_records(f, nElem, 16 * 8, [&](char *p, int64_t e) {     // 0-based ids, npElem first
        char *q = p;
        q += fmt_int(q, 10, npElem);
        for (int a = 0; a < ncol; ++a) q += fmt_int(q, 10, conn[a * nElem + e]);
        *q++ = '\n';
        return static_cast<int>(q - p);
    });
    std::fprintf(f, "CELL_TYPES%10lld\n", static_cast<long long>(nElem));
    emit_records(f, nElem, 16, [&](char *p, int64_t) { int k = fmt_int(p, 3, cell_type); p[k] = '\n'; return k + 1; });
    std::fprintf(f, "CELL_DATA%10lld\n", static_cast<long long>(nElem));
    std::fputs("SCALARS procid int 1\nLOOKUP_TABLE default\n", f);
    emit_records(f, nElem, 16, [&](char *p, int64_t e) { int k = fmt_int(p, 3, elem_procid[e]); p[k] = '\n'; return k + 1; });
    std::fprintf(f, "POINT_DATA%10lld\n", static_cast<long long>(nNode));
    if (ndof == 1) {
        std::fputs("SCALARS solution float 1\nLOOKUP_TABLE default\n", f);
        emit_records(f, nNode, 404, [&](char *p, int64_t n) { int k = fmt_f12_6(p, soln[n]); p[k] = '\n'; return k + 1; });
    } else {
        std::fputs("VECTORS solution float\n", f);
        emit_records(f, nNode, 1204, [&](char *p, int64_t n) {
            return triple(p, soln[n * ndof], soln[n * ndof + 1], ndof == 2 ? 0.0 : soln[n * ndof + 2]);
        });
    }
    const bool ok = std::fflush(f) == 0 && !std::ferror(f);
    std::fclose(f);
    return ok ? PFEM_OK : PFEM_ERR_ARG;
}

// temp.dat of the drivers (tetrapoissonparallelimpl1.F:935-942: "ii, ind, value" per free dof; the elasticity driver
// :1031-1046 writes the value alone): `ind` NULL selects the value-only form.  Numbers as the Python mirror printed them
// before (" %11d %11d   %.16E"), which is what the fixtures of the reference's runs are compared through.
extern "C" int pfem_write_temp_dat(const char *path, int64_t n, const int64_t *ii, const int64_t *ind, const double *val)
{
    if (!path || n < 0 || (n && !val) || ((ii == nullptr) != (ind == nullptr))) return PFEM_ERR_ARG;
    std::FILE *f = std::fopen(path, "w");
    if (!f) return PFEM_ERR_ARG;
    if (ind)
        emit_records(f, n, 96, [&](char *p, int64_t i) {
            return std::snprintf(p, 96, " %11lld %11lld   %.16E\n", static_cast<long long>(ii[i]), static_cast<long long>(ind[i]), val[i]);
        });
    else
        emit_records(f, n, 64, [&](char *p, int64_t i) { return std::snprintf(p, 64, "   %.16E\n", val[i]); });
    const bool ok = std::fflush(f) == 0 && !std::ferror(f);
    std::fclose(f);
    return ok ? PFEM_OK : PFEM_ERR_ARG;
}

// ---------------------------------------------------------------------------
// 6. mesh ingest (the step before the path): whitespace-separated ASCII tables, one record per
//    line, as the drivers read them with list-directed READs (tetrapoissonparallelimpl1.F:216-355)
// ---------------------------------------------------------------------------
namespace {
inline bool is_blank(char c) { return c == ' ' || c == '\t' || c == '\r' || c == ','; }
}

namespace {
// The buffer is cut into one piece per thread at record boundaries; both passes below walk the same pieces.
struct TextPiece {
    int64_t begin = 0, end = 0, rows = 0;
    int first_tokens = 0, min_tokens = 1 << 30;
};

std::vector<TextPiece> text_pieces(const char *buf, int64_t len)
{
    int nt = 1;
#ifdef _OPENMP
    nt = std::max(1, omp_get_max_threads());
#endif
    nt = static_cast<int>(std::min<int64_t>(nt, std::max<int64_t>(1, len >> 16)));
    std::vector<TextPiece> pc(static_cast<size_t>(nt));
    for (int t = 0; t < nt; ++t) {
        int64_t b = len * t / nt;
        if (t > 0) {                                       // start after the newline that ends the record we fell into
            while (b < len && buf[b - 1] != '\n') ++b;
        }
        pc[t].begin = b;
        if (t > 0) pc[t - 1].end = b;
    }
    pc[nt - 1].end = len;
    for (int t = 0; t + 1 < nt; ++t) pc[t].end = std::max(pc[t].end, pc[t].begin);
    PFEM_OMP_QUIET();
#pragma omp parallel for schedule(static, 1)
    for (int t = 0; t < nt; ++t) {
        TextPiece &q = pc[t];
        int64_t i = q.begin;
        while (i < q.end) {
            int tokens = 0;
            while (i < q.end && buf[i] != '\n') {
                while (i < q.end && is_blank(buf[i])) ++i;
                if (i < q.end && buf[i] != '\n') {
                    ++tokens;
                    while (i < q.end && buf[i] != '\n' && !is_blank(buf[i])) ++i;
                }
            }
            if (i < q.end) ++i;   // newline
            if (tokens > 0) {
                if (q.rows == 0) q.first_tokens = tokens;
                q.min_tokens = std::min(q.min_tokens, tokens);
                ++q.rows;
            }
        }
    }
    return pc;
}

// One number.  Plain decimals with at most 15 significant digits and at most 22 fractional digits -- every value the
// mesh files hold ("%d", "%.8f") -- are converted exactly by one IEEE division (both operands are exact doubles, so the
// quotient is the correctly rounded value strtod would return); anything else (exponents, Fortran D, long mantissas,
// inf/nan) goes to strtod.
inline bool parse_number(const char *&p, const char *end, double *out)
{
    static const double p10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                   1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
    const char *t0 = p;
    bool neg = false;
    if (p < end && (*p == '-' || *p == '+')) { neg = *p == '-'; ++p; }
    uint64_t mant = 0;
    int digits = 0, frac = 0;
    bool simple = true;
    while (p < end && *p >= '0' && *p <= '9') { mant = mant * 10 + static_cast<uint64_t>(*p - '0'); if (mant || digits) ++digits; ++p; if (digits > 15) { simple = false; break; } }
    const bool int_part = p > t0 + ((t0 < end && (*t0 == '-' || *t0 == '+')) ? 1 : 0);
    bool frac_part = false;
    if (simple && p < end && *p == '.') {
        ++p;
        while (p < end && *p >= '0' && *p <= '9') {
            mant = mant * 10 + static_cast<uint64_t>(*p - '0');
            if (mant || digits) ++digits;
            ++frac; ++p; frac_part = true;
            if (digits > 15 || frac > 22) { simple = false; break; }
        }
    }
    if (simple && (int_part || frac_part) && (p >= end || *p == '\n' || is_blank(*p))) {
        const double v = static_cast<double>(mant) / p10[frac];
        *out = neg ? -v : v;
        return true;
    }
    // general token: bounded, NUL-terminated copy for strtod
    p = t0;
    char tok[64];
    int n = 0;
    while (p < end && *p != '\n' && !is_blank(*p) && n < 63) tok[n++] = *p++;
    tok[n] = 0;
    for (int k = 0; k < n; ++k) if (tok[k] == 'D' || tok[k] == 'd') tok[k] = 'e';   // Fortran exponent
    char *q = nullptr;
    *out = std::strtod(tok, &q);
    return q != tok;
}
}  // namespace

extern "C" int pfem_text_table_shape(const char *buf, int64_t len, int64_t *rows, int *cols)
{
    if (!buf || len < 0 || !rows || !cols) return PFEM_ERR_ARG;
    const std::vector<TextPiece> pc = text_pieces(buf, len);
    int64_t r = 0;
    int c0 = 0, mn = 1 << 30;
    for (const TextPiece &q : pc) {
        if (q.rows == 0) continue;
        if (r == 0) c0 = q.first_tokens;
        mn = std::min(mn, q.min_tokens);
        r += q.rows;
    }
    if (r > 0 && mn < c0) return PFEM_ERR_ARG;              // short record
    *rows = r;
    *cols = c0;
    return PFEM_OK;
}

extern "C" int pfem_text_table_parse(const char *buf, int64_t len, int64_t rows, int cols, double *out)
{
    if (!buf || len < 0 || rows < 0 || cols < 1 || !out) return PFEM_ERR_ARG;
    std::vector<TextPiece> pc = text_pieces(buf, len);
    std::vector<int64_t> first(pc.size() + 1, 0);
    for (size_t t = 0; t < pc.size(); ++t) first[t + 1] = first[t] + pc[t].rows;
    if (first.back() != rows) return PFEM_ERR_ARG;
    int bad = 0;
    const int np = static_cast<int>(pc.size());
    PFEM_OMP_QUIET();
#pragma omp parallel for schedule(static, 1) reduction(| : bad)
    for (int t = 0; t < np; ++t) {
        const char *p = buf + pc[t].begin;
        const char *end = buf + pc[t].end;
        int64_t r = first[t];
        while (p < end && !bad) {
            while (p < end && is_blank(*p)) ++p;
            if (p >= end) break;
            if (*p == '\n') { ++p; continue; }             // empty line
            for (int c = 0; c < cols; ++c) {
                while (p < end && is_blank(*p)) ++p;
                double v;
                if (p >= end || *p == '\n' || !parse_number(p, end, &v)) { bad |= 1; break; }
                out[static_cast<int64_t>(c) * rows + r] = v;
            }
            ++r;
            while (p < end && *p != '\n') ++p;              // extra columns are ignored, as a list-directed READ does
            if (p < end) ++p;
        }
    }
    return bad ? PFEM_ERR_ARG : PFEM_OK;
}
