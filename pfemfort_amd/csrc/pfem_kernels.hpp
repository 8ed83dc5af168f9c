// pfem_kernels.hpp -- gfx950 device kernels of the hot path (included once by pfem_device.hip).
//
// Data layout in HBM (all SoA, coalesced in the element / row index):
//   mesh     conn[a][e] int32, edof[i][e] int32 (LOCAL dof ids, -1 = Dirichlet),
//            xyz[d][n] f64, solnApplied[n*ndof+d] f64
//   matrix   wave-sliced CSR ("SELL-64"): rows are grouped in slices of 64 (one
//            wavefront); inside a slice entry k of lane l sits at
//            slice_off[s] + 64*k + l, so one wave-wide load of entry k is 512 B (f64
//            values) / 256 B (int32 columns) contiguous.  Columns ascending per row,
//            rows padded to the slice width with (col = own row, val = 0).
//   vectors  f64[n_local], owned rows first, ghost rows after.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "pfem_elem.hpp"

namespace pfem {

constexpr int kBlock = 256;        // 4 wavefronts
constexpr int kMaxGrid = 2048;     // 8 blocks per CU x 256 CUs; also the partial-sum capacity
constexpr int kXcds = 8;

struct MeshDev {
    int kind, npe, ndof, nsize, ndim;
    int64_t nElem, nNode;
    const int32_t *conn;
    const int32_t *edof;
    const double *xyz;
    const double *soln;
};

struct SellDev {
    int64_t n_rows;       // n_local
    int64_t n_slices;
    const int64_t *slice_off;  // [n_slices+1], in entries
    const int32_t *rowlen;     // [n_rows]
    const int32_t *cols;
    double *vals;
};

struct ElemPrm {
    double ed[6];   // elemData
    double af;      // timeData(2)
};

// ---------------------------------------------------------------------------
// wave / block reductions (wave = 64 lanes)
// ---------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// Sum over the 256 threads of the block; result valid in every thread.  `sm` has
// >= 4 doubles.  Fixed association order -> bitwise reproducible.
__device__ __forceinline__ double block_sum(double v, double *sm)
{
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) sm[w] = v;
    __syncthreads();
    return (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

// The same sum -- (w0 + w1) + (w2 + w3), the same bits -- written to *out by whichever wave finishes LAST, without a barrier:
// a wave that is done leaves, its slot goes to the next workgroup's waves instead of idling until the block's slowest wave
// arrives (the (p,Ap) partial of the CG SpMV: the WITH_DOT variant cost 12 % over the plain one, profiles/r05).  `cnt` is an
// LDS word that was zero before any wave got here (set before a barrier at the kernel's start, where the waves are together
// anyway); sm has >= 4 doubles.  LDS operations of one workgroup are served by one unit in order: the release fence orders the
// wave's partial before its ticket, the acquire fence the last ticket before the reads.
__device__ __forceinline__ void block_sum_last_wave(double v, double *sm, int *cnt, double *out)
{
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) {
        sm[threadIdx.x >> 6] = v;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        const int ticket = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (ticket == (kBlock >> 6) - 1) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const volatile double *vs = sm;
            *out = (vs[0] + vs[1]) + (vs[2] + vs[3]);
        }
    }
}

// Every block sums the same `n` per-block partials in the same order, so all blocks
// (and all runs) obtain the same bits without a separate reduction launch.
__device__ __forceinline__ double sum_partials(const double *part, int n, double *sm)
{
    double a = 0.0;
    for (int i = threadIdx.x; i < n; i += kBlock) a += part[i];
    return block_sum(a, sm);
}

// ---------------------------------------------------------------------------
// symbolic phase
// ---------------------------------------------------------------------------
constexpr uint64_t kNoKey = ~0ull;

// (row,col) key of every element-matrix entry: the INSERT_VALUES pass of
// tetrapoissonparallelimpl1.F:791-802, negative indices ignored.
__global__ void __launch_bounds__(kBlock) k_emit_keys(MeshDev m, uint64_t *keys)
{
    const int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (e >= m.nElem) return;
    int32_t dof[12];
    for (int i = 0; i < m.nsize; ++i) dof[i] = m.edof[i * m.nElem + e];
    for (int i = 0; i < m.nsize; ++i)
        for (int j = 0; j < m.nsize; ++j) {
            const bool ok = dof[i] >= 0 && dof[j] >= 0;
            keys[(static_cast<int64_t>(i) * m.nsize + j) * m.nElem + e] =
                ok ? (static_cast<uint64_t>(static_cast<uint32_t>(dof[i])) << 32) | static_cast<uint32_t>(dof[j])
                   : kNoKey;
        }
}

// the same for elements [e0, e0 + ne): meshes whose nsize^2 * nElem keys exceed what one sort call can index are
// processed in element ranges (pfem_pattern_build)
__global__ void __launch_bounds__(kBlock) k_emit_keys_range(MeshDev m, int64_t e0, int64_t ne, uint64_t *keys)
{
    const int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (t >= ne) return;
    const int64_t e = e0 + t;
    int32_t dof[12];
    for (int i = 0; i < m.nsize; ++i) dof[i] = m.edof[i * m.nElem + e];
    for (int i = 0; i < m.nsize; ++i)
        for (int j = 0; j < m.nsize; ++j) {
            const bool ok = dof[i] >= 0 && dof[j] >= 0;
            keys[(static_cast<int64_t>(i) * m.nsize + j) * ne + t] =
                ok ? (static_cast<uint64_t>(static_cast<uint32_t>(dof[i])) << 32) | static_cast<uint32_t>(dof[j])
                   : kNoKey;
        }
}

// rowptr from the sorted unique keys (rows without entries get empty ranges)
__global__ void __launch_bounds__(kBlock) k_row_bounds(const uint64_t *keys, int64_t nnz, int64_t n_rows,
                                                        int64_t *rowptr)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (nnz == 0) {
        if (i <= n_rows) rowptr[i] = 0;
        return;
    }
    if (i >= nnz) return;
    const int64_t r = static_cast<int64_t>(keys[i] >> 32);
    const int64_t rp = i ? static_cast<int64_t>(keys[i - 1] >> 32) : -1;
    for (int64_t rr = rp + 1; rr <= r; ++rr) rowptr[rr] = i;
    if (i == nnz - 1)
        for (int64_t rr = r + 1; rr <= n_rows; ++rr) rowptr[rr] = nnz;
}

// per-slice padded size in entries (64 * max row length); one thread per slice
__global__ void __launch_bounds__(kBlock) k_slice_sizes(const int64_t *rowptr, int64_t n_rows, int64_t n_slices,
                                                         int32_t *rowlen, int64_t *slice_entries)
{
    const int64_t r = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    int len = 0;
    if (r < n_rows) {
        len = static_cast<int>(rowptr[r + 1] - rowptr[r]);
        rowlen[r] = len;
    }
    // wave == slice: max over the 64 lanes
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) len = max(len, __shfl_xor(len, o, 64));
    const int64_t s = r >> 6;
    if ((threadIdx.x & 63) == 0 && s < n_slices) slice_entries[s] = 64LL * len;
    if (r == 0) slice_entries[n_slices] = 0;
}

__global__ void __launch_bounds__(kBlock) k_fill_sell(const uint64_t *keys, const int64_t *rowptr, int64_t n_rows,
                                                       int64_t n_slices, const int64_t *slice_off, int32_t *cols)
{
    const int64_t r = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    const int64_t s = r >> 6;
    if (s >= n_slices) return;
    const int64_t off = slice_off[s] + (r & 63);
    const int width = static_cast<int>((slice_off[s + 1] - slice_off[s]) >> 6);
    int64_t p0 = 0;
    int len = 0;
    if (r < n_rows) { p0 = rowptr[r]; len = static_cast<int>(rowptr[r + 1] - p0); }
    const int32_t pad = r < n_rows ? static_cast<int32_t>(r) : 0;
    for (int k = 0; k < width; ++k)
        cols[off + 64LL * k] = k < len ? static_cast<int32_t>(keys[p0 + k] & 0xffffffffu) : pad;
}

// global -> local dof ids (multi-rank): owned rows [row_start,row_start+n_owned) first,
// ghosts after, ascending global id
__global__ void __launch_bounds__(kBlock) k_localize_dofs(int32_t *edof, int64_t count, int64_t row_start,
                                                           int64_t n_owned, const int64_t *ghost_gid, int64_t n_ghost)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= count) return;
    const int64_t g = edof[i];
    if (g < 0) return;
    if (g >= row_start && g < row_start + n_owned) { edof[i] = static_cast<int32_t>(g - row_start); return; }
    int64_t lo = 0, hi = n_ghost;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (ghost_gid[mid] < g) lo = mid + 1; else hi = mid; }
    edof[i] = static_cast<int32_t>(n_owned + lo);
}

// ---------------------------------------------------------------------------
// internal renumbering of the owned dofs of an uploaded mesh whose numbering has no locality (Morton order of the nodes)
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint64_t morton_spread21(uint64_t v)
{
    v &= 0x1fffffull;
    v = (v | (v << 32)) & 0x1f00000000ffffull;
    v = (v | (v << 16)) & 0x1f0000ff0000ffull;
    v = (v | (v << 8)) & 0x100f00f00f00f00full;
    v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}
// key of every owned dof = Morton code of its node (21 bits per axis inside the bounding box); the dofs of a node share
// the key and stay in their order (the sort that follows is stable)
__global__ void __launch_bounds__(kBlock) k_morton_keys(MeshDev m, int64_t n_owned, double x0, double y0, double z0, double sx, double sy,
                                                         double sz, uint64_t *__restrict__ keys)
{
    const int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (t >= m.nElem * m.npe) return;
    const int64_t e = t % m.nElem;
    const int a = static_cast<int>(t / m.nElem);
    const int32_t nd = m.conn[a * m.nElem + e];
    const double px = (m.xyz[nd] - x0) * sx, py = (m.xyz[m.nNode + nd] - y0) * sy;
    const double pz = m.ndim > 2 ? (m.xyz[2 * m.nNode + nd] - z0) * sz : 0.0;
    const uint64_t key = morton_spread21(static_cast<uint64_t>(px)) | (morton_spread21(static_cast<uint64_t>(py)) << 1) |
                         (morton_spread21(static_cast<uint64_t>(pz)) << 2);
    for (int d = 0; d < m.ndof; ++d) {
        const int32_t l = m.edof[(a * m.ndof + d) * m.nElem + e];
        if (l >= 0 && l < n_owned) keys[l] = key;          // every visit of a dof writes the same value
    }
}
// ---- the nodes follow their dofs (internal renumbering): a node's key = its smallest internal dof, nodes without dofs last
__global__ void __launch_bounds__(kBlock) k_node_first_dof(MeshDev m, unsigned long long *__restrict__ key)
{
    const int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (t >= m.nElem * m.npe) return;
    const int64_t e = t % m.nElem;
    const int a = static_cast<int>(t / m.nElem);
    const int32_t nd = m.conn[a * m.nElem + e];
    for (int d = 0; d < m.ndof; ++d) {
        const int32_t l = m.edof[(a * m.ndof + d) * m.nElem + e];
        if (l >= 0) atomicMin(&key[nd], static_cast<unsigned long long>(l));
    }
}
__global__ void __launch_bounds__(kBlock) k_relabel_nodes(int32_t *__restrict__ conn, int64_t count, const int32_t *__restrict__ nperm)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i < count) conn[i] = nperm[conn[i]];
}
// out[c][new] = in[c][order[new]] for c < planes (SoA node arrays)
__global__ void __launch_bounds__(kBlock) k_gather_nodes(int64_t n, int planes, const int32_t *__restrict__ order, const double *__restrict__ in,
                                                          double *__restrict__ out)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= n) return;
    const int64_t o = order[i];
    for (int c = 0; c < planes; ++c) out[c * n + i] = in[c * n + o];
}
// (node-major arrays: solnApplied[node * ndof + d])
__global__ void __launch_bounds__(kBlock) k_gather_node_rows(int64_t n, int w, const int32_t *__restrict__ order, const double *__restrict__ in,
                                                              double *__restrict__ out)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= n) return;
    const int64_t o = order[i];
    for (int c = 0; c < w; ++c) out[i * w + c] = in[o * w + c];
}

__global__ void __launch_bounds__(kBlock) k_perm_from_order(int64_t n, const int32_t *__restrict__ order, int32_t *__restrict__ perm)
{
    const int64_t k = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (k < n) perm[order[k]] = static_cast<int32_t>(k);
}
__global__ void __launch_bounds__(kBlock) k_apply_perm(int32_t *__restrict__ edof, int64_t count, int64_t n_owned, const int32_t *__restrict__ perm)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= count) return;
    const int32_t l = edof[i];
    if (l >= 0 && l < n_owned) edof[i] = perm[l];
}
// out[i] = in[perm[i]] over the owned part, the rest copied
__global__ void __launch_bounds__(kBlock) k_gather_perm(int64_t n, int64_t n_owned, const int32_t *__restrict__ perm, const double *__restrict__ in,
                                                         double *__restrict__ out)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i < n) out[i] = in[i < n_owned ? perm[i] : i];
}

// ---------------------------------------------------------------------------
// structured box generated on the device (genTetra.cpp's mesh + the driver's numbering for one slab along any axis)
// ---------------------------------------------------------------------------
struct BoxOwnerDev {
    int64_t start;            // global id of the first free dof of the owning rank
    int lo[3], cnt[3];        // its box of free nodes, numbered x-fastest
};
struct BoxDev {
    int N[3];                 // nodes per side of the WHOLE box
    int axis, l0, l1;         // hex layers [l0,l1) of this slab along `axis`; local node planes l0..l1
    int own_lo;               // first node plane along `axis` that this rank owns (plane l0 belongs to the rank below)
    int bc_mode, ndof;
    int Ln[3], Le[3];         // local node / hex box
    BoxOwnerDev own, prev;    // the reference's renumbering: ranks concatenated, ascending old id inside a rank
    const double *X, *Y, *Z;  // axis tables as read back from the "%.8f" node file
};

// global free-dof id of (i,j,k,d) in the reference's numbering (free dofs counted scanning the NEW node order:
// rank by rank, lexicographic (k,j,i) inside a rank), -1 if constrained.  bc_mode 0: all six faces; 1: the plane j == 0.
__device__ __forceinline__ int32_t box_dof(const BoxDev &b, int i, int j, int k, int d)
{
    const int c[3] = {i, j, k};
    const BoxOwnerDev &o = c[b.axis] >= b.own_lo ? b.own : b.prev;
    const int a0 = i - o.lo[0], a1 = j - o.lo[1], a2 = k - o.lo[2];
    if (a0 < 0 || a1 < 0 || a2 < 0 || a0 >= o.cnt[0] || a1 >= o.cnt[1] || a2 >= o.cnt[2]) return -1;
    const int64_t node = (static_cast<int64_t>(a2) * o.cnt[1] + a1) * o.cnt[0] + a0;
    return static_cast<int32_t>(o.start + node * b.ndof + d);
}

// coordinates of the slab's nodes (local id x-fastest over the local node box), prescribed values cleared
__global__ void __launch_bounds__(kBlock) k_box_nodes(BoxDev b, int64_t nNode, double *xyz, double *soln)
{
    const int64_t n = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (n >= nNode) return;
    const int64_t plane = static_cast<int64_t>(b.Ln[0]) * b.Ln[1];
    const int k = static_cast<int>(n / plane), rem = static_cast<int>(n % plane);
    const int j = rem / b.Ln[0], i = rem % b.Ln[0];
    const int o0 = b.axis == 0 ? b.l0 : 0, o1 = b.axis == 1 ? b.l0 : 0, o2 = b.axis == 2 ? b.l0 : 0;
    xyz[n] = b.X[o0 + i];
    xyz[nNode + n] = b.Y[o1 + j];
    xyz[2 * nNode + n] = b.Z[o2 + k];
    for (int d = 0; d < b.ndof; ++d) soln[n * b.ndof + d] = 0.0;
}

// six tets per hex in genTetra.cpp's order and orientation (:263-322): connectivity (local node ids) and the
// element dof array (GLOBAL dof ids, -1 = constrained), both SoA.  Local hexes in ascending global element id.
__global__ void __launch_bounds__(kBlock) k_box_elems(BoxDev b, int64_t nHex, int32_t *conn, int32_t *edof)
{
    const int64_t h = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (h >= nHex) return;
    const int i = static_cast<int>(h % b.Le[0]), j = static_cast<int>((h / b.Le[0]) % b.Le[1]),
              k = static_cast<int>(h / (static_cast<int64_t>(b.Le[0]) * b.Le[1]));
    const int o0 = b.axis == 0 ? b.l0 : 0, o1 = b.axis == 1 ? b.l0 : 0, o2 = b.axis == 2 ? b.l0 : 0;
    const int64_t plane = static_cast<int64_t>(b.Ln[0]) * b.Ln[1], nElem = 6 * nHex;
    const int split[6][4] = {{0, 1, 3, 5}, {0, 3, 2, 5}, {2, 3, 7, 5}, {4, 6, 7, 2}, {4, 7, 5, 2}, {0, 4, 5, 2}};
    for (int t = 0; t < 6; ++t) {
        const int64_t e = 6 * h + t;
        for (int a = 0; a < 4; ++a) {
            const int c = split[t][a];
            const int ci = i + (c & 1), cj = j + ((c >> 1) & 1), ck = k + ((c >> 2) & 1);
            conn[a * nElem + e] = static_cast<int32_t>(plane * ck + static_cast<int64_t>(b.Ln[0]) * cj + ci);
            for (int d = 0; d < b.ndof; ++d) edof[(a * b.ndof + d) * nElem + e] = box_dof(b, o0 + ci, o1 + cj, o2 + ck, d);
        }
    }
}

// prescribed values at the listed (local node * ndof + dof) slots
__global__ void __launch_bounds__(kBlock) k_box_bc(const int64_t *slot, const double *val, int64_t n, double *soln)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i < n) soln[slot[i]] = val[i];
}

// ---------------------------------------------------------------------------
// numeric assembly: one thread per element
// ---------------------------------------------------------------------------
// slot of (row,col) in the wave-sliced storage, -1 if not in the pattern
__device__ __forceinline__ int64_t find_slot(const SellDev &A, int row, int col)
{
    const int64_t base = A.slice_off[row >> 6] + (row & 63);
    int lo = 0, hi = A.rowlen[row] - 1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        const int c = A.cols[base + (static_cast<int64_t>(mid) << 6)];
        if (c == col) return base + (static_cast<int64_t>(mid) << 6);
        if (c < col) lo = mid + 1; else hi = mid - 1;
    }
    return -1;
}

__device__ __forceinline__ void add_f64(double *p, double v)
{
    // hardware global_atomic_add_f64 (-munsafe-fp-atomics), device scope
    atomicAdd(p, v);
}

// Poisson tet / tria (1 dof per node): Ke + Fe, Dirichlet lifting, scatter.
//   MatSetValues reads the column-major Klocal row-major, i.e. entry (row_i,col_j)
//   receives Klocal(j,i) (tetrapoissonparallelimpl1.F:851); lifting :859-870;
//   VecSetValues :880.
template <int KIND>
__global__ void __launch_bounds__(kBlock) k_assemble_scalar(MeshDev m, SellDev A, double *rhs, ElemPrm prm, int *err,
                                                             const uint8_t *__restrict__ only_nodes)
{
    // only_nodes != nullptr: the hub pass of the gather form -- only the rows of flagged nodes are scattered
    constexpr int NPE = (KIND == PFEM_POISSON_TET) ? 4 : 3;
    const int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (e >= m.nElem) return;
    int nd[NPE], dof[NPE];
    double x[NPE], y[NPE], z[NPE];
    bool any = only_nodes == nullptr;
#pragma unroll
    for (int a = 0; a < NPE; ++a) {
        nd[a] = m.conn[a * m.nElem + e];
        dof[a] = m.edof[a * m.nElem + e];
        if (only_nodes && only_nodes[nd[a]]) any = true;
    }
    if (!any) return;
#pragma unroll
    for (int a = 0; a < NPE; ++a) {
        x[a] = m.xyz[nd[a]];
        y[a] = m.xyz[m.nNode + nd[a]];
        z[a] = (KIND == PFEM_POISSON_TET) ? m.xyz[2 * m.nNode + nd[a]] : 0.0;
    }
    double K[NPE * NPE], F[NPE];
    const double valC[4] = {0.0, 0.0, 0.0, 0.0};   // drivers pass valC = 0 (:824)
    bool ok;
    if constexpr (KIND == PFEM_POISSON_TET)
        ok = poisson_tet(x, y, z, prm.ed[0], prm.ed[1], prm.ed[2], prm.af, valC, K, F);
    else if constexpr (KIND == PFEM_POISSON_TRIA)
        ok = poisson_tria(x, y, prm.ed[0], prm.ed[1], prm.af, valC, K, F);
    else
        ok = poisson_tria_inline(x, y, K, F);
    if (!ok) { atomicMax(err, PFEM_ERR_NEG_JAC); return; }
#pragma unroll
    for (int i = 0; i < NPE; ++i)
        if (dof[i] < 0) {
            const double fact = m.soln[nd[i]];
#pragma unroll
            for (int j = 0; j < NPE; ++j)
                if (dof[j] >= 0) F[j] = F[j] - K[j + NPE * i] * fact;
        }
#pragma unroll
    for (int i = 0; i < NPE; ++i) {
        if (dof[i] < 0) continue;
        if (only_nodes && !only_nodes[nd[i]]) continue;
#pragma unroll
        for (int j = 0; j < NPE; ++j) {
            if (dof[j] < 0) continue;
            const int64_t s = find_slot(A, dof[i], dof[j]);
            if (s < 0) { atomicMax(err, PFEM_ERR_PATTERN); continue; }
            add_f64(&A.vals[s], K[j + NPE * i]);
        }
        add_f64(&rhs[dof[i]], F[i]);
    }
}

// Linear elasticity tet (3 dofs per node).  The 12x12 Ke is never materialised: the
// B^T (D B) contraction is streamed as sixteen 3x3 node blocks from registers
// (pfem_elem.hpp: elast_block) straight into lifting + scatter.  A node's free dofs
// are consecutive global ids, hence consecutive entries of a sorted row, so one
// binary search per (row, node) serves up to three columns.
__global__ void __launch_bounds__(kBlock) k_assemble_elast(MeshDev m, SellDev A, double *rhs, ElemPrm prm, int *err,
                                                            const uint8_t *__restrict__ only_nodes)
{
    const int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (e >= m.nElem) return;
    int nd[4], dof[12];
    double x[4], y[4], z[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) nd[a] = m.conn[a * m.nElem + e];
    if (only_nodes && !(only_nodes[nd[0]] | only_nodes[nd[1]] | only_nodes[nd[2]] | only_nodes[nd[3]])) return;   // hub pass
#pragma unroll
    for (int i = 0; i < 12; ++i) dof[i] = m.edof[i * m.nElem + e];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        x[a] = m.xyz[nd[a]];
        y[a] = m.xyz[m.nNode + nd[a]];
        z[a] = m.xyz[2 * m.nNode + nd[a]];
    }
    TetGeom g;
    tet_geometry(x, y, z, g);
    if (g.jac < 0.0) { atomicMax(err, PFEM_ERR_NEG_JAC); return; }
    const double dvol = kGaussWtTet * g.jac;
    const ElastMat mat = elast_material(prm.ed[0], prm.ed[1]);
    double F[12];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const double b4 = dvol * 0.25;
        F[3 * a + 0] = 0.0 + b4 * prm.ed[3];
        F[3 * a + 1] = 0.0 + b4 * prm.ed[4];
        F[3 * a + 2] = 0.0 + b4 * prm.ed[5];
    }
    // Dirichlet values of the constrained dofs (solnApplied(elemDofGlobal), :938-950)
    double fact[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) fact[i] = dof[i] < 0 ? m.soln[3LL * nd[i / 3] + i % 3] : 0.0;

#pragma unroll
    for (int a = 0; a < 4; ++a) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            double blk[3][3];
            elast_block(g, mat, dvol, a, b, blk);   // blk[p][q] = Klocal(3a+p, 3b+q)
            // lifting: Flocal(3a+p) -= Klocal(3a+p, 3b+q) * u_D(3b+q), q ascending
#pragma unroll
            for (int q = 0; q < 3; ++q)
                if (dof[3 * b + q] < 0) {
#pragma unroll
                    for (int p = 0; p < 3; ++p)
                        if (dof[3 * a + p] >= 0) F[3 * a + p] = F[3 * a + p] - blk[p][q] * fact[3 * b + q];
                }
            // MatSetValues row-major read: entry (row = dof(3b+q), col = dof(3a+p)) += Klocal(3a+p,3b+q)
            int firstp = -1;
#pragma unroll
            for (int p = 2; p >= 0; --p)
                if (dof[3 * a + p] >= 0) firstp = p;
            if (firstp < 0) continue;
            if (only_nodes && !only_nodes[nd[b]]) continue;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int row = dof[3 * b + q];
                if (row < 0) continue;
                int64_t s = find_slot(A, row, dof[3 * a + firstp]);
                if (s < 0) { atomicMax(err, PFEM_ERR_PATTERN); continue; }
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    if (dof[3 * a + p] < 0) continue;
                    add_f64(&A.vals[s], blk[p][q]);
                    s += 64;
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 12; ++i)
        if (dof[i] >= 0 && (!only_nodes || only_nodes[nd[i / 3]])) add_f64(&rhs[dof[i]], F[i]);
}

// ---------------------------------------------------------------------------
// numeric assembly, gather form: one thread per NODE, no atomics
// ---------------------------------------------------------------------------
// The scatter form above pays one memory-side atomic per element-matrix entry
// (960 M atomics / 15.9 GB of HBM writes at 200^3; profiles/r01).  Here each node walks
// its incident elements in ASCENDING element order (inc_ea, built in the symbolic phase),
// re-evaluates their Ke and adds its own rows with plain read-modify-writes: a matrix row
// has exactly one writer.  Ke is evaluated ~4x redundantly (~70 GFLOP at 200^3, nothing on
// this chip) in exchange for zero atomics and 1x value traffic.  Because every slot receives
// its element contributions in the same order as the serial reference loop
// (tetrapoissonparallelimpl1.F:828-884 on one rank), K and F are BIT-IDENTICAL to the
// reference-equivalent serial assembly, and run-to-run deterministic.
__device__ __forceinline__ int64_t find_slot_in(const SellDev &A, int64_t base, int len, int col)
{
    int lo = 0, hi = len - 1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        const int c = A.cols[base + (static_cast<int64_t>(mid) << 6)];
        if (c == col) return base + (static_cast<int64_t>(mid) << 6);
        if (c < col) lo = mid + 1; else hi = mid - 1;
    }
    return -1;
}

// inc_key = (node << 32) | (4*e + a) for every (element, local node)
__global__ void __launch_bounds__(kBlock) k_emit_inc_keys(MeshDev m, uint64_t *keys)
{
    const int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (e >= m.nElem) return;
    for (int a = 0; a < m.npe; ++a)
        keys[a * m.nElem + e] = (static_cast<uint64_t>(static_cast<uint32_t>(m.conn[a * m.nElem + e])) << 32) |
                                static_cast<uint32_t>(4 * e + a);
}

// per-node incidence count and per-chunk (64 nodes = one wave) padded size in entries
__global__ void __launch_bounds__(kBlock) k_inc_chunk_sizes(const int64_t *node_ptr, int64_t nNode, int64_t n_chunks,
                                                             int32_t *cnt, int64_t *chunk_entries)
{
    const int64_t n = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    int c = 0;
    if (n < nNode) {
        c = static_cast<int>(node_ptr[n + 1] - node_ptr[n]);
        cnt[n] = c;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c = max(c, __shfl_xor(c, o, 64));
    const int64_t ch = n >> 6;
    if ((threadIdx.x & 63) == 0 && ch < n_chunks) chunk_entries[ch] = 64LL * c;
    if (n == 0) chunk_entries[n_chunks] = 0;
}

// node-contiguous sorted keys -> wave-sliced (4e+a) lists, -1 padded
__global__ void __launch_bounds__(kBlock) k_inc_fill(const uint64_t *keys, const int64_t *node_ptr, int64_t nNode,
                                                      int64_t n_chunks, const int64_t *chunk_off, int32_t *ea)
{
    const int64_t n = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    const int64_t ch = n >> 6;
    if (ch >= n_chunks) return;
    const int64_t off = chunk_off[ch] + (n & 63);
    const int width = static_cast<int>((chunk_off[ch + 1] - chunk_off[ch]) >> 6);
    int64_t p0 = 0;
    int c = 0;
    if (n < nNode) { p0 = node_ptr[n]; c = static_cast<int>(node_ptr[n + 1] - p0); }
    for (int j = 0; j < width; ++j) ea[off + 64LL * j] = j < c ? static_cast<int32_t>(keys[p0 + j] & 0xffffffffu) : -1;
}

// ---- the pattern from the incidence lists, no sort of element-matrix keys --------------------------------------------
// The rows of a node share one column set: the free dofs of every node of every incident element.  One thread per node
// collects the distinct neighbour nodes of its node in LDS (sorted insertion; entry k of thread t at u[k*T + t]) under the
// key first free dof * 4 + (free dofs - 1): a node's free dofs are consecutive numbers (checked by k_node_keys; what the
// gather assembly relies on as well), so ascending keys expand to ascending columns.  FILL = false: the length of the
// node's rows; FILL = true: the columns, straight into the wave-sliced storage.  The sorted form (pattern_from_keys: every
// entry of every element matrix through a 64-bit radix sort, 768 M keys at config 3) remains for patterns without a mesh and
// as the fallback (nodes of very many elements, dofs of a node not consecutive).
__global__ void __launch_bounds__(kBlock) k_node_keys(MeshDev m, const int64_t *__restrict__ inc_ptr, const int32_t *__restrict__ inc_cnt,
                                                       const int32_t *__restrict__ inc_ea, int32_t *__restrict__ node_key, int *__restrict__ info)
{
    const int64_t n = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    const int cnt = n < m.nNode ? inc_cnt[n] : 0;
    {
        int c = cnt;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c = max(c, __shfl_xor(c, o, 64));              // one atomic per wave
        if ((threadIdx.x & 63) == 0 && c > 0) atomicMax(info + 1, c);
    }
    if (n >= m.nNode) return;
    int32_t key = -1;
    if (cnt > 0) {
        const int ea0 = inc_ea[inc_ptr[n >> 6] + (n & 63)];
        int first = -1, nfree = 0;
        bool ok = true;
        for (int q = 0; q < m.ndof; ++q) {
            const int d = m.edof[static_cast<int64_t>(m.ndof * (ea0 & 3) + q) * m.nElem + (ea0 >> 2)];
            if (d < 0) continue;
            if (first < 0) first = d;
            else ok = ok && d == first + nfree;
            ++nfree;
        }
        if (!ok || first >= (1 << 29) || nfree > 4) atomicMax(info, 1);
        if (nfree > 0) key = first * 4 + (nfree - 1);
    }
    node_key[n] = key;
}
template <bool FILL>
__global__ void __launch_bounds__(kBlock) k_pattern_rows(MeshDev m, const int64_t *__restrict__ inc_ptr, const int32_t *__restrict__ inc_cnt,
                                                          const int32_t *__restrict__ inc_ea, const int32_t *__restrict__ node_key,
                                                          int32_t *__restrict__ rowlen, const int64_t *__restrict__ slice_off,
                                                          int32_t *__restrict__ cols)
{
    extern __shared__ int32_t lds_u[];
    const int T = blockDim.x;
    const int64_t n = static_cast<int64_t>(blockIdx.x) * T + threadIdx.x;
    if (n >= m.nNode) return;
    const int cnt = inc_cnt[n];
    const int32_t own = node_key[n];
    if (cnt == 0 || own < 0) return;
    int32_t *u = lds_u + threadIdx.x;
    u[0] = own;                                  // the node itself (local node `a` of every incident element: not looked up again)
    int len = 1;
    const int64_t beg = inc_ptr[n >> 6] + (n & 63);
    for (int j = 0; j < cnt; ++j) {
        const int ea = inc_ea[beg + 64LL * j];
        const int64_t e = ea >> 2;
        for (int b = 0; b < m.npe; ++b) {
            if (b == (ea & 3)) continue;
            const int32_t key = node_key[m.conn[b * m.nElem + e]];
            if (key < 0) continue;
            int k = len;
            while (k > 0 && u[(k - 1) * T] > key) --k;
            if (k > 0 && u[(k - 1) * T] == key) continue;
            for (int q = len; q > k; --q) u[q * T] = u[(q - 1) * T];
            u[k * T] = key;
            ++len;
        }
    }
    const int first = own >> 2, nfree = (own & 3) + 1;
    if (!FILL) {
        int total = 0;
        for (int k = 0; k < len; ++k) total += (u[k * T] & 3) + 1;
        for (int q = 0; q < nfree; ++q) rowlen[first + q] = total;
    } else {
        for (int q = 0; q < nfree; ++q) {
            const int64_t r = first + q;
            int32_t *c = cols + slice_off[r >> 6] + (r & 63);
            int at = 0;
            for (int k = 0; k < len; ++k) {
                const int32_t key = u[k * T];
                for (int d = 0; d <= (key & 3); ++d) c[64LL * at++] = (key >> 2) + d;
            }
        }
    }
}
// padding of the wave-sliced column array: a row's own index (rows past the end: 0), as k_fill_sell leaves it
__global__ void __launch_bounds__(kBlock) k_fill_pad_cols(int64_t n_rows, int64_t n_slices, const int64_t *__restrict__ slice_off,
                                                           const int32_t *__restrict__ rowlen, int32_t *__restrict__ cols)
{
    const int64_t r = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    const int64_t s = r >> 6;
    if (s >= n_slices) return;
    const int64_t off = slice_off[s] + (r & 63);
    const int width = static_cast<int>((slice_off[s + 1] - slice_off[s]) >> 6);
    const int len = r < n_rows ? rowlen[r] : 0;
    const int32_t pad = r < n_rows ? static_cast<int32_t>(r) : 0;
    for (int k = len; k < width; ++k) cols[off + 64LL * k] = pad;
}
// rowptr[i] for i <= n from the row lengths' exclusive scan (int64: more than 2^31 stored entries are legal here)
__global__ void __launch_bounds__(kBlock) k_widen_i32(const int32_t *__restrict__ in, int64_t n, int64_t *__restrict__ out)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i < n) out[i] = in[i];
}

// Slot map of the gather form, built once per pattern: for incidence t = (node n, element e) the
// byte b of inc_slots[t] is the entry index k, inside n's matrix rows, of the FIRST free dof of
// e's local node b (0xff: node b fully constrained).  The rows of one node share their column
// set and a node's free dofs are consecutive columns, so one byte per element node serves all
// (row, column) pairs; the numeric kernels then never search.
__global__ void __launch_bounds__(kBlock) k_build_inc_slots(MeshDev m, SellDev A, const int64_t *__restrict__ inc_ptr,
                                                             const int32_t *__restrict__ inc_cnt,
                                                             const int32_t *__restrict__ inc_ea, uint32_t *inc_slots, int *err,
                                                             uint8_t *node_hub, int *n_hubs)
{
    const int64_t n = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (n >= m.nNode) return;
    node_hub[n] = 0;
    // incidence lists are wave-sliced like the matrix: entry j of node n sits at
    // inc_ptr[n >> 6] + 64*j + (n & 63), so a wave reads one contiguous 256-B run per step
    const int cnt = inc_cnt[n];
    if (cnt == 0) return;
    const int64_t beg = inc_ptr[n >> 6] + (n & 63), end = beg + 64LL * cnt;
    const int ea0 = inc_ea[beg];
    int row = -1;
    for (int p = m.ndof - 1; p >= 0; --p) {
        const int r = m.edof[static_cast<int64_t>(m.ndof * (ea0 & 3) + p) * m.nElem + (ea0 >> 2)];
        if (r >= 0) row = r;
    }
    if (row < 0) { for (int64_t t = beg; t < end; t += 64) inc_slots[t] = 0xffffffffu; return; }
    const int64_t base = A.slice_off[row >> 6] + (row & 63);
    const int len = A.rowlen[row];
    if (len > 255) { node_hub[n] = 1; atomicAdd(n_hubs, 1); return; }   // byte-sized entry index: a hub, left to the hub pass
    for (int64_t t = beg; t < end; t += 64) {
        const int64_t e = inc_ea[t] >> 2;
        uint32_t w = 0;
        for (int b = 0; b < 4; ++b) {
            uint32_t k = 0xffu;
            if (b < m.npe) {
                int c = -1;
                for (int q = m.ndof - 1; q >= 0; --q) {
                    const int d = m.edof[static_cast<int64_t>(m.ndof * b + q) * m.nElem + e];
                    if (d >= 0) c = d;
                }
                if (c >= 0) {
                    const int64_t sl = find_slot_in(A, base, len, c);
                    if (sl < 0) atomicMax(err, 2); else k = static_cast<uint32_t>((sl - base) >> 6);
                }
            }
            w |= k << (8 * b);
        }
        inc_slots[t] = w;
    }
}

// Packed incidence record: {o0, o1, o2, slots}, the element's OTHER nodes in local order (the
// node itself is implied by the list it sits in; its local position a rides in the sign bits of
// o0 / o1) and the slot bytes.  With it the numeric kernels stream 16 B per visit, coalesced, and
// read neither connectivity nor dof arrays.  1-dof kinds: a slot byte 0xff says "constrained";
// kinds with more dofs per node also get inc_flags (bit ndof*b+q set: dof q of local node b is
// constrained).  node_row[ndof*n+p] is the matrix row of dof p of node n, -1 if it has none.
__global__ void __launch_bounds__(kBlock) k_build_inc_rec(MeshDev m, const int64_t *__restrict__ inc_ptr,
                                                           const int32_t *__restrict__ inc_cnt,
                                                           const int32_t *__restrict__ inc_ea,
                                                           const uint32_t *__restrict__ inc_slots, int4 *inc_rec,
                                                           uint16_t *inc_flags, int32_t *node_row,
                                                           const uint8_t *__restrict__ node_hub)
{
    const int64_t n = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (n >= m.nNode) return;
    const int cnt = inc_cnt[n];
    if (cnt == 0 || node_hub[n]) {          // no incident element, or a hub (its rows belong to the hub pass)
        for (int p = 0; p < m.ndof; ++p) node_row[n * m.ndof + p] = -1;
        return;
    }
    const int64_t beg = inc_ptr[n >> 6] + (n & 63), end = beg + 64LL * cnt;
    const int ea0 = inc_ea[beg];
    for (int p = 0; p < m.ndof; ++p)
        node_row[n * m.ndof + p] = m.edof[static_cast<int64_t>(m.ndof * (ea0 & 3) + p) * m.nElem + (ea0 >> 2)];
    for (int64_t t = beg; t < end; t += 64) {
        const int ea = inc_ea[t];
        const int64_t e = ea >> 2;
        const uint32_t a = static_cast<uint32_t>(ea & 3);
        uint32_t o[3] = {0u, 0u, 0u};
        int q = 0;
        for (int b = 0; b < m.npe; ++b)
            if (b != static_cast<int>(a)) o[q++] = static_cast<uint32_t>(m.conn[b * m.nElem + e]);
        o[0] |= (a & 1u) << 31;
        o[1] |= (a >> 1) << 31;
        inc_rec[t] = make_int4(static_cast<int>(o[0]), static_cast<int>(o[1]), static_cast<int>(o[2]),
                               static_cast<int>(inc_slots[t]));
        if (inc_flags) {
            uint32_t fl = 0;
            for (int i = 0; i < m.nsize; ++i)
                if (m.edof[static_cast<int64_t>(i) * m.nElem + e] < 0) fl |= 1u << i;
            inc_flags[t] = static_cast<uint16_t>(fl);
        }
    }
}

// ---- incidence lists as TRANSLATED copies of a few patterns ----
// The packed records are 16 B per visit: 3.1 GB at config 3, two thirds of what the gather kernel fetches from HBM (rocprofv3
// --pmc: 3.3 GB fetched, 1.5 GB written in 1.48 ms -- and the same kernel with the element geometry for nothing is no faster,
// profiles/r06/assembly_kernel_bound.txt: the kernel waits for memory, not for its arithmetic).  On a mesh numbered along lines
// (the reference's boxes; any mesh whose generator numbers a repeated cell the same way) the list of node n is the list of node
// n + 1 shifted by one: with the other nodes stored RELATIVE to n (the slot bytes are row-relative already) a few hundred
// distinct lists remain -- interior, and next to each face, edge and corner of the constrained boundary.  The node keeps a 2-byte
// pattern number; the patterns (kIncPatMax x longest list x 16 B) stay in cache.  Nothing is assumed: every node's list is
// compared with its pattern word for word when the table is built (k_incpat_assign), a mesh with more patterns than the table
// holds keeps its own records.  Same records in the same order: the same bits.
constexpr int kIncPatMax = 1024;                 // patterns (ids are 16 bits)
constexpr int kIncPatSlots = 4096;               // hash slots, >= 2 x kIncPatMax
struct IncPatSlot { unsigned long long key; int rep; int id; };
struct IncPatState { int count, overflow, fail, stride; };
// other-node word of a record: low 31 bits = node number, or (pattern form) the two's-complement difference to the list's node
__device__ __forceinline__ int incpat_rel31(int word, int n, bool rel)
{
    const uint32_t w = static_cast<uint32_t>(word);
    return static_cast<int>((rel ? (w & 0x7fffffffu) - static_cast<uint32_t>(n) : w) & 0x7fffffffu) | static_cast<int>(w & 0x80000000u);
}
__device__ __forceinline__ int incpat_node(int word, int n) { return n + (static_cast<int>(static_cast<uint32_t>(word) << 1) >> 1); }
__device__ __forceinline__ unsigned long long incpat_mix(unsigned long long h, unsigned long long v)
{
    h ^= v + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
    h *= 0xff51afd7ed558ccdull;
    return h ^ (h >> 32);
}
// visit t of node n in relative form (npe - 1 other nodes; the unused words of a triangle's record stay as they are)
__device__ __forceinline__ int4 incpat_record(const int4 rc, int n, int npe)
{
    return make_int4(incpat_rel31(rc.x, n, true), incpat_rel31(rc.y, n, true), npe > 3 ? rc.z - n : rc.z, rc.w);
}
__device__ __forceinline__ unsigned long long incpat_hash(int64_t n, int cnt, int64_t beg, const int4 *__restrict__ inc_rec,
                                                          const uint16_t *__restrict__ inc_flags, int npe)
{
    unsigned long long h = incpat_mix(0x243f6a8885a308d3ull, static_cast<unsigned long long>(cnt));
    for (int t = 0; t < cnt; ++t) {
        const int4 r = incpat_record(inc_rec[beg + 64LL * t], static_cast<int>(n), npe);
        h = incpat_mix(h, (static_cast<unsigned long long>(static_cast<uint32_t>(r.x)) << 32) | static_cast<uint32_t>(r.y));
        h = incpat_mix(h, (static_cast<unsigned long long>(static_cast<uint32_t>(r.z)) << 32) | static_cast<uint32_t>(r.w));
        if (inc_flags) h = incpat_mix(h, inc_flags[beg + 64LL * t]);
    }
    return h == 0 ? 1 : h;
}
// every node with a row of its own enters its list's hash; the smallest such node of a pattern represents it
__global__ void __launch_bounds__(kBlock) k_incpat_collect(int64_t nNode, int ndof, int npe, const int64_t *__restrict__ inc_ptr,
                                                            const int32_t *__restrict__ inc_cnt, const int4 *__restrict__ inc_rec,
                                                            const uint16_t *__restrict__ inc_flags, const int32_t *__restrict__ node_row,
                                                            IncPatSlot *table, IncPatState *st)
{
    const int64_t n = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (n >= nNode || st->overflow) return;
    bool has_row = false;
    for (int p = 0; p < ndof; ++p) has_row |= node_row[n * ndof + p] >= 0;
    if (!has_row) return;
    const int cnt = inc_cnt[n];
    const int64_t beg = inc_ptr[n >> 6] + (n & 63);
    const unsigned long long h = incpat_hash(n, cnt, beg, inc_rec, inc_flags, npe);
    uint32_t q = static_cast<uint32_t>(h >> 20) & (kIncPatSlots - 1);
    for (int tries = 0; tries < kIncPatSlots; ++tries, q = (q + 1) & (kIncPatSlots - 1)) {
        unsigned long long seen = *reinterpret_cast<volatile unsigned long long *>(&table[q].key);
        if (seen == 0) {
            seen = atomicCAS(&table[q].key, 0ull, h);
            if (seen == 0) {
                if (atomicAdd(&st->count, 1) >= kIncPatMax) { st->overflow = 1; return; }
                seen = h;
            }
        }
        if (seen == h) {
            // (millions of nodes share the interior's pattern: an atomic only where it can change the word -- the smallest node
            // of a pattern and the longest list are reached after a few of them; 8 M atomics on one address cost 100 ms)
            if (static_cast<int>(n) < *reinterpret_cast<volatile int *>(&table[q].rep)) atomicMin(&table[q].rep, static_cast<int>(n));
            if (cnt > *reinterpret_cast<volatile int *>(&st->stride)) atomicMax(&st->stride, cnt);
            return;
        }
    }
    st->overflow = 1;
}
// patterns numbered by their representatives (ascending node number -- whatever order the slots were claimed in)
__global__ void __launch_bounds__(kBlock) k_incpat_number(IncPatSlot *table, IncPatState *st, int32_t *pat_rep)
{
    if (st->overflow) return;
    __shared__ int reps[kIncPatSlots];                 // (INT_MAX in the empty slots: never below anybody's)
    for (int j = threadIdx.x; j < kIncPatSlots; j += kBlock) reps[j] = table[j].key != 0 ? table[j].rep : INT_MAX;
    __syncthreads();
    const int q = blockIdx.x * kBlock + threadIdx.x;   // (kIncPatSlots / kBlock blocks: a slot to the thread)
    if (q >= kIncPatSlots || reps[q] == INT_MAX) return;
    const int rep = reps[q];
    int id = 0;
    for (int j = 0; j < kIncPatSlots; ++j) id += reps[j] < rep ? 1 : 0;
    table[q].id = id;
    pat_rep[id] = rep;
}
// the table: pattern id's visit t at pat_rec[id * stride + t] (+ the element's constrained-dof bits for kinds with ndof > 1)
__global__ void __launch_bounds__(kBlock) k_incpat_fill(const IncPatState *st, const int32_t *__restrict__ pat_rep, int npe,
                                                         const int64_t *__restrict__ inc_ptr, const int32_t *__restrict__ inc_cnt,
                                                         const int4 *__restrict__ inc_rec, const uint16_t *__restrict__ inc_flags,
                                                         int4 *pat_rec, uint16_t *pat_flags, int32_t *pat_cnt)
{
    if (st->overflow) return;
    const int stride = st->stride, np = min(st->count, kIncPatMax);
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= static_cast<int64_t>(np) * stride) return;
    const int id = static_cast<int>(i / stride), t = static_cast<int>(i % stride);
    const int n = pat_rep[id], cnt = inc_cnt[n];
    const int64_t beg = inc_ptr[n >> 6] + (n & 63);
    if (t == 0) pat_cnt[id] = cnt;
    pat_rec[i] = t < cnt ? incpat_record(inc_rec[beg + 64LL * t], n, npe) : make_int4(0, 0, 0, 0);
    if (pat_flags) pat_flags[i] = t < cnt ? inc_flags[beg + 64LL * t] : 0;
}
// every node: its pattern's number, and the word-for-word comparison of its list with the table (a hash collision, or anything
// else that makes them differ, refuses the form for the whole mesh)
__global__ void __launch_bounds__(kBlock) k_incpat_assign(int64_t nNode, int ndof, int npe, const int64_t *__restrict__ inc_ptr,
                                                           const int32_t *__restrict__ inc_cnt, const int4 *__restrict__ inc_rec,
                                                           const uint16_t *__restrict__ inc_flags, const int32_t *__restrict__ node_row,
                                                           const IncPatSlot *__restrict__ table, IncPatState *st, const int4 *__restrict__ pat_rec,
                                                           const uint16_t *__restrict__ pat_flags, const int32_t *__restrict__ pat_cnt,
                                                           uint16_t *node_pat)
{
    const int64_t n = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (n >= nNode || st->overflow) return;
    node_pat[n] = 0;
    bool has_row = false;
    for (int p = 0; p < ndof; ++p) has_row |= node_row[n * ndof + p] >= 0;
    if (!has_row) return;
    const int cnt = inc_cnt[n], stride = st->stride;
    const int64_t beg = inc_ptr[n >> 6] + (n & 63);
    const unsigned long long h = incpat_hash(n, cnt, beg, inc_rec, inc_flags, npe);
    uint32_t q = static_cast<uint32_t>(h >> 20) & (kIncPatSlots - 1);
    int id = -1;
    for (int tries = 0; tries < kIncPatSlots; ++tries, q = (q + 1) & (kIncPatSlots - 1)) {
        const unsigned long long key = table[q].key;
        if (key == h) { id = table[q].id; break; }
        if (key == 0) break;
    }
    bool same = id >= 0 && pat_cnt[id] == cnt;
    for (int t = 0; same && t < cnt; ++t) {
        const int4 a = incpat_record(inc_rec[beg + 64LL * t], static_cast<int>(n), npe), b = pat_rec[static_cast<int64_t>(id) * stride + t];
        same = a.x == b.x && a.y == b.y && a.z == b.z && a.w == b.w;
        if (same && pat_flags) same = inc_flags[beg + 64LL * t] == pat_flags[static_cast<int64_t>(id) * stride + t];
    }
    if (!same) { st->fail = 1; return; }
    node_pat[n] = static_cast<uint16_t>(id);
}

// longest matrix row among those the gather kernels own (hub rows excluded): sizes their LDS accumulators
__global__ void __launch_bounds__(kBlock) k_max_gather_row(const int32_t *__restrict__ node_row, int64_t n, const int32_t *__restrict__ rowlen,
                                                            int *out)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    int len = 0;
    if (i < n) {
        const int r = node_row[i];
        if (r >= 0) len = rowlen[r];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) len = max(len, __shfl_xor(len, o, 64));          // one atomic per wave
    if ((threadIdx.x & 63) == 0 && len > 0) atomicMax(out, len);
}

// The gather form never evaluates an element whose nodes are all constrained, but the
// reference STOPs on ANY element with a negative Jacobian (elementutilitiespoisson.F:157):
// the sign test runs once per mesh in the symbolic phase (coordinates do not change).
__global__ void __launch_bounds__(kBlock) k_check_jacobian(MeshDev m, int *err)
{
    const int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (e >= m.nElem) return;
    double x[4], y[4], z[4];
    for (int a = 0; a < m.npe; ++a) {
        const int n = m.conn[a * m.nElem + e];
        x[a] = m.xyz[n];
        y[a] = m.xyz[m.nNode + n];
        z[a] = m.ndim == 3 ? m.xyz[2 * m.nNode + n] : 0.0;
    }
    bool neg;
    if (m.ndim == 3) {
        TetGeom g;
        tet_geometry(x, y, z, g);
        neg = g.jac < 0.0;
    } else if (m.kind == PFEM_POISSON_TRIA || m.kind == PFEM_ELAST_TRIA) {
        // computeBasisFunctions2D: Jac = B11*B22 - B12*B21 with the accumulated differences
        const double b11 = ((0.0 + x[0] * -1.0) + x[1]) + x[2] * 0.0, b21 = ((0.0 + x[0] * -1.0) + x[1] * 0.0) + x[2];
        const double b12 = ((0.0 + y[0] * -1.0) + y[1]) + y[2] * 0.0, b22 = ((0.0 + y[0] * -1.0) + y[1] * 0.0) + y[2];
        neg = (b11 * b22 - b12 * b21) < 0.0;
    } else {
        neg = false;   // the inline element of triapoissonserialimpl1 has no orientation test
    }
    if (neg) atomicMax(err, PFEM_ERR_NEG_JAC);
}

// Workgroups are handed to the 8 XCDs round-robin (block b runs on XCD b % 8), each with its own L2.  The gather kernels
// read the coordinates of a node's neighbours, i.e. of the node chunks a line or a plane away: with the plain order those
// chunks are worked on by OTHER XCDs, and every XCD fetches every coordinate line it needs itself (PMC at cfg 3: 4.2 GB
// of fabric reads for 0.26 GB of coordinates).  Giving each XCD one contiguous eighth of the chunks keeps a chunk's
// neighbours in the same L2.  `per` = ceil(chunks / 8); the launch has 8 * per blocks, those past the end leave.
__device__ __forceinline__ int64_t xcd_contiguous_block(unsigned b, unsigned per, bool remap)
{
    return remap ? static_cast<int64_t>(b & 7u) * per + (b >> 3) : static_cast<int64_t>(b);
}

// LDSACC: the node's row is accumulated in LDS (entry k of thread t at acc[k*T + t], T = block
// size: conflict free) and written out once, coalesced, instead of ~6 global read-modify-writes
// per entry; the caller provides maxlen*T*8 bytes of dynamic LDS.  Same additions in the same order.
template <int KIND, bool LDSACC>
__global__ void __launch_bounds__(kBlock) k_gather_scalar(MeshDev m, SellDev A, double *rhs, ElemPrm prm,
                                                           const int64_t *__restrict__ inc_ptr,
                                                           const int32_t *__restrict__ inc_cnt,
                                                           const int4 *__restrict__ inc_rec,
                                                           const int32_t *__restrict__ node_row, int *err)
{
    extern __shared__ __attribute__((aligned(16))) double lds_acc[];
    constexpr int NPE = (KIND == PFEM_POISSON_TET) ? 4 : 3;
    const int T = blockDim.x;                  // 256, or 128 / 64 when long rows need the LDS space
    const int64_t n = static_cast<int64_t>(blockIdx.x) * T + threadIdx.x;
    if (n >= m.nNode) return;
    // incidence lists are wave-sliced like the matrix: entry j of node n sits at
    // inc_ptr[n >> 6] + 64*j + (n & 63), so a wave reads one contiguous 1-KiB run per step
    const int row = node_row[n];
    if (row < 0) return;                       // Dirichlet node (or a node of no element): no row
    const int cnt = inc_cnt[n];
    const int64_t beg = inc_ptr[n >> 6] + (n & 63), end = beg + 64LL * cnt;
    const int64_t base = A.slice_off[row >> 6] + (row & 63);
    const int len = A.rowlen[row];
    const double valC[4] = {0.0, 0.0, 0.0, 0.0};
    double facc = 0.0;
    double *acc = lds_acc + threadIdx.x;
    if (LDSACC)
        for (int k = 0; k < len; ++k) acc[k * T] = 0.0;
    for (int64_t t = beg; t < end; t += 64) {
        const int4 rc = inc_rec[t];
        const uint32_t slots = static_cast<uint32_t>(rc.w);
        const int a = static_cast<int>((static_cast<uint32_t>(rc.x) >> 31) | ((static_cast<uint32_t>(rc.y) >> 31) << 1));
        const int o[3] = {rc.x & 0x7fffffff, rc.y & 0x7fffffff, rc.z};
        int nd[NPE];
        bool fixed[NPE];
#pragma unroll
        for (int i = 0; i < NPE; ++i) {
            const int q = i < a ? i : (i > 0 ? i - 1 : 0);
            nd[i] = (i == a) ? static_cast<int>(n) : o[q];
            fixed[i] = ((slots >> (8 * i)) & 0xffu) == 0xffu;
        }
        double x[NPE], y[NPE], z[NPE];
#pragma unroll
        for (int i = 0; i < NPE; ++i) {
            x[i] = m.xyz[nd[i]];
            y[i] = m.xyz[m.nNode + nd[i]];
            z[i] = (KIND == PFEM_POISSON_TET) ? m.xyz[2 * m.nNode + nd[i]] : 0.0;
        }
        // this node's entries of the element: Klocal(a,:) for the lifting, Klocal(:,a) for the row
        double Krow[NPE], Kcol[NPE], f = 0.0;
        bool ok;
        if constexpr (KIND == PFEM_POISSON_TET) {
            bool any_fixed = false;
#pragma unroll
            for (int i = 0; i < NPE; ++i) any_fixed |= fixed[i];
#pragma unroll
            for (int i = 0; i < NPE; ++i) Krow[i] = 0.0;
            ok = poisson_tet_node_lean(x, y, z, prm.ed[0], prm.ed[1], prm.ed[2], prm.af, a, any_fixed, Kcol, Krow, f);
        } else {
            double K[NPE * NPE], F[NPE];
            if constexpr (KIND == PFEM_POISSON_TRIA)
                ok = poisson_tria(x, y, prm.ed[0], prm.ed[1], prm.af, valC, K, F);
            else
                ok = poisson_tria_inline(x, y, K, F);
#pragma unroll
            for (int i = 0; i < NPE; ++i)
                if (i == a) {
                    f = F[i];
#pragma unroll
                    for (int j = 0; j < NPE; ++j) { Krow[j] = K[i + NPE * j]; Kcol[j] = K[j + NPE * i]; }
                }
        }
        if (!ok) { atomicMax(err, PFEM_ERR_NEG_JAC); return; }
#pragma unroll
        for (int i = 0; i < NPE; ++i)          // Flocal(a) -= Klocal(a,i)*u_D(i)   (:859-870)
            if (fixed[i]) f = f - Krow[i] * m.soln[nd[i]];
        facc += f;                             // VecSetValues(ADD_VALUES) :880
#pragma unroll
        for (int j = 0; j < NPE; ++j) {        // entry (row, dof_j) += Klocal(j,a)   (:851, row-major read)
            if (fixed[j]) continue;
            const uint32_t k = (slots >> (8 * j)) & 0xffu;
            if (LDSACC) acc[k * T] += Kcol[j];
            else A.vals[base + (static_cast<int64_t>(k) << 6)] += Kcol[j];
        }
    }
    if (LDSACC)
        for (int k = 0; k < len; ++k) A.vals[base + (static_cast<int64_t>(k) << 6)] = acc[k * T];
    rhs[row] = facc;
}

// {x, y, z, solnApplied} of every node side by side (32 B): one sector per gathered node instead of three or four
__global__ void __launch_bounds__(kBlock) k_pack_node4(MeshDev m, double4 *__restrict__ out)
{
    const int64_t n = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (n >= m.nNode) return;
    out[n] = double4{m.xyz[n], m.xyz[m.nNode + n], m.xyz[2 * m.nNode + n], m.ndof == 1 ? m.soln[n] : 0.0};
}

// one visit of k_gather_poisson_tet4: the element's contribution to the node's row and right-hand side, the node being local
// node A of the element (self) and c0, c1, c2 the element's other three nodes in their local order
template <int A>
__device__ __forceinline__ bool gather_visit(const double4 &self, const double4 &c0, const double4 &c1, const double4 &c2, uint32_t slots,
                                             const ElemPrm &prm, double *acc, int T, double &facc)
{
    const double4 &n0 = A == 0 ? self : c0;
    const double4 &n1 = A == 1 ? self : (A < 1 ? c0 : c1);
    const double4 &n2 = A == 2 ? self : (A < 2 ? c1 : c2);
    const double4 &n3 = A == 3 ? self : c2;
    const double x[4] = {n0.x, n1.x, n2.x, n3.x}, y[4] = {n0.y, n1.y, n2.y, n3.y}, z[4] = {n0.z, n1.z, n2.z, n3.z};
    const double ud[4] = {n0.w, n1.w, n2.w, n3.w};
    bool fixed[4], any_fixed = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        fixed[i] = ((slots >> (8 * i)) & 0xffu) == 0xffu;
        any_fixed |= fixed[i];
    }
    double Krow[4] = {0.0, 0.0, 0.0, 0.0}, Kcol[4], f = 0.0;
    if (!poisson_tet_node_lean(x, y, z, prm.ed[0], prm.ed[1], prm.ed[2], prm.af, A, any_fixed, Kcol, Krow, f)) return false;
#pragma unroll
    for (int i = 0; i < 4; ++i)            // Flocal(a) -= Klocal(a,i)*u_D(i)   (:859-870)
        if (fixed[i]) f = f - Krow[i] * ud[i];
    facc += f;                             // VecSetValues(ADD_VALUES) :880
    // entry (row, dof_j) += Klocal(j,a)   (:851, row-major read).  The four slots of a visit are four different columns of the
    // row (the element's four nodes): read them all, then write them all -- one LDS round trip per visit instead of four
    // dependent ones; every slot still takes exactly one addition per visit, in element order: the same bits
    double cur[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) cur[j] = fixed[j] ? 0.0 : acc[((slots >> (8 * j)) & 0xffu) * T];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (!fixed[j]) acc[((slots >> (8 * j)) & 0xffu) * T] = cur[j] + Kcol[j];
    return true;
}

// k_gather_scalar<PFEM_POISSON_TET, true> with its two memory habits changed (same arithmetic, same order of additions):
// the coordinates and the Dirichlet value of a node come from ONE 32-B record (k_pack_node4) instead of four arrays
// -- the gathers, not the streamed incidence records, were what kept this kernel waiting: 9 scattered 8-B reads per visit
// -- and the next incidence record is already in flight while an element is evaluated.
__global__ void __launch_bounds__(kBlock) k_gather_poisson_tet4(int64_t nNode, SellDev A, double *rhs, ElemPrm prm,
                                                                 const int64_t *__restrict__ inc_ptr,
                                                                 const int32_t *__restrict__ inc_cnt,
                                                                 const int4 *__restrict__ inc_rec,
                                                                 const int32_t *__restrict__ node_row,
                                                                 const double4 *__restrict__ node4, int *err,
                                                                 unsigned xcd_per, const uint8_t *__restrict__ relk = nullptr,
                                                                 const int64_t *__restrict__ rslice_off = nullptr,
                                                                 double *__restrict__ rvals = nullptr,
                                                                 double *__restrict__ dinv_out = nullptr, double *__restrict__ ratio_out = nullptr,
                                                                 const VdHashEntry *__restrict__ vhash = nullptr, uint16_t *__restrict__ codes16 = nullptr,
                                                                 VdState *__restrict__ vstate = nullptr,
                                                                 const uint16_t *__restrict__ node_pat = nullptr, const int4 *__restrict__ pat_rec = nullptr,
                                                                 int pat_stride = 0)
{
    extern __shared__ __attribute__((aligned(16))) double lds_acc[];
    const int T = blockDim.x;
    const int64_t n = xcd_contiguous_block(blockIdx.x, xcd_per, xcd_per != 0) * T + threadIdx.x;
    if (n >= nNode) return;
    const int row = node_row[n];
    if (row < 0) return;
    const int cnt = inc_cnt[n];
    // the node's list: its own records (slot t of lane l at beg + 64 t), or -- pat_rec -- its pattern's, the other nodes relative to n
    const bool rel = pat_rec != nullptr;
    const int4 *__restrict__ rp = rel ? pat_rec + static_cast<int64_t>(node_pat[n]) * pat_stride : inc_rec + (inc_ptr[n >> 6] + (n & 63));
    const int rstep = rel ? 1 : 64;
    const auto other = [&](int word) -> int { return rel ? incpat_node(word, static_cast<int>(n)) : (word & 0x7fffffff); };
    const int64_t base = A.slice_off[row >> 6] + (row & 63);
    const int len = A.rowlen[row];
    double facc = 0.0;
    double *acc = lds_acc + threadIdx.x;
    for (int k = 0; k < len; ++k) acc[k * T] = 0.0;
    // Two visits ahead: while element t is evaluated, the node records of element t+1 and the incidence record of element
    // t+2 are in flight (the node loads depend on the incidence record, the arithmetic on the node loads).
    // Round 5 (SQ counters of round 4: VALU issuing 75 % of the cycles, ~110 of 194 VALU instructions per visit moves and
    // selects): (1) the node's OWN record is loaded once, a visit gathers its three other nodes only; (2) the visit's body is
    // compiled once per local index a of the node in the element (gather_visit<A>: which of the four positions is the node's
    // own is then known at compile time -- no select chains on a) and reached through a switch on a: on lattice numbering the
    // lanes of a wave walk translated copies of the same elements in the same order, so a is wave-uniform and three of the
    // four cases are skipped by a scalar branch; at boundaries the cases run one after the other under their lanes' masks.
    // Same arithmetic in the same order: K and F stay bit-identical to the serial loop.
    const double4 self = node4[n];
    int4 rc_next = cnt > 0 ? rp[0] : int4{0, 0, 0, 0};
    int4 rc_next2 = cnt > 1 ? rp[rstep] : int4{0, 0, 0, 0};
    double4 co[3];
    if (cnt > 0) {
        co[0] = node4[other(rc_next.x)];
        co[1] = node4[other(rc_next.y)];
        co[2] = node4[rel ? static_cast<int>(n) + rc_next.z : rc_next.z];
    } else {
        co[0] = co[1] = co[2] = double4{0.0, 0.0, 0.0, 0.0};
    }
    bool neg_jac = false;
    // (the slot of the row's own column -- the diagonal -- is the node's own slot in any of its visits: the first one's)
    const int kd = cnt > 0 ? static_cast<int>((static_cast<uint32_t>(rc_next.w) >>
                                               (8 * ((static_cast<uint32_t>(rc_next.x) >> 31) | ((static_cast<uint32_t>(rc_next.y) >> 31) << 1)))) & 0xffu) : -1;
    for (int v = 0; v < cnt; ++v) {
        const int4 rc = rc_next;
        const double4 c0 = co[0], c1 = co[1], c2 = co[2];
        rc_next = rc_next2;
        if (v + 1 < cnt) {
            co[0] = node4[other(rc_next.x)];
            co[1] = node4[other(rc_next.y)];
            co[2] = node4[rel ? static_cast<int>(n) + rc_next.z : rc_next.z];
            if (v + 2 < cnt) rc_next2 = rp[static_cast<int64_t>(v + 2) * rstep];
        }
        const uint32_t slots = static_cast<uint32_t>(rc.w);
        const int a = static_cast<int>((static_cast<uint32_t>(rc.x) >> 31) | ((static_cast<uint32_t>(rc.y) >> 31) << 1));
        bool ok = true;
        switch (a) {
        case 0: ok = gather_visit<0>(self, c0, c1, c2, slots, prm, acc, T, facc); break;
        case 1: ok = gather_visit<1>(self, c0, c1, c2, slots, prm, acc, T, facc); break;
        case 2: ok = gather_visit<2>(self, c0, c1, c2, slots, prm, acc, T, facc); break;
        default: ok = gather_visit<3>(self, c0, c1, c2, slots, prm, acc, T, facc); break;
        }
        neg_jac = neg_jac || !ok;
    }
    if (neg_jac) {
        atomicMax(err, PFEM_ERR_NEG_JAC);
        return;
    }
    if (rvals) {
        // both forms: the row form, and the relative-group copy the CG's SpMV streams (k_spmvr: group g = row / 4, plane row % 4,
        // union entry relk[slot] of the group's slice -- the explicit zeros of that copy were set when the map was built)
        const int64_t g = row >> 2;
        double *rp = rvals + 4 * rslice_off[g >> 6] + (g & 63) + 64 * (row & 3);          // (4 = kRelRows, asserted at k_rel_slot_map)
        for (int k = 0; k < len; ++k) {
            const int64_t q = base + (static_cast<int64_t>(k) << 6);
            const double v = acc[k * T];
            A.vals[q] = v;
            rp[4LL * 64 * relk[q]] = v;
        }
    } else if (codes16) {
        // the row form, and the relative-group form's CODES (pfem_vdhash.hpp: the dictionary of the last step through its hash
        // table; word = entry of the group's slice, the row's 16 bits at plane row % 4): no fp64 copy of that form, no encode pass
        const int64_t g = row >> 2;
        uint16_t *cp = codes16 + 4 * (rslice_off[g >> 6] + (g & 63)) + (row & 3);
        bool missed = false;
        for (int k = 0; k < len; ++k) {
            const int64_t q = base + (static_cast<int64_t>(k) << 6);
            const double v = acc[k * T];
            A.vals[q] = v;
            int code = vd_hash_find(vhash, static_cast<unsigned long long>(__double_as_longlong(v)));
            if (code < 0) { missed = true; code = 0; }
            cp[4LL * 64 * relk[q]] = static_cast<uint16_t>(code);
        }
        if (missed) vstate->miss = 1;
    } else
    for (int k = 0; k < len; ++k) A.vals[base + (static_cast<int64_t>(k) << 6)] = acc[k * T];
    rhs[row] = facc;
    if (dinv_out) {
        // what k_amg_diag_bound computes from the stored row -- the inverse diagonal and the row's share of the Gershgorin bound
        // max_i sum_j |a_ij| / a_ii, the absolute values added in slot order -- while the row is still in LDS: the multigrid's
        // numeric phase then does not read the assembled matrix (1.4 GB at config 3) for them.  One rank only (no ghost columns).
        double sabs = 0.0;
        for (int k = 0; k < len; ++k) sabs += fabs(acc[k * T]);
        const double d = kd >= 0 && kd < len ? acc[kd * T] : 0.0;
        const bool pos = d > 0.0;
        dinv_out[row] = pos ? 1.0 / d : 1.0;
        ratio_out[row] = pos ? sabs / d : 1.0;
    }
}

// Elasticity gather, one thread per (node, dof) ROW: the row is accumulated in LDS (entry k of
// thread t at acc[k*T + t]: conflict free; T = blockDim.x chosen so maxlen*T doubles fit) and
// stored once, coalesced -- no read-modify-write of the matrix in global memory (what bounded
// the thread-per-node form: 5.0 ms on the 50x300x50 beam).  The three threads of a node walk the same packed incidence list and
// recompute the element geometry; same additions in the same (ascending element) order.
__global__ void __launch_bounds__(kBlock) k_gather_elast_rows(MeshDev m, SellDev A, double *rhs, ElemPrm prm,
                                                               const int64_t *__restrict__ inc_ptr,
                                                               const int32_t *__restrict__ inc_cnt,
                                                               const int4 *__restrict__ inc_rec,
                                                               const uint16_t *__restrict__ inc_flags,
                                                               const int32_t *__restrict__ node_row, int *err, unsigned xcd_per,
                                                               const int32_t *__restrict__ row_group = nullptr,
                                                               const int32_t *__restrict__ group_row0 = nullptr,
                                                               const int64_t *__restrict__ gslice_off = nullptr,
                                                               double *__restrict__ gvals = nullptr,
                                                               const uint16_t *__restrict__ node_pat = nullptr, const int4 *__restrict__ pat_rec = nullptr,
                                                               const uint16_t *__restrict__ pat_flags = nullptr, int pat_stride = 0)
{
    extern __shared__ __attribute__((aligned(16))) double lds_acc[];
    const int T = blockDim.x;
    const int64_t tid = xcd_contiguous_block(blockIdx.x, xcd_per, xcd_per != 0) * T + threadIdx.x;
    if (tid >= 3 * m.nNode) return;
    const int64_t n = tid / 3;
    const int p = static_cast<int>(tid - 3 * n);
    const int row = node_row[tid];
    if (row < 0) return;
    const int cnt = inc_cnt[n];
    // the node's list: its own records, or its pattern's with the other nodes relative to n (k_incpat_*)
    const bool rel = pat_rec != nullptr;
    const int64_t beg = rel ? static_cast<int64_t>(node_pat[n]) * pat_stride : inc_ptr[n >> 6] + (n & 63);
    const int4 *__restrict__ rp = (rel ? pat_rec : inc_rec) + beg;
    const uint16_t *__restrict__ fp = (rel ? pat_flags : inc_flags) + beg;
    const int rstep = rel ? 1 : 64;
    const int64_t base = A.slice_off[row >> 6] + (row & 63);
    const int len = A.rowlen[row];
    double *acc = lds_acc + threadIdx.x;
    for (int k = 0; k < len; ++k) acc[k * T] = 0.0;
    const ElastMat mat = elast_material(prm.ed[0], prm.ed[1]);
    const double bf = p == 0 ? prm.ed[3] : (p == 1 ? prm.ed[4] : prm.ed[5]);
    double facc = 0.0;
    for (int v = 0; v < cnt; ++v) {
        const int4 rc = rp[static_cast<int64_t>(v) * rstep];
        const uint32_t flags = fp[static_cast<int64_t>(v) * rstep];
        const uint32_t slots = static_cast<uint32_t>(rc.w);
        const int a = static_cast<int>((static_cast<uint32_t>(rc.x) >> 31) | ((static_cast<uint32_t>(rc.y) >> 31) << 1));
        const int o[3] = {rel ? incpat_node(rc.x, static_cast<int>(n)) : (rc.x & 0x7fffffff),
                          rel ? incpat_node(rc.y, static_cast<int>(n)) : (rc.y & 0x7fffffff), rel ? static_cast<int>(n) + rc.z : rc.z};
        int nd[4];
        double x[4], y[4], z[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = i < a ? i : (i > 0 ? i - 1 : 0);
            nd[i] = (i == a) ? static_cast<int>(n) : o[q];
            x[i] = m.xyz[nd[i]];
            y[i] = m.xyz[m.nNode + nd[i]];
            z[i] = m.xyz[2 * m.nNode + nd[i]];
        }
        TetGeom g;
        tet_geometry_lean(x, y, z, g);    // zero signs aside, the literal geometry (pfem_elem.hpp)
        if (g.jac < 0.0) { atomicMax(err, PFEM_ERR_NEG_JAC); return; }
        const double dvol = kGaussWtTet * g.jac;
        double ax = g.gx[0], ay = g.gy[0], az = g.gz[0];
#pragma unroll
        for (int i = 1; i < 4; ++i)
            if (i == a) { ax = g.gx[i]; ay = g.gy[i]; az = g.gz[i]; }
        const double b4 = dvol * 0.25;
        double f = 0.0 + b4 * bf;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const uint32_t fb = (flags >> (3 * b)) & 7u;
            if (fb != 0) {                 // lifting, q ascending:  Flocal(3a+p) -= Klocal(3a+p,3b+q) * u_D(3b+q)
                double kab[3][3];
                elast_block_v(ax, ay, az, g.gx[b], g.gy[b], g.gz[b], mat, dvol, kab);
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    if (fb & (1u << q)) {
                        const double kpq = p == 0 ? kab[0][q] : (p == 1 ? kab[1][q] : kab[2][q]);
                        f = f - kpq * m.soln[3LL * nd[b] + q];
                    }
                if (fb == 7u) continue;
            }
            double kba[3][3];
            elast_block_v(g.gx[b], g.gy[b], g.gz[b], ax, ay, az, mat, dvol, kba);   // Klocal(3b+q, 3a+p) = kba[q][p]
            // entry (row, dof(3b+q)) += Klocal(3b+q, 3a+p); node b's free dofs are consecutive columns
            int k = static_cast<int>((slots >> (8 * b)) & 0xffu);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                if (fb & (1u << q)) continue;
                acc[k * T] += p == 0 ? kba[q][0] : (p == 1 ? kba[q][1] : kba[q][2]);
                ++k;
            }
        }
        facc += f;
    }
    if (gvals) {
        // both forms: the row form, and the node-group copy the CG's SpMV streams (k_spmvg: group = the node's rows, plane =
        // row - first row of the group; the zero padding of that copy was set when the groups were built)
        const int64_t g = row_group[row];
        double *gp = gvals + 3 * gslice_off[g >> 6] + (g & 63) + 64 * (row - group_row0[g]);          // (3 = kGroupRows)
        for (int k = 0; k < len; ++k) {
            const double v = acc[k * T];
            A.vals[base + (static_cast<int64_t>(k) << 6)] = v;
            gp[3LL * 64 * k] = v;
        }
    } else
    for (int k = 0; k < len; ++k) A.vals[base + (static_cast<int64_t>(k) << 6)] = acc[k * T];
    rhs[row] = facc;
}

// ---------------------------------------------------------------------------
// plane-stress elasticity on P1 triangles (2 dofs per node, next row 8f.1): same two
// formulations as the tetrahedron kernels, 2x2 node blocks from elast2d_block_v
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_assemble_elast2d(MeshDev m, SellDev A, double *rhs, ElemPrm prm, int *err,
                                                              const uint8_t *__restrict__ only_nodes)
{
    const int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (e >= m.nElem) return;
    int nd[3], dof[6];
    double x[3], y[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) nd[a] = m.conn[a * m.nElem + e];
    if (only_nodes && !(only_nodes[nd[0]] | only_nodes[nd[1]] | only_nodes[nd[2]])) return;   // hub pass
#pragma unroll
    for (int i = 0; i < 6; ++i) dof[i] = m.edof[i * m.nElem + e];
#pragma unroll
    for (int a = 0; a < 3; ++a) { x[a] = m.xyz[nd[a]]; y[a] = m.xyz[m.nNode + nd[a]]; }
    TriaGeom g;
    tria_geometry(x, y, g);
    if (g.jac < 0.0) { atomicMax(err, PFEM_ERR_NEG_JAC); return; }
    const double dvol = 0.5 * (g.jac * prm.ed[2]);
    const Elast2dMat mat = elast2d_material(prm.ed[0], prm.ed[1]);
    double N[3], F[6], fact[6];
    tria_shape_gp(N);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double b4 = dvol * N[a];
        F[2 * a] = 0.0 + b4 * prm.ed[3];
        F[2 * a + 1] = 0.0 + b4 * prm.ed[4];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) fact[i] = dof[i] < 0 ? m.soln[2LL * nd[i / 2] + i % 2] : 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            double blk[2][2];
            elast2d_block_v(g.gx[a], g.gy[a], g.gx[b], g.gy[b], mat, dvol, blk);   // Klocal(2a+p, 2b+q)
#pragma unroll
            for (int q = 0; q < 2; ++q)
                if (dof[2 * b + q] < 0) {
#pragma unroll
                    for (int p = 0; p < 2; ++p)
                        if (dof[2 * a + p] >= 0) F[2 * a + p] = F[2 * a + p] - blk[p][q] * fact[2 * b + q];
                }
            // MatSetValues row-major read: entry (row = dof(2b+q), col = dof(2a+p)) += Klocal(2a+p,2b+q)
            const int firstp = dof[2 * a] >= 0 ? 0 : (dof[2 * a + 1] >= 0 ? 1 : -1);
            if (firstp < 0) continue;
            if (only_nodes && !only_nodes[nd[b]]) continue;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int row = dof[2 * b + q];
                if (row < 0) continue;
                int64_t s = find_slot(A, row, dof[2 * a + firstp]);
                if (s < 0) { atomicMax(err, PFEM_ERR_PATTERN); continue; }
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    if (dof[2 * a + p] < 0) continue;
                    add_f64(&A.vals[s], blk[p][q]);
                    s += 64;
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i)
        if (dof[i] >= 0 && (!only_nodes || only_nodes[nd[i / 2]])) add_f64(&rhs[dof[i]], F[i]);
}

// Plane-stress sibling of k_gather_elast_rows: one thread per (node, dof) row, 2x2 node blocks.
__global__ void __launch_bounds__(kBlock) k_gather_elast2d_rows(MeshDev m, SellDev A, double *rhs, ElemPrm prm,
                                                                 const int64_t *__restrict__ inc_ptr,
                                                                 const int32_t *__restrict__ inc_cnt,
                                                                 const int4 *__restrict__ inc_rec,
                                                                 const uint16_t *__restrict__ inc_flags,
                                                                 const int32_t *__restrict__ node_row, int *err)
{
    extern __shared__ __attribute__((aligned(16))) double lds_acc[];
    const int T = blockDim.x;
    const int64_t tid = static_cast<int64_t>(blockIdx.x) * T + threadIdx.x;
    if (tid >= 2 * m.nNode) return;
    const int64_t n = tid >> 1;
    const int p = static_cast<int>(tid & 1);
    const int row = node_row[tid];
    if (row < 0) return;
    const int cnt = inc_cnt[n];
    const int64_t beg = inc_ptr[n >> 6] + (n & 63), end = beg + 64LL * cnt;
    const int64_t base = A.slice_off[row >> 6] + (row & 63);
    const int len = A.rowlen[row];
    double *acc = lds_acc + threadIdx.x;
    for (int k = 0; k < len; ++k) acc[k * T] = 0.0;
    const Elast2dMat mat = elast2d_material(prm.ed[0], prm.ed[1]);
    double N[3];
    tria_shape_gp(N);
    const double bf = p == 0 ? prm.ed[3] : prm.ed[4];
    double facc = 0.0;
    for (int64_t t = beg; t < end; t += 64) {
        const int4 rc = inc_rec[t];
        const uint32_t flags = inc_flags[t];
        const uint32_t slots = static_cast<uint32_t>(rc.w);
        const int a = static_cast<int>((static_cast<uint32_t>(rc.x) >> 31) | ((static_cast<uint32_t>(rc.y) >> 31) << 1));
        const int o0 = rc.x & 0x7fffffff, o1 = rc.y & 0x7fffffff, me = static_cast<int>(n);
        const int nd[3] = {a == 0 ? me : o0, a == 0 ? o0 : (a == 1 ? me : o1), a == 2 ? me : o1};
        double x[3], y[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            x[i] = m.xyz[nd[i]];
            y[i] = m.xyz[m.nNode + nd[i]];
        }
        TriaGeom g;
        tria_geometry(x, y, g);
        if (g.jac < 0.0) { atomicMax(err, PFEM_ERR_NEG_JAC); return; }
        const double dvol = 0.5 * (g.jac * prm.ed[2]);
        double ax = g.gx[0], ay = g.gy[0], na = N[0];
#pragma unroll
        for (int i = 1; i < 3; ++i)
            if (i == a) { ax = g.gx[i]; ay = g.gy[i]; na = N[i]; }
        const double b4 = dvol * na;
        double f = 0.0 + b4 * bf;
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const uint32_t fb = (flags >> (2 * b)) & 3u;
            if (fb != 0) {                 // lifting, q ascending:  Flocal(2a+p) -= Klocal(2a+p,2b+q) * u_D(2b+q)
                double kab[2][2];
                elast2d_block_v(ax, ay, g.gx[b], g.gy[b], mat, dvol, kab);
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    if (fb & (1u << q)) f = f - (p == 0 ? kab[0][q] : kab[1][q]) * m.soln[2LL * nd[b] + q];
                if (fb == 3u) continue;
            }
            double kba[2][2];
            elast2d_block_v(g.gx[b], g.gy[b], ax, ay, mat, dvol, kba);   // Klocal(2b+q, 2a+p) = kba[q][p]
            int k = static_cast<int>((slots >> (8 * b)) & 0xffu);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (fb & (1u << q)) continue;
                acc[k * T] += p == 0 ? kba[q][0] : kba[q][1];
                ++k;
            }
        }
        facc += f;
    }
    for (int k = 0; k < len; ++k) A.vals[base + (static_cast<int64_t>(k) << 6)] = acc[k * T];
    rhs[row] = facc;
}

// Parity inspection: Ke/Fe of every element exactly as the assembly kernels compute them.
__global__ void __launch_bounds__(kBlock) k_eval_elems(MeshDev m, ElemPrm prm, double *Kout, double *Fout, int *err)
{
    const int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (e >= m.nElem) return;
    double x[4], y[4], z[4];
    for (int a = 0; a < m.npe; ++a) {
        const int n = m.conn[a * m.nElem + e];
        x[a] = m.xyz[n];
        y[a] = m.xyz[m.nNode + n];
        z[a] = m.ndim == 3 ? m.xyz[2 * m.nNode + n] : 0.0;
    }
    const double valC[4] = {0.0, 0.0, 0.0, 0.0};
    double *K = Kout + e * m.nsize * m.nsize, *F = Fout + e * m.nsize;
    bool ok = false;
    switch (m.kind) {
    case PFEM_POISSON_TET: ok = poisson_tet(x, y, z, prm.ed[0], prm.ed[1], prm.ed[2], prm.af, valC, K, F); break;
    case PFEM_POISSON_TRIA: ok = poisson_tria(x, y, prm.ed[0], prm.ed[1], prm.af, valC, K, F); break;
    case PFEM_POISSON_TRIA_INLINE: ok = poisson_tria_inline(x, y, K, F); break;
    case PFEM_ELAST_TET: {
        const double bf[3] = {prm.ed[3], prm.ed[4], prm.ed[5]};
        ok = elast_tet(x, y, z, prm.ed[0], prm.ed[1], bf, K, F);
        break;
    }
    case PFEM_ELAST_TRIA: {
        const double bf[2] = {prm.ed[3], prm.ed[4]};
        ok = elast_tria(x, y, prm.ed[0], prm.ed[1], prm.ed[2], bf, K, F);
        break;
    }
    }
    if (!ok) atomicMax(err, PFEM_ERR_NEG_JAC);
}

// ---------------------------------------------------------------------------
// matrix utilities
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_sell_to_csr(SellDev A, const int64_t *rowptr, int32_t *cols, double *vals)
{
    const int64_t r = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (r >= A.n_rows) return;
    const int64_t base = A.slice_off[r >> 6] + (r & 63);
    const int64_t p0 = rowptr[r];
    const int len = A.rowlen[r];
    for (int k = 0; k < len; ++k) {
        if (cols) cols[p0 + k] = A.cols[base + 64LL * k];
        if (vals) vals[p0 + k] = A.vals[base + 64LL * k];
    }
}

__global__ void __launch_bounds__(kBlock) k_csr_vals_to_sell(SellDev A, const int64_t *rowptr, const double *vals)
{
    const int64_t r = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (r >= A.n_rows) return;
    const int64_t base = A.slice_off[r >> 6] + (r & 63);
    const int64_t p0 = rowptr[r];
    const int len = A.rowlen[r];
    for (int k = 0; k < len; ++k) A.vals[base + 64LL * k] = vals[p0 + k];
}

__global__ void __launch_bounds__(kBlock) k_extract_diag(SellDev A, double *diag)
{
    const int64_t r = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (r >= A.n_rows) return;
    const int64_t s = find_slot(A, static_cast<int>(r), static_cast<int>(r));
    diag[r] = s >= 0 ? A.vals[s] : 0.0;
}

// rhs[idx[i]] += v[i]  (nodal forces; indices are distinct dofs)
__global__ void __launch_bounds__(kBlock) k_add_values(double *rhs, const int32_t *idx, const double *v, int64_t n)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i < n) add_f64(&rhs[idx[i]], v[i]);
}

__global__ void __launch_bounds__(kBlock) k_invert(double *d, int64_t n)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i < n) d[i] = 1.0 / d[i];
}

// ---------------------------------------------------------------------------
// SpMV  y = A x  on the wave-sliced storage: one lane per row, one wave per slice, one
// block per group of four consecutive slices (no grid-stride).
// The roofline kernel: per row it streams 12 B per stored entry (f64 value + int32
// column), gathers x through L1/L2 (banded reuse) and writes 8 B.
//
// Measured choices (tools/spmv_lab.py, 200^3 Poisson, MI355X):
//   * plain block->slice order beats an XCD-contiguous remap (5.6 vs 4.6 TB/s): the
//     stream is read once, so spreading every XCD over the whole address range
//     balances the HBM channels, while x is small enough to live in L2/MALL anyway;
//   * one slice group per block beats a 2048-block grid-stride loop (+7 %);
//   * matrix values/columns are touched exactly once -> nontemporal loads (+3 %),
//     keeping L2 for the x gather.
// With WITH_DOT the kernel also emits the per-block partial of x.y over rows < n_dot
// (CG's (p, Ap)); k_reduce_partials turns the partials into one scalar, in fixed order.
// ---------------------------------------------------------------------------
// Which slices a SpMV launch works on: all of them in order (list == nullptr), or the `count` slices named in
// `list` (multi-GPU: the slices that hold shared rows run first, the interior ones under the neighbour exchange).
struct SliceSel {
    const int32_t *list;
    int64_t count;
};
__device__ inline int64_t pick_slice(const SliceSel &sel, int64_t idx, int64_t n_all)
{
    if (!sel.list) return idx;
    return idx < sel.count ? static_cast<int64_t>(sel.list[idx]) : n_all;
}

struct CgCtl {            // device-resident control block of the CG iteration
    double beta[2];       // (r,z) ping-pong by iteration parity
    double rn0, ttol;     // ||z0||, max(rtol*rn0, abstol)
    double rn;            // last preconditioned residual norm
    double dtol;
    double alpha;         // step length of the current iteration (k_cg_update -> k_cg_direction)
    int flag;             // 0 = running, else KSPConvergedReason   } one aligned 8-byte word, published with a single
    int its;              // iterations completed                     } store by the direction kernel (ctl_publish)
    int its_dir;          // iteration index handed from k_cg_update to k_cg_direction (graph launches carry no `it`);
                          // -2 once the update kernel has seen the solve finished
    int pad_;
    double dpi[2];        // single-reduction form: (p,Ap) by recurrence, ping-pong by iteration parity
    unsigned gbar;        // arrival counter of the grid barrier in k_pc_post_dots_direction (zeroed with the block at the start of a solve)
    unsigned gpad_;
};
static_assert(offsetof(CgCtl, flag) % 8 == 0 && offsetof(CgCtl, its) == offsetof(CgCtl, flag) + 4, "flag/its share a word");

// The direction kernel is the only one that writes `flag` in a launch whose other blocks still read it.  A block
// that starts after the lead thread has published the verdict of THIS iteration must not take it for "finished
// before this launch" (it would skip its share of the final x += alpha p): {flag, its} travel in one word and a
// block leaves early only when the verdict belongs to an earlier iteration.
__device__ inline void ctl_publish(CgCtl *ctl, int flag, int its)
{
    const unsigned long long w = static_cast<unsigned long long>(static_cast<unsigned>(flag)) |
                                 (static_cast<unsigned long long>(static_cast<unsigned>(its)) << 32);
    // agent scope: the readers are blocks of this device (the host reads the block after a stream sync)
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(&ctl->flag), w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline bool ctl_finished_before(const CgCtl *ctl, int it)
{
    const unsigned long long w = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(&ctl->flag), __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT);
    return static_cast<int>(w & 0xffffffffu) != 0 && static_cast<int>(w >> 32) != it + 1;
}

template <bool WITH_DOT>
__global__ void __launch_bounds__(kBlock) k_spmv(SellDev A, const double *__restrict__ x, double *__restrict__ y,
                                                  int64_t n_dot, double *partial, const CgCtl *ctl, SliceSel sel)
{
    __shared__ double sm[4];
    if (WITH_DOT && ctl->flag != 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t s = pick_slice(sel, (static_cast<int64_t>(blockIdx.x) << 2) + wave, A.n_slices);
    double dot = 0.0;
    if (s < A.n_slices) {
        const int64_t off = A.slice_off[s];
        const int width = static_cast<int>((A.slice_off[s + 1] - off) >> 6);
        const int32_t *__restrict__ cp = A.cols + off + lane;
        const double *__restrict__ vp = A.vals + off + lane;
        double acc = 0.0;
        int k = 0;
        for (; k + 4 <= width; k += 4) {
            int c[4];
            double v[4], xv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                c[j] = __builtin_nontemporal_load(cp + 64 * (k + j));
                v[j] = __builtin_nontemporal_load(vp + 64 * (k + j));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) xv[j] = x[c[j]];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_fma(v[j], xv[j], acc);
        }
        for (; k < width; ++k)
            acc = __builtin_fma(__builtin_nontemporal_load(vp + 64 * k), x[__builtin_nontemporal_load(cp + 64 * k)], acc);
        const int64_t row = (s << 6) + lane;
        if (row < A.n_rows) {
            y[row] = acc;
            if (WITH_DOT && row < n_dot) dot = x[row] * acc;
        }
    }
    if (WITH_DOT) {
        const double t = block_sum(dot, sm);
        if (threadIdx.x == 0) partial[blockIdx.x] = t;
    }
}

// ---------------------------------------------------------------------------
// SpMV with 16-bit column deltas.  Columns are ascending inside a row, so a row is stored as
// its first column (int32, col0[row]) plus unsigned 16-bit gaps to the next column, two gaps per
// 32-bit word, in the same wave-sliced order as the values: word j of lane l of slice s sits
// at slice_doff[s] + 64*j + l and holds gap(2j+1) | gap(2j+2) << 16.  Padding entries carry gap
// 0 (and value 0).  A Poisson row of 15 entries costs 4 + 28 = 32 B of index data instead of
// 60 B: the SpMV streams 152 instead of 180 B per row.  Used when every gap fits 16 bits
// (structured cfg 3: max gap 39 202; checked when the pattern is built), otherwise the int32
// kernel above runs.  Same products, same summation order -> bit-identical y.
// ---------------------------------------------------------------------------
// Dictionary form of a 16-bit gap stream, for patterns with gaps beyond 65535 (numbering planes of more than 65 535
// dofs: a slab of config 5, an elasticity mesh of more than 21 845 nodes per plane) that have FEW DISTINCT large gaps
// -- any regularly numbered mesh has a handful: gaps below 32768 are stored as they are, every other gap as
// 0x8000 | its index in a per-matrix table of at most kGapTable values, which every SpMV block copies to LDS.  Same
// bytes per nonzero as the literal 16-bit form; more distinct gaps than the table holds: 32-bit gaps (relative row
// groups) or int32 columns (row forms).
constexpr int kGapTable = 256;
enum RelGapMode { kGapLit16 = 0, kGap32 = 1, kGapDict16 = 2 };

__device__ inline void gap_table_insert(uint32_t *tbl, uint32_t gap, int *overflow)
{
    for (int i = 0; i < kGapTable; ++i) {
        const uint32_t cur = __hip_atomic_load(&tbl[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == gap) return;
        if (cur == 0u) {
            const uint32_t old = atomicCAS(&tbl[i], 0u, gap);
            if (old == 0u || old == gap) return;
        }
    }
    atomicOr(overflow, 4);       // more distinct large gaps than the table holds
}
// Every large gap of a pattern is put into the table by the walk that precedes the fill (k_row_gap_table /
// k_rel_gap_table).  Should the two walks ever disagree, a gap would be missing here: that is flagged (`miss`), and the
// builder drops the dictionary form instead of shipping a zero gap, i.e. wrong columns.
__device__ inline uint32_t gap_table_code(const uint32_t *tbl, uint32_t gap, int *miss)
{
    if (gap < 0x8000u) return gap;
    for (int i = 0; i < kGapTable; ++i)
        if (tbl[i] == gap) return 0x8000u | static_cast<uint32_t>(i);
    atomicMax(miss, 1);
    return 0u;
}

// a 16-bit code of a gap stream -> the gap
template <bool DICT>
__device__ __forceinline__ int gap16(uint32_t code, const uint32_t *tbl)
{
    if constexpr (DICT) return (code & 0x8000u) ? static_cast<int>(tbl[code & (kGapTable - 1)]) : static_cast<int>(code);
    else return static_cast<int>(code);
}

struct Sell16Dev {
    const int32_t *col0;        // [64 * n_slices]
    const uint32_t *dwords;     // packed gaps
    const int64_t *slice_doff;  // [n_slices+1], in words
    const uint32_t *gap_table;  // dictionary form (see above), else unused
};

// words per slice = 64 * ceil((width-1)/2); one thread per slice
__global__ void __launch_bounds__(kBlock) k_cols16_sizes(const int64_t *slice_off, int64_t n_slices, int64_t *slice_words)
{
    const int64_t s = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (s > n_slices) return;
    if (s == n_slices) { slice_words[s] = 0; return; }
    const int width = static_cast<int>((slice_off[s + 1] - slice_off[s]) >> 6);
    slice_words[s] = 64LL * (width > 1 ? (width / 2) : 0);     // ceil((width-1)/2) == width/2
}

// 32-bit gaps (relative row groups whose offsets are further apart than 65535): one word per entry after the first
__global__ void __launch_bounds__(kBlock) k_gap32_sizes(const int64_t *slice_off, int64_t n_slices, int64_t *slice_words)
{
    const int64_t s = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (s > n_slices) return;
    if (s == n_slices) { slice_words[s] = 0; return; }
    const int width = static_cast<int>((slice_off[s + 1] - slice_off[s]) >> 6);
    slice_words[s] = 64LL * (width > 1 ? width - 1 : 0);
}

// one thread per row: first column + packed gaps; *overflow is set if a gap needs > 16 bits
template <bool DICT>
__global__ void __launch_bounds__(kBlock) k_cols16_fill(SellDev A, const int64_t *slice_doff, int32_t *col0,
                                                         uint32_t *dwords, int *overflow, const uint32_t *gap_table)
{
    const int64_t r = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    const int64_t s = r >> 6;
    if (s >= A.n_slices) return;
    const int lane = static_cast<int>(r & 63);
    const int64_t off = A.slice_off[s];
    const int width = static_cast<int>((A.slice_off[s + 1] - off) >> 6);
    const int len = r < A.n_rows ? A.rowlen[r] : 0;
    const int32_t *cp = A.cols + off + lane;
    int prev = len > 0 ? cp[0] : 0;
    col0[r] = prev;
    uint32_t *wp = dwords + slice_doff[s] + lane;
    bool bad = false;
    for (int j = 0; 2 * j + 1 < width; ++j) {
        uint32_t w = 0;
        for (int h = 0; h < 2; ++h) {
            const int k = 2 * j + 1 + h;
            uint32_t gap = 0;
            if (k < len) {
                const int c = cp[64 * k];
                const int64_t g = static_cast<int64_t>(c) - prev;
                if (g < 0 || (!DICT && g > 65535)) bad = true;
                gap = (DICT ? gap_table_code(gap_table, static_cast<uint32_t>(g), overflow) : static_cast<uint32_t>(g)) & 0xffffu;
                prev = c;
            }
            w |= gap << (16 * h);
        }
        wp[64LL * j] = w;
    }
    if (bad) atomicMax(overflow, 1);
}

// the distinct gaps of 32768 and more between consecutive columns of a row (patterns whose literal 16-bit form overflowed)
__global__ void __launch_bounds__(kBlock) k_row_gap_table(SellDev A, uint32_t *tbl, int *overflow)
{
    const int64_t r = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (r >= A.n_rows) return;
    const int len = A.rowlen[r];
    const int32_t *cp = A.cols + A.slice_off[r >> 6] + (r & 63);
    for (int k = 1; k < len; ++k) {
        const int64_t g = static_cast<int64_t>(cp[64LL * k]) - cp[64LL * (k - 1)];
        if (g >= 0x8000) gap_table_insert(tbl, static_cast<uint32_t>(g), overflow);
    }
}

// W words = 2W consecutive entries (all inside the slice: caller guarantees 2*(j+W) < width)
template <int W, bool DICT>
__device__ __forceinline__ void spmv16_trip(const double *__restrict__ vp, const uint32_t *__restrict__ wp,
                                            const double *__restrict__ x, int &j, int &c, double &acc, const uint32_t *tbl)
{
    uint32_t w[W];
    double v[2 * W], xv[2 * W];
#pragma unroll
    for (int t = 0; t < W; ++t) w[t] = __builtin_nontemporal_load(wp + 64 * (j + t));
#pragma unroll
    for (int t = 0; t < 2 * W; ++t) v[t] = __builtin_nontemporal_load(vp + 64 * (2 * j + 1 + t));
    int cc = c;
#pragma unroll
    for (int t = 0; t < W; ++t) {
        cc += gap16<DICT>(w[t] & 0xffffu, tbl);
        xv[2 * t] = x[cc];
        cc += gap16<DICT>(w[t] >> 16, tbl);
        xv[2 * t + 1] = x[cc];
    }
    // (no scheduling barrier here: measured 252 -> 266 us; it pays in the multi-row kernels below)
#pragma unroll
    for (int t = 0; t < 2 * W; ++t) acc = __builtin_fma(v[t], xv[t], acc);
    c = cc;
    j += W;
}

template <bool WITH_DOT, bool DICT = false>
__global__ void __launch_bounds__(kBlock) k_spmv16(SellDev A, Sell16Dev C, const double *__restrict__ x,
                                                    double *__restrict__ y, int64_t n_dot, double *partial,
                                                    const CgCtl *ctl, SliceSel sel)
{
    __shared__ double sm[4];
    __shared__ uint32_t tbl[DICT ? kGapTable : 1];
    if (WITH_DOT && ctl->flag != 0) return;
    if (DICT) {
        static_assert(kGapTable == kBlock, "one table entry per thread");
        tbl[DICT ? threadIdx.x : 0] = C.gap_table[threadIdx.x];
        __syncthreads();
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t s = pick_slice(sel, (static_cast<int64_t>(blockIdx.x) << 2) + wave, A.n_slices);
    double dot = 0.0;
    if (s < A.n_slices) {
        const int64_t off = A.slice_off[s];
        const int width = static_cast<int>((A.slice_off[s + 1] - off) >> 6);
        const double *__restrict__ vp = A.vals + off + lane;
        const uint32_t *__restrict__ wp = C.dwords + C.slice_doff[s] + lane;
        int c = __builtin_nontemporal_load(C.col0 + (s << 6) + lane);
        double acc = 0.0;
        if (width > 0) acc = __builtin_nontemporal_load(vp) * x[c];
        const int nw = width / 2;          // words: entries 1 .. 2*nw (the last one may be a pad)
        int j = 0;
        // cascade of trip sizes: all streaming loads of a trip are issued before the first use, so
        // a wave keeps up to 8 words + 16 values (+16 gathers) in flight
        while (2 * (j + 8) < width) spmv16_trip<8, DICT>(vp, wp, x, j, c, acc, tbl);
        if (2 * (j + 4) < width) spmv16_trip<4, DICT>(vp, wp, x, j, c, acc, tbl);
        if (2 * (j + 2) < width) spmv16_trip<2, DICT>(vp, wp, x, j, c, acc, tbl);
        for (; j < nw; ++j) {
            const uint32_t w0 = __builtin_nontemporal_load(wp + 64 * j);
            const int c0 = c + gap16<DICT>(w0 & 0xffffu, tbl);
            const int c1 = c0 + gap16<DICT>(w0 >> 16, tbl);
            acc = __builtin_fma(__builtin_nontemporal_load(vp + 64 * (2 * j + 1)), x[c0], acc);
            if (2 * j + 2 < width) acc = __builtin_fma(__builtin_nontemporal_load(vp + 64 * (2 * j + 2)), x[c1], acc);
            c = c1;
        }
        const int64_t row = (s << 6) + lane;
        if (row < A.n_rows) {
            y[row] = acc;
            if (WITH_DOT && row < n_dot) dot = x[row] * acc;
        }
    }
    if (WITH_DOT) {
        const double t = block_sum(dot, sm);
        if (threadIdx.x == 0) partial[blockIdx.x] = t;
    }
}

// The same with ESCAPES: a gap that does not fit 16 bits is stored as 0xffff and the column is read from the matrix's own
// int32 column array instead -- numberings whose far neighbours are many and irregular (the reference's partition
// renumbering: every line of a part ends at a neighbour a million rows away; curve-ordered meshes) keep 10 bytes per nonzero
// for all but the escaping entries.  Same products in the same order as k_spmv / k_spmv16.
__global__ void __launch_bounds__(kBlock) k_cols16_fill_escape(SellDev A, const int64_t *slice_doff, int32_t *col0, uint32_t *dwords,
                                                                unsigned long long *escapes)
{
    const int64_t r = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    const int64_t s = r >> 6;
    if (s >= A.n_slices) return;
    const int lane = static_cast<int>(r & 63);
    const int64_t off = A.slice_off[s];
    const int width = static_cast<int>((A.slice_off[s + 1] - off) >> 6);
    const int len = r < A.n_rows ? A.rowlen[r] : 0;
    const int32_t *cp = A.cols + off + lane;
    int prev = len > 0 ? cp[0] : 0;
    col0[r] = prev;
    uint32_t *wp = dwords + slice_doff[s] + lane;
    unsigned esc = 0;
    for (int j = 0; 2 * j + 1 < width; ++j) {
        uint32_t w = 0;
        for (int h = 0; h < 2; ++h) {
            const int k = 2 * j + 1 + h;
            uint32_t gap = 0;                      // pads: gap 0 (the product is with a stored zero)
            if (k < len) {
                const int c = cp[64 * k];
                const int64_t g = static_cast<int64_t>(c) - prev;
                if (g < 0 || g >= 0xffff) { gap = 0xffffu; ++esc; }
                else gap = static_cast<uint32_t>(g);
                prev = c;
            }
            w |= gap << (16 * h);
        }
        wp[64LL * j] = w;
    }
    if (esc) atomicAdd(escapes, static_cast<unsigned long long>(esc));
}
template <int W>
__device__ __forceinline__ void spmv16e_trip(const double *__restrict__ vp, const uint32_t *__restrict__ wp, const int32_t *__restrict__ cp,
                                             const double *__restrict__ x, int &j, int &c, double &acc)
{
    uint32_t w[W];
    double v[2 * W], xv[2 * W];
#pragma unroll
    for (int t = 0; t < W; ++t) w[t] = __builtin_nontemporal_load(wp + 64 * (j + t));
#pragma unroll
    for (int t = 0; t < 2 * W; ++t) v[t] = __builtin_nontemporal_load(vp + 64 * (2 * j + 1 + t));
    // the escaped columns first, all of them in flight together (their addresses do not depend on the running column), then the
    // running sum: a load inside the sum would wait for memory once per escape
    int e[2 * W];
#pragma unroll
    for (int t = 0; t < 2 * W; ++t) {
        const uint32_t code = (t & 1) ? (w[t >> 1] >> 16) : (w[t >> 1] & 0xffffu);
        e[t] = 0;
        if (code == 0xffffu) e[t] = __builtin_nontemporal_load(cp + 64 * (2 * j + 1 + t));
    }
    int cc = c;
#pragma unroll
    for (int t = 0; t < 2 * W; ++t) {
        const uint32_t code = (t & 1) ? (w[t >> 1] >> 16) : (w[t >> 1] & 0xffffu);
        cc = code == 0xffffu ? e[t] : cc + static_cast<int>(code);
        xv[t] = x[cc];
    }
#pragma unroll
    for (int t = 0; t < 2 * W; ++t) acc = __builtin_fma(v[t], xv[t], acc);
    c = cc;
    j += W;
}
template <bool WITH_DOT>
__global__ void __launch_bounds__(kBlock) k_spmv16e(SellDev A, Sell16Dev C, const double *__restrict__ x, double *__restrict__ y,
                                                     int64_t n_dot, double *partial, const CgCtl *ctl, SliceSel sel)
{
    __shared__ double sm[4];
    if (WITH_DOT && ctl->flag != 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t s = pick_slice(sel, (static_cast<int64_t>(blockIdx.x) << 2) + wave, A.n_slices);
    double dot = 0.0;
    if (s < A.n_slices) {
        const int64_t off = A.slice_off[s];
        const int width = static_cast<int>((A.slice_off[s + 1] - off) >> 6);
        const double *__restrict__ vp = A.vals + off + lane;
        const int32_t *__restrict__ cp = A.cols + off + lane;
        const uint32_t *__restrict__ wp = C.dwords + C.slice_doff[s] + lane;
        int c = __builtin_nontemporal_load(C.col0 + (s << 6) + lane);
        double acc = 0.0;
        if (width > 0) acc = __builtin_nontemporal_load(vp) * x[c];
        const int nw = width / 2;
        int j = 0;
        while (2 * (j + 8) < width) spmv16e_trip<8>(vp, wp, cp, x, j, c, acc);
        if (2 * (j + 4) < width) spmv16e_trip<4>(vp, wp, cp, x, j, c, acc);
        if (2 * (j + 2) < width) spmv16e_trip<2>(vp, wp, cp, x, j, c, acc);
        for (; j < nw; ++j) {
            const uint32_t w0 = __builtin_nontemporal_load(wp + 64 * j);
            const uint32_t lo = w0 & 0xffffu, hi = w0 >> 16;
            const int c0 = lo == 0xffffu ? __builtin_nontemporal_load(cp + 64 * (2 * j + 1)) : c + static_cast<int>(lo);
            acc = __builtin_fma(__builtin_nontemporal_load(vp + 64 * (2 * j + 1)), x[c0], acc);
            int c1 = c0;
            if (2 * j + 2 < width) {
                c1 = hi == 0xffffu ? __builtin_nontemporal_load(cp + 64 * (2 * j + 2)) : c0 + static_cast<int>(hi);
                acc = __builtin_fma(__builtin_nontemporal_load(vp + 64 * (2 * j + 2)), x[c1], acc);
            }
            c = c1;
        }
        const int64_t row = (s << 6) + lane;
        if (row < A.n_rows) {
            y[row] = acc;
            if (WITH_DOT && row < n_dot) dot = x[row] * acc;
        }
    }
    if (WITH_DOT) {
        const double t = block_sum(dot, sm);
        if (threadIdx.x == 0) partial[blockIdx.x] = t;
    }
}

// ---------------------------------------------------------------------------
// Row-grouped SpMV (problems with several dofs per node).  Consecutive rows with IDENTICAL column
// sets -- the dof rows of one node -- form a group of up to kGroupRows rows served by ONE lane: the
// column stream (first column + 16-bit gaps, as above) and the x gather are shared by the rows of
// the group, so an entry costs 8 B + 2/3 B instead of 8 + 2 B and a third of the gather
// instructions.  Groups are found from the pattern alone (k_group_breaks), no mesh needed; the
// values are a regrouped copy of the matrix made at the start of a solve (k_group_vals).  Every
// row still accumulates its products in ascending column order: y is bit-identical to k_spmv16.
// ---------------------------------------------------------------------------
constexpr int kGroupRows = 3;

struct SellGDev {
    int64_t n_groups, n_gslices;
    const int32_t *group_row0;   // [n_groups+1] first row of every group (sentinel: n_rows)
    const int64_t *gslice_off;   // [n_gslices+1] entries per plane, 64 * width of the slice of 64 groups
    const double *vals;          // entry k of row p of lane l: vals[kGroupRows*off + (kGroupRows*k + p)*64 + l]
    const int32_t *col0;         // [64 * n_gslices]
    const uint32_t *dwords;      // packed gaps of the group's column list
    const int64_t *gslice_doff;  // [n_gslices+1] in words
    const uint32_t *gap_table;   // dictionary form, else unused
};

// run_start[r] = r if row r cannot share a group with row r-1 (different length or columns), else 0;
// an inclusive max-scan then gives every row the first row of its run
__global__ void __launch_bounds__(kBlock) k_group_breaks(SellDev A, int32_t *run_start)
{
    const int64_t r = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (r >= A.n_rows) return;
    bool brk = r == 0;
    if (!brk) {
        const int len = A.rowlen[r];
        brk = len != A.rowlen[r - 1] || len == 0;
        const int64_t b1 = A.slice_off[r >> 6] + (r & 63), b0 = A.slice_off[(r - 1) >> 6] + ((r - 1) & 63);
        for (int k = 0; k < len && !brk; ++k) brk = A.cols[b1 + 64LL * k] != A.cols[b0 + 64LL * k];
    }
    run_start[r] = brk ? static_cast<int32_t>(r) : 0;
}

__global__ void __launch_bounds__(kBlock) k_group_flags(const int32_t *run_start, int64_t n, char *flag)
{
    const int64_t r = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (r < n) flag[r] = ((r - run_start[r]) % kGroupRows) == 0;
}

// width of every slice of 64 groups (the longest row among them), as 64*width entries per plane
__global__ void __launch_bounds__(kBlock) k_gslice_sizes(const int32_t *group_row0, const int32_t *rowlen, int64_t n_groups,
                                                          int64_t n_gslices, int64_t *entries)
{
    const int64_t g = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    int c = g < n_groups ? rowlen[group_row0[g]] : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c = max(c, __shfl_xor(c, o, 64));
    const int64_t gs = g >> 6;
    if ((threadIdx.x & 63) == 0 && gs < n_gslices) entries[gs] = 64LL * c;
    if (g == 0) entries[n_gslices] = 0;
}

// first column + packed 16-bit gaps of every group's column list (that of its first row)
__global__ void __launch_bounds__(kBlock) k_group_cols_fill(SellDev A, const int32_t *group_row0, int64_t n_groups,
                                                             int64_t n_gslices, const int64_t *gslice_off,
                                                             const int64_t *gslice_doff, int32_t *col0, uint32_t *dwords,
                                                             const uint32_t *gap_table /* dictionary form, else null */, int *miss)
{
    const int64_t g = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    const int64_t gs = g >> 6;
    if (gs >= n_gslices) return;
    const int lane = static_cast<int>(g & 63);
    const int width = static_cast<int>((gslice_off[gs + 1] - gslice_off[gs]) >> 6);
    int len = 0;
    const int32_t *cp = A.cols;
    if (g < n_groups) {
        const int64_t r = group_row0[g];
        len = A.rowlen[r];
        cp = A.cols + A.slice_off[r >> 6] + (r & 63);
    }
    int prev = len > 0 ? cp[0] : 0;
    col0[g] = prev;
    uint32_t *wp = dwords + gslice_doff[gs] + lane;
    for (int j = 0; 2 * j + 1 < width; ++j) {
        uint32_t w = 0;
        for (int h = 0; h < 2; ++h) {
            const int k = 2 * j + 1 + h;
            if (k < len) {
                const int c = cp[64LL * k];
                const uint32_t gap = static_cast<uint32_t>(c - prev);           // fits / is in the table: checked on the row form
                w |= ((gap_table ? gap_table_code(gap_table, gap, miss) : gap) & 0xffffu) << (16 * h);
                prev = c;
            }
        }
        wp[64LL * j] = w;
    }
}

// matrix values, row form -> grouped form (zero padded); once per solve
__global__ void __launch_bounds__(kBlock) k_group_vals(SellDev A, SellGDev G, double *out)
{
    const int64_t g = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    const int64_t gs = g >> 6;
    if (gs >= G.n_gslices) return;
    const int lane = static_cast<int>(g & 63);
    const int64_t off = G.gslice_off[gs];
    const int width = static_cast<int>((G.gslice_off[gs + 1] - off) >> 6);
    double *op = out + kGroupRows * off + lane;
    int64_t r0 = 0;
    int sz = 0, len = 0;
    if (g < G.n_groups) { r0 = G.group_row0[g]; sz = G.group_row0[g + 1] - static_cast<int>(r0); len = A.rowlen[r0]; }
#pragma unroll
    for (int p = 0; p < kGroupRows; ++p) {
        const int64_t r = r0 + p;
        const double *vp = A.vals + A.slice_off[(p < sz ? r : r0) >> 6] + ((p < sz ? r : r0) & 63);
        for (int k = 0; k < width; ++k)
            op[(static_cast<int64_t>(kGroupRows) * k + p) * 64] = (p < sz && k < len) ? vp[64LL * k] : 0.0;
    }
}

// W words = 2W consecutive entries of kGroupRows rows (caller guarantees 2*(j+W) < width)
template <int W, bool DICT>
__device__ __forceinline__ void spmvg_trip(const double *__restrict__ vp, const uint32_t *__restrict__ wp,
                                           const double *__restrict__ x, int &j, int &c, double (&acc)[kGroupRows],
                                           const uint32_t *tbl)
{
    uint32_t w[W];
    double v[2 * W][kGroupRows], xv[2 * W];
#pragma unroll
    for (int t = 0; t < W; ++t) w[t] = __builtin_nontemporal_load(wp + 64 * (j + t));
#pragma unroll
    for (int t = 0; t < 2 * W; ++t)
#pragma unroll
        for (int p = 0; p < kGroupRows; ++p)
            v[t][p] = __builtin_nontemporal_load(vp + 64 * (kGroupRows * (2 * j + 1 + t) + p));
#pragma unroll
    for (int t = 0; t < W; ++t) {
        const int c0 = c + gap16<DICT>(w[t] & 0xffffu, tbl);
        const int c1 = c0 + gap16<DICT>(w[t] >> 16, tbl);
        xv[2 * t] = x[c0];
        xv[2 * t + 1] = x[c1];
        c = c1;
    }
    __builtin_amdgcn_sched_barrier(0);     // all loads of the trip are issued before the first use
#pragma unroll
    for (int t = 0; t < 2 * W; ++t)
#pragma unroll
        for (int p = 0; p < kGroupRows; ++p) acc[p] = __builtin_fma(v[t][p], xv[t], acc[p]);
    j += W;
}

template <bool WITH_DOT, bool DICT = false>
__global__ void __launch_bounds__(kBlock) k_spmvg(SellGDev G, int64_t n_rows, const double *__restrict__ x,
                                                   double *__restrict__ y, int64_t n_dot, double *partial, const CgCtl *ctl,
                                                   SliceSel sel)
{
    __shared__ double sm[4];
    __shared__ uint32_t tbl[DICT ? kGapTable : 1];
    if (WITH_DOT && ctl->flag != 0) return;
    if (DICT) {
        tbl[DICT ? threadIdx.x : 0] = G.gap_table[threadIdx.x];
        __syncthreads();
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t gs = pick_slice(sel, (static_cast<int64_t>(blockIdx.x) << 2) + wave, G.n_gslices);
    double dot = 0.0;
    if (gs < G.n_gslices) {
        const int64_t off = G.gslice_off[gs];
        const int width = static_cast<int>((G.gslice_off[gs + 1] - off) >> 6);
        const double *__restrict__ vp = G.vals + kGroupRows * off + lane;
        const uint32_t *__restrict__ wp = G.dwords + G.gslice_doff[gs] + lane;
        int c = __builtin_nontemporal_load(G.col0 + (gs << 6) + lane);
        double acc[kGroupRows];
#pragma unroll
        for (int p = 0; p < kGroupRows; ++p) acc[p] = 0.0;
        if (width > 0) {
            const double x0 = x[c];
#pragma unroll
            for (int p = 0; p < kGroupRows; ++p) acc[p] = __builtin_nontemporal_load(vp + 64 * p) * x0;
        }
        const int nw = width / 2;
        int j = 0;
        while (2 * (j + 4) < width) spmvg_trip<4, DICT>(vp, wp, x, j, c, acc, tbl);
        if (2 * (j + 2) < width) spmvg_trip<2, DICT>(vp, wp, x, j, c, acc, tbl);
        for (; j < nw; ++j) {
            const uint32_t w0 = __builtin_nontemporal_load(wp + 64 * j);
            const int c0 = c + gap16<DICT>(w0 & 0xffffu, tbl);
            const int c1 = c0 + gap16<DICT>(w0 >> 16, tbl);
            const double x0 = x[c0];
#pragma unroll
            for (int p = 0; p < kGroupRows; ++p)
                acc[p] = __builtin_fma(__builtin_nontemporal_load(vp + 64 * (kGroupRows * (2 * j + 1) + p)), x0, acc[p]);
            if (2 * j + 2 < width) {
                const double x1 = x[c1];
#pragma unroll
                for (int p = 0; p < kGroupRows; ++p)
                    acc[p] = __builtin_fma(__builtin_nontemporal_load(vp + 64 * (kGroupRows * (2 * j + 2) + p)), x1, acc[p]);
            }
            c = c1;
        }
        const int64_t g = (gs << 6) + lane;
        if (g < G.n_groups) {
            const int64_t r0 = G.group_row0[g];
            const int sz = G.group_row0[g + 1] - static_cast<int>(r0);
#pragma unroll
            for (int p = 0; p < kGroupRows; ++p)
                if (p < sz) {
                    y[r0 + p] = acc[p];
                    if (WITH_DOT && r0 + p < n_dot) dot = __builtin_fma(x[r0 + p], acc[p], dot);
                }
        }
    }
    if (WITH_DOT) {
        const double t = block_sum(dot, sm);
        if (threadIdx.x == 0) partial[blockIdx.x] = t;
    }
}

// One block of 1024 threads: out[j] = sum of part_j[0..n) for up to two partial arrays, in a
// fixed association order (bitwise reproducible run to run).
__global__ void __launch_bounds__(1024) k_reduce_partials(const double *part0, const double *part1, int n, double *out,
                                                           const CgCtl *ctl)
{
    __shared__ double sm[16];
    if (ctl && ctl->flag != 0) return;
    for (int j = 0; j < 2; ++j) {
        const double *part = j ? part1 : part0;
        if (!part) continue;
        double a = 0.0;
        for (int i = threadIdx.x; i < n; i += 1024) a += part[i];
        a = wave_sum(a);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = a;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int w = 0; w < 16; ++w) t += sm[w];
            out[j] = t;
        }
    }
}

// First stage of a two-stage fold of MANY partials (one per SpMV block): block b of kFoldBlocks
// sums the b-th contiguous chunk into out[b]; the consumer kernel re-sums those kFoldBlocks values
// (sum_partials).  Fixed chunking and order -> bitwise reproducible; ~2x quicker than one block.
constexpr int kFoldBlocks = 32;
__global__ void __launch_bounds__(kBlock) k_fold_partials(const double *__restrict__ part, int n, double *out,
                                                           const CgCtl *ctl)
{
    __shared__ double sm[4];
    if (ctl && ctl->flag != 0) return;
    const int chunk = (n + kFoldBlocks - 1) / kFoldBlocks;
    const int lo = blockIdx.x * chunk, hi = min(n, lo + chunk);
    double a = 0.0;
    for (int i = lo + threadIdx.x; i < hi; i += kBlock) a += part[i];
    const double t = block_sum(a, sm);
    if (threadIdx.x == 0) out[blockIdx.x] = t;
}

// ---------------------------------------------------------------------------
// Jacobi-PCG vector kernels (PETSc KSPCG semantics)
// ---------------------------------------------------------------------------
// x = 0, r = b, p = z = r*dinv; partials of (r,z) and (z,z) over the owned rows.
__global__ void __launch_bounds__(kBlock) k_cg_init(int64_t n, int64_t n_owned, const double *__restrict__ b,
                                                     const double *__restrict__ dinv, double *__restrict__ x,
                                                     double *__restrict__ r, double *__restrict__ p,
                                                     double *part_rz, double *part_zz)
{
    __shared__ double sm[4];
    double rz = 0.0, zz = 0.0;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n;
         i += static_cast<int64_t>(gridDim.x) * kBlock) {
        const double ri = b[i], zi = ri * dinv[i];
        x[i] = 0.0;
        r[i] = ri;
        p[i] = zi;
        if (i < n_owned) { rz = __builtin_fma(ri, zi, rz); zz = __builtin_fma(zi, zi, zz); }
    }
    const double a = block_sum(rz, sm), c = block_sum(zz, sm);
    if (threadIdx.x == 0) { part_rz[blockIdx.x] = a; part_zz[blockIdx.x] = c; }
}

// one block: finish the initial reductions, set tolerances (KSPConvergedDefault)
__global__ void __launch_bounds__(kBlock) k_cg_start(CgCtl *ctl, const double *part_rz, const double *part_zz, int nparts,
                                                      const double *reduced /*[rz,zz] or null*/, double rtol, double abstol,
                                                      double dtol, double *hist)
{
    __shared__ double sm[4];
    double rz, zz;
    if (reduced) { rz = reduced[0]; zz = reduced[1]; }
    else { rz = sum_partials(part_rz, nparts, sm); zz = sum_partials(part_zz, nparts, sm); }
    if (threadIdx.x == 0) {
        const double rn0 = sqrt(zz);
        ctl->beta[0] = rz;
        ctl->beta[1] = 0.0;
        ctl->rn0 = rn0;
        ctl->rn = rn0;
        ctl->ttol = fmax(rtol * rn0, abstol);
        ctl->dtol = dtol;
        ctl->its = 0;
        ctl->flag = (rn0 <= abstol) ? 3 : ((rz < 0.0) ? -8 : 0);
        hist[0] = rn0;
    }
}

// The inverse diagonal as the Jacobi loop's two vector kernels read it: doubles, or -- where the diagonal repeats its values like
// the matrix does (structured meshes; pfem_valdict.hpp) -- 16-bit codes into a dictionary of the distinct ones: 2 B a row
// instead of 8 in each of the two kernels of an iteration, 94 MB of 630 at config 3.  The verdict of the encoding stays on the
// device (state[1]: the collection overflowed, state[2]: a value missing): the kernels branch on it, the host never waits.
struct DinvView {
    const double *dinv;
    const uint16_t *codes;      // null: doubles only
    const double *dict;
    const int *state;
    __device__ __forceinline__ bool coded() const { return codes != nullptr && state[1] == 0 && state[2] == 0; }
    __device__ __forceinline__ double at(int64_t i, bool use_codes) const { return use_codes ? dict[codes[i]] : dinv[i]; }
};

// alpha = beta/(p,w); r -= alpha w; partials of (r,z), (z,z), z = r*dinv.  The solution update x += alpha p
// is done by k_cg_direction, which holds p anyway (alpha travels in ctl): same-box A/B -1.3 % per iteration.
// `it_arg` < 0 (launches replayed from a hipGraph): the iteration index is taken from the control block.
__global__ void __launch_bounds__(kBlock) k_cg_update(CgCtl *ctl, int it_arg, int64_t n, int64_t n_owned,
                                                       const double *part_pw, int nparts, const double *reduced_pw,
                                                       const double *__restrict__ p, const double *__restrict__ w,
                                                       DinvView dinv, double *__restrict__ x,
                                                       double *__restrict__ r, double *part_rz, double *part_zz)
{
    __shared__ double sm[4];
    if (ctl->flag != 0) {                   // finished in an earlier iteration (this kernel never writes flag)
        if (blockIdx.x == 0 && threadIdx.x == 0) ctl->its_dir = -2;
        return;
    }
    const int it = it_arg >= 0 ? it_arg : ctl->its;
    if (blockIdx.x == 0 && threadIdx.x == 0) ctl->its_dir = it;      // nobody in this launch reads it
    const double pw = reduced_pw ? *reduced_pw : sum_partials(part_pw, nparts, sm);
    if (!(pw > 0.0)) {                      // KSP_DIVERGED_INDEFINITE_MAT
        if (blockIdx.x == 0 && threadIdx.x == 0) { ctl->its = it; }
        part_rz[blockIdx.x] = 0.0; part_zz[blockIdx.x] = -1.0;   // signals breakdown to k_cg_direction
        return;
    }
    const double alpha = ctl->beta[it & 1] / pw;
    if (blockIdx.x == 0 && threadIdx.x == 0) ctl->alpha = alpha;
    double rz = 0.0, zz = 0.0;
    const bool dc = dinv.coded();
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n;
         i += static_cast<int64_t>(gridDim.x) * kBlock) {
        const double ri = __builtin_fma(-alpha, w[i], r[i]);
        r[i] = ri;
        const double zi = ri * dinv.at(i, dc);
        if (i < n_owned) { rz = __builtin_fma(ri, zi, rz); zz = __builtin_fma(zi, zi, zz); }
    }
    const double a = block_sum(rz, sm), c = block_sum(zz, sm);
    if (threadIdx.x == 0) { part_rz[blockIdx.x] = a; part_zz[blockIdx.x] = c; }
}

// finish (r,z), ||z||; x += alpha p (the step of THIS iteration, also when it turns out to be the last one: as in
// KSPCG the test follows the update; on a breakdown x is not advanced); convergence test; p = z + (beta_new/beta) p
__global__ void __launch_bounds__(kBlock) k_cg_direction(CgCtl *ctl, int it_arg, int64_t n, const double *part_rz,
                                                          const double *part_zz, int nparts, const double *reduced,
                                                          const double *__restrict__ r, DinvView dinv,
                                                          double *__restrict__ p, double *hist, int hist_cap, int maxits, double *__restrict__ x)
{
    __shared__ double sm[4];
    const int it = it_arg >= 0 ? it_arg : ctl->its_dir;              // ctl->its is written by this launch: not read here
    if (ctl_finished_before(ctl, it)) return;
    double rz, zz;
    if (reduced) { rz = reduced[0]; zz = reduced[1]; }
    else { rz = sum_partials(part_rz, nparts, sm); zz = sum_partials(part_zz, nparts, sm); }
    const bool lead = blockIdx.x == 0 && threadIdx.x == 0;
    if (zz < 0.0) {                          // breakdown flagged by k_cg_update: KSP_DIVERGED_INDEFINITE_MAT
        if (lead) ctl_publish(ctl, -10, it + 1);
        return;
    }
    const double rn = sqrt(zz);
    const double beta_old = ctl->beta[it & 1];
    int flag = 0;
    if (rn <= ctl->ttol) flag = 2;           // KSP_CONVERGED_RTOL (or ATOL, resolved on the host)
    else if (rn >= ctl->dtol * ctl->rn0) flag = -4;
    else if (rz < 0.0) flag = -8;            // KSP_DIVERGED_INDEFINITE_PC
    else if (it + 1 >= maxits) flag = -3;
    if (lead) {
        // other blocks of THIS launch read only beta[it&1]/ttol/dtol/rn0/alpha and test the verdict word against `it`
        ctl->beta[(it + 1) & 1] = rz;
        ctl->rn = rn;
        if (it + 1 < hist_cap) hist[it + 1] = rn;
        ctl_publish(ctl, flag, it + 1);
    }
    const double alpha = ctl->alpha;
    if (flag != 0) {
        for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n;
             i += static_cast<int64_t>(gridDim.x) * kBlock)
            __builtin_nontemporal_store(__builtin_fma(alpha, p[i], __builtin_nontemporal_load(x + i)), x + i);
        return;
    }
    const double bb = rz / beta_old;
    const bool dc = dinv.coded();
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n;
         i += static_cast<int64_t>(gridDim.x) * kBlock) {
        const double pi = p[i];
        __builtin_nontemporal_store(__builtin_fma(alpha, pi, __builtin_nontemporal_load(x + i)), x + i);
        p[i] = __builtin_fma(bb, pi, r[i] * dinv.at(i, dc));
    }
}

// ---------------------------------------------------------------------------
// Single-reduction form of the same iteration (Chronopoulos & Gear; PETSc: KSPCGUseSingleReduction,
// -ksp_cg_single_reduction): s = A z is formed instead of w = A p, so that (z,r), (z,s) and (z,z) are all known at ONE
// point of the iteration -- one all-reduce per iteration on several ranks instead of two -- and
//     b = beta/beta_old,  (p,Ap) = (z,s) - beta^2 (p,Ap)_old / beta_old^2,  a = beta/(p,Ap),
//     p = z + b p,  w = s + b w  (= A p by recurrence),  x += a p,  r -= a w,  z = M^-1 r.
// Step `it` first judges the iterate that step it-1 produced (its norm is only known now, one SpMV later than in the
// two-reduction loop: the price PETSc's variant pays too), then advances.  Every block derives the same verdict from the
// same bits, so a block that starts late and finds the lead block's verdict already published does what it would have
// decided itself.  The (r,z)/(z,z) partials are read and written in the same launch: two sets, by iteration parity.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) k_reduce_partials3(const double *part0, int n0, const double *part1, const double *part2,
                                                            int n12, double *out, const CgCtl *ctl)
{
    __shared__ double sm[16];
    if (ctl && ctl->flag != 0) return;
    for (int j = 0; j < 3; ++j) {
        const double *part = j == 0 ? part0 : (j == 1 ? part1 : part2);
        const int n = j == 0 ? n0 : n12;
        double a = 0.0;
        for (int i = threadIdx.x; i < n; i += 1024) a += part[i];
        a = wave_sum(a);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = a;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int w = 0; w < 16; ++w) t += sm[w];
            out[j] = t;
        }
    }
}

__global__ void __launch_bounds__(kBlock) k_cg1_step(CgCtl *ctl, int it, int64_t n, int64_t n_owned, const double *part_zs, int n_zs,
                                                      const double *part_rz, const double *part_zz, int nparts,
                                                      const double *reduced /* [(z,s), (r,z), (z,z)] summed over the ranks, or null */,
                                                      const double *__restrict__ dinv, double *__restrict__ z,
                                                      const double *__restrict__ sv, double *__restrict__ p, double *__restrict__ w,
                                                      double *__restrict__ x, double *__restrict__ r, double *out_rz, double *out_zz,
                                                      double rtol, double abstol, double dtol, double *hist, int hist_cap, int maxits)
{
    __shared__ double sm[4];
    if (ctl->flag != 0) return;
    double zs, rz, zz;
    if (reduced) { zs = reduced[0]; rz = reduced[1]; zz = reduced[2]; }
    else { zs = sum_partials(part_zs, n_zs, sm); rz = sum_partials(part_rz, nparts, sm); zz = sum_partials(part_zz, nparts, sm); }
    const bool lead = blockIdx.x == 0 && threadIdx.x == 0;
    const double rn = sqrt(zz);
    int flag = 0;
    if (it == 0 && rn <= abstol) flag = 3;
    else if (it > 0 && rn <= ctl->ttol) flag = 2;                    // KSP_CONVERGED_RTOL (or ATOL, resolved on the host)
    else if (it > 0 && rn >= ctl->dtol * ctl->rn0) flag = -4;
    else if (rz < 0.0) flag = -8;                                    // KSP_DIVERGED_INDEFINITE_PC
    else if (it >= maxits) flag = -3;
    if (lead) {
        if (it == 0) { ctl->rn0 = rn; ctl->ttol = fmax(rtol * rn, abstol); ctl->dtol = dtol; }
        ctl->rn = rn;
        if (it < hist_cap) hist[it] = rn;
    }
    if (flag != 0) {
        if (lead) ctl_publish(ctl, flag, it);
        return;
    }
    double b = 0.0, dpi = zs;
    if (it > 0) {
        const double beta_old = ctl->beta[(it + 1) & 1], dpi_old = ctl->dpi[(it + 1) & 1];
        b = rz / beta_old;
        dpi = zs - rz * rz * dpi_old / (beta_old * beta_old);
    }
    if (!(dpi > 0.0)) {                          // KSP_DIVERGED_INDEFINITE_MAT: x is not advanced
        if (lead) ctl_publish(ctl, -10, it + 1);
        return;
    }
    const double a = rz / dpi;
    if (lead) {
        ctl->beta[it & 1] = rz;
        ctl->dpi[it & 1] = dpi;
        ctl->alpha = a;
        ctl_publish(ctl, 0, it);
    }
    double nrz = 0.0, nzz = 0.0;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n;
         i += static_cast<int64_t>(gridDim.x) * kBlock) {
        const double zi = z[i], si = sv[i];
        const double pi = it ? __builtin_fma(b, p[i], zi) : zi;
        const double wi = it ? __builtin_fma(b, w[i], si) : si;
        p[i] = pi;
        w[i] = wi;
        __builtin_nontemporal_store(__builtin_fma(a, pi, __builtin_nontemporal_load(x + i)), x + i);
        const double ri = __builtin_fma(-a, wi, r[i]);
        r[i] = ri;
        const double zn = ri * dinv[i];
        z[i] = zn;
        if (i < n_owned) { nrz = __builtin_fma(ri, zn, nrz); nzz = __builtin_fma(zn, zn, nzz); }
    }
    const double t0 = block_sum(nrz, sm), t1 = block_sum(nzz, sm);
    if (threadIdx.x == 0) { out_rz[blockIdx.x] = t0; out_zz[blockIdx.x] = t1; }
}

// ---------------------------------------------------------------------------
// Relative row groups (scalar problems on regularly numbered meshes).  kRelRows CONSECUTIVE rows
// r0..r0+3 are served by one lane with ONE column stream: the union of their column sets taken
// RELATIVE to the row (col - row), i.e. entry k of row r0+p multiplies x[c_k + p].  Where adjacent
// rows see the same stencil (x-lines of a structured mesh) the union is as long as one row, so an
// entry costs 8 + 2/4 B instead of 8 + 2 B, and the four x values of an entry are one contiguous
// 32-B read.  A row that lacks an offset of the union carries an explicit zero there (the product
// is +-0 and leaves the sum unchanged); x is read up to kRelRows-1 places outside [0,n) for such
// zeros, hence the guard band of the SpMV input vector.  Built from the pattern alone and kept only
// when it is smaller than the row form (unstructured numbering: unions grow, the row form stays).
// ---------------------------------------------------------------------------
constexpr int kRelRows = 4;
constexpr int kVecGuard = 8;       // zeros kept before and after the SpMV input vector

struct SellRDev {
    int64_t n_groups, n_gslices;
    const int64_t *gslice_off;   // [n_gslices+1] entries per plane, 64 * width
    const double *vals;          // entry k of row p of lane l: vals[kRelRows*off + (kRelRows*k + p)*64 + l]
    const int32_t *col0;         // [64 * n_gslices] c_0 = r0 + (smallest relative offset); may be < 0
    const uint32_t *dwords;      // packed 16-bit gaps between the ascending relative offsets
    const int64_t *gslice_doff;
    const uint32_t *gap_table;   // dictionary form: a 16-bit code >= 0x8000 stands for gap_table[code & 0x7fff]
};

// Walks the union of the relative column lists of rows r0 .. r0+nr-1 in ascending order and calls
// f(k, offset).  Returns the union size.
template <class F>
__device__ __forceinline__ int rel_union_walk(const SellDev &A, int64_t r0, int nr, F &&f)
{
    int head[kRelRows], len[kRelRows];
    const int32_t *cp[kRelRows];
#pragma unroll
    for (int p = 0; p < kRelRows; ++p) {
        head[p] = 0;
        len[p] = p < nr ? A.rowlen[r0 + p] : 0;
        cp[p] = A.cols + A.slice_off[(r0 + (p < nr ? p : 0)) >> 6] + ((r0 + (p < nr ? p : 0)) & 63);
    }
    int k = 0;
    for (;;) {
        int64_t best = INT64_MAX;
#pragma unroll
        for (int p = 0; p < kRelRows; ++p)
            if (head[p] < len[p]) {
                const int64_t o = static_cast<int64_t>(cp[p][64LL * head[p]]) - (r0 + p);
                best = o < best ? o : best;
            }
        if (best == INT64_MAX) break;
        f(k, best);
        ++k;
#pragma unroll
        for (int p = 0; p < kRelRows; ++p)
            if (head[p] < len[p] && static_cast<int64_t>(cp[p][64LL * head[p]]) - (r0 + p) == best) ++head[p];
    }
    return k;
}

// union size of every group (as 64*width per slice of 64 groups) and the 16-bit test of its gaps
__global__ void __launch_bounds__(kBlock) k_rel_sizes(SellDev A, int64_t n_groups, int64_t n_gslices, int64_t *entries,
                                                       int *overflow)
{
    const int64_t g = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    int u = 0;
    if (g < n_groups) {
        const int64_t r0 = g * kRelRows;
        const int nr = static_cast<int>(min(static_cast<int64_t>(kRelRows), A.n_rows - r0));
        int64_t prev = 0;
        bool bad = false;
        bool wide = false;
        u = rel_union_walk(A, r0, nr, [&](int k, int64_t o) {
            if (k > 0 && o - prev > 65535) wide = true;                  // needs the 32-bit gap stream
            if (k == 0 && (r0 + o < INT32_MIN || r0 + o > INT32_MAX)) bad = true;
            prev = o;
        });
        if (wide) atomicOr(overflow, 1);
        if (bad) atomicOr(overflow, 2);
    }
    int c = u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c = max(c, __shfl_xor(c, o, 64));
    const int64_t gs = g >> 6;
    if ((threadIdx.x & 63) == 0 && gs < n_gslices) entries[gs] = 64LL * c;
    if (g == 0) entries[n_gslices] = 0;
}

// second walk, only for patterns with a gap beyond 65535: the distinct gaps of 32768 and more
__global__ void __launch_bounds__(kBlock) k_rel_gap_table(SellDev A, int64_t n_groups, uint32_t *tbl, int *overflow)
{
    const int64_t g = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (g >= n_groups) return;
    const int64_t r0 = g * kRelRows;
    const int nr = static_cast<int>(min(static_cast<int64_t>(kRelRows), A.n_rows - r0));
    int64_t prev = 0;
    rel_union_walk(A, r0, nr, [&](int k, int64_t o) {
        if (k > 0 && o - prev >= 0x8000) {
            if (o - prev > 0xffffffffLL) atomicOr(overflow, 4);
            else gap_table_insert(tbl, static_cast<uint32_t>(o - prev), overflow);
        }
        prev = o;
    });
}

template <int MODE>
__global__ void __launch_bounds__(kBlock) k_rel_cols_fill(SellDev A, int64_t n_groups, int64_t n_gslices,
                                                           const int64_t *gslice_off, const int64_t *gslice_doff, int32_t *col0,
                                                           uint32_t *dwords, const uint32_t *gap_table, int *miss)
{
    constexpr bool GAP32 = MODE == kGap32;
    const int64_t g = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    const int64_t gs = g >> 6;
    if (gs >= n_gslices) return;
    const int lane = static_cast<int>(g & 63);
    const int width = static_cast<int>((gslice_off[gs + 1] - gslice_off[gs]) >> 6);
    uint32_t *wp = dwords + gslice_doff[gs] + lane;
    if (GAP32) { for (int j = 0; j + 1 < width; ++j) wp[64LL * j] = 0u; }
    else { for (int j = 0; 2 * j + 1 < width; ++j) wp[64LL * j] = 0u; }        // pads: gap 0
    const int64_t r0 = g * kRelRows;
    col0[g] = static_cast<int32_t>(g < n_groups ? r0 : 0);          // empty group: any valid address
    if (g >= n_groups) return;
    const int nr = static_cast<int>(min(static_cast<int64_t>(kRelRows), A.n_rows - r0));
    int64_t prev = 0;
    rel_union_walk(A, r0, nr, [&](int k, int64_t o) {
        if (k == 0) col0[g] = static_cast<int32_t>(r0 + o);
        else if (GAP32) wp[64LL * (k - 1)] = static_cast<uint32_t>(o - prev);
        else {
            const int j = (k - 1) >> 1, h = (k - 1) & 1;
            const uint32_t code = MODE == kGapDict16 ? gap_table_code(gap_table, static_cast<uint32_t>(o - prev), miss)
                                                     : static_cast<uint32_t>(o - prev);
            wp[64LL * j] |= (code & 0xffffu) << (16 * h);
        }
        prev = o;
    });
}

// matrix values, row form -> relative-group form (explicit zeros where a row lacks an offset); once per solve.
// One 256-thread block per slice of 64 groups; wave p converts row r0+p of every group, so its stores are 512-B
// runs and the four waves of the block share the row-form lines they read.
template <int MODE>
__global__ void __launch_bounds__(kBlock) k_rel_vals(SellDev A, SellRDev G, double *out)
{
    constexpr bool GAP32 = MODE == kGap32;
    static_assert(kBlock == 64 * kRelRows, "one wave per row of the group");
    const int64_t gs = blockIdx.x;
    if (gs >= G.n_gslices) return;
    const int lane = threadIdx.x & 63, p = threadIdx.x >> 6;
    const int64_t g = (gs << 6) + lane;
    const int64_t off = G.gslice_off[gs];
    const int width = static_cast<int>((G.gslice_off[gs + 1] - off) >> 6);
    double *op = out + kRelRows * off + lane + 64 * p;
    const uint32_t *wp = G.dwords + G.gslice_doff[gs] + lane;
    const int64_t r = g * kRelRows + p;
    const bool live = g < G.n_groups && r < A.n_rows;
    const int len = live ? A.rowlen[r] : 0;
    const int64_t base = live ? A.slice_off[r >> 6] + (r & 63) : 0;
    int64_t c = G.col0[g] + p;          // column of entry k in THIS row
    int j = 0;
    for (int k = 0; k < width; ++k) {
        if (k > 0) {
            if (GAP32) c += wp[64LL * (k - 1)];
            else {
                const uint32_t w = wp[64LL * ((k - 1) >> 1)];
                const uint32_t code = ((k - 1) & 1) ? (w >> 16) : (w & 0xffffu);
                c += (MODE == kGapDict16 && (code & 0x8000u)) ? G.gap_table[code & (kGapTable - 1)] : code;
            }
        }
        double v = 0.0;
        while (j < len && A.cols[base + 64LL * j] < c) ++j;
        if (j < len && A.cols[base + 64LL * j] == c) { v = A.vals[base + 64LL * j]; ++j; }   // ++j: pads repeat c
        op[static_cast<int64_t>(kRelRows) * k * 64] = v;
    }
}

// union entry (place in the relative-group form) of every stored entry of the row form: relk[slot] = k, for an assembly
// that writes both forms (same walk as k_rel_vals); *wide when a union has more than 255 offsets
template <int MODE>
__global__ void __launch_bounds__(kBlock) k_rel_slot_map(SellDev A, SellRDev G, uint8_t *__restrict__ relk, int *__restrict__ wide)
{
    constexpr bool GAP32 = MODE == kGap32;
    static_assert(kRelRows == 4, "k_gather_poisson_tet4 writes the relative-group copy with 4 rows per group");
    const int64_t gs = blockIdx.x;
    if (gs >= G.n_gslices) return;
    const int lane = threadIdx.x & 63, p = threadIdx.x >> 6;
    const int64_t g = (gs << 6) + lane;
    const int64_t off = G.gslice_off[gs];
    const int width = static_cast<int>((G.gslice_off[gs + 1] - off) >> 6);
    if (width > 255) { *wide = 1; return; }
    const uint32_t *wp = G.dwords + G.gslice_doff[gs] + lane;
    const int64_t r = g * kRelRows + p;
    const bool live = g < G.n_groups && r < A.n_rows;
    const int len = live ? A.rowlen[r] : 0;
    const int64_t base = live ? A.slice_off[r >> 6] + (r & 63) : 0;
    int64_t c = G.col0[g] + p;
    int j = 0;
    for (int k = 0; k < width; ++k) {
        if (k > 0) {
            if (GAP32) c += wp[64LL * (k - 1)];
            else {
                const uint32_t w = wp[64LL * ((k - 1) >> 1)];
                const uint32_t code = ((k - 1) & 1) ? (w >> 16) : (w & 0xffffu);
                c += (MODE == kGapDict16 && (code & 0x8000u)) ? G.gap_table[code & (kGapTable - 1)] : code;
            }
        }
        while (j < len && A.cols[base + 64LL * j] < c) ++j;
        if (j < len && A.cols[base + 64LL * j] == c) { relk[base + 64LL * j] = static_cast<uint8_t>(k); ++j; }
    }
}

typedef double pfem_double2u __attribute__((ext_vector_type(2), aligned(8)));

// x[c .. c+3]: two 16-B loads (8-B aligned)
__device__ __forceinline__ void load_x4(const double *__restrict__ x, int c, double (&xv)[kRelRows])
{
    const pfem_double2u a = *reinterpret_cast<const pfem_double2u *>(x + c);
    const pfem_double2u b = *reinterpret_cast<const pfem_double2u *>(x + c + 2);
    xv[0] = a.x; xv[1] = a.y; xv[2] = b.x; xv[3] = b.y;
}

template <bool WITH_DOT, bool DICT = false>
__global__ void __launch_bounds__(kBlock) k_spmvr(SellRDev G, int64_t n_rows, const double *__restrict__ x,
                                                   double *__restrict__ y, int64_t n_dot, double *partial, const CgCtl *ctl,
                                                   SliceSel sel)
{
    __shared__ double sm[4];
    __shared__ uint32_t gap_tbl[DICT ? kGapTable : 1];
    if (WITH_DOT && ctl->flag != 0) return;
    if (DICT) {
        static_assert(kGapTable == kBlock, "one table entry per thread");
        gap_tbl[DICT ? threadIdx.x : 0] = G.gap_table[threadIdx.x];
        __syncthreads();
    }
    // a 16-bit code of the gap stream -> the gap (dictionary form: codes with the top bit set index the table)
    const auto gap_of = [&](uint32_t code) -> int {
        return (DICT && (code & 0x8000u)) ? static_cast<int>(gap_tbl[DICT ? (code & (kGapTable - 1)) : 0]) : static_cast<int>(code);
    };
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t gs = pick_slice(sel, (static_cast<int64_t>(blockIdx.x) << 2) + wave, G.n_gslices);
    double dot = 0.0;
    if (gs < G.n_gslices) {
        const int64_t off = G.gslice_off[gs];
        const int width = static_cast<int>((G.gslice_off[gs + 1] - off) >> 6);
        const double *__restrict__ vp = G.vals + kRelRows * off + lane;
        const uint32_t *__restrict__ wp = G.dwords + G.gslice_doff[gs] + lane;
        int c = __builtin_nontemporal_load(G.col0 + (gs << 6) + lane);
        double acc[kRelRows] = {0.0, 0.0, 0.0, 0.0};
        if (width > 0) {
            double xv[kRelRows];
            load_x4(x, c, xv);
#pragma unroll
            for (int p = 0; p < kRelRows; ++p) acc[p] = __builtin_nontemporal_load(vp + 64 * p) * xv[p];
        }
        const int nw = width / 2;
        int j = 0;
        // two words = four entries per trip: 16 value loads + 4 x quads in flight.  The gap words are fetched one
        // trip ahead so that the x addresses never wait on them, and the scheduling barrier keeps the compiler
        // from re-using x registers (it would serialise the quads behind vmcnt(0) waits)
        uint32_t w0 = 0, w1 = 0;
        if (2 * (j + 2) < width) { w0 = __builtin_nontemporal_load(wp + 64 * j); w1 = __builtin_nontemporal_load(wp + 64 * (j + 1)); }
        while (2 * (j + 2) < width) {
            const int c0 = c + gap_of(w0 & 0xffffu), c1 = c0 + gap_of(w0 >> 16);
            const int c2 = c1 + gap_of(w1 & 0xffffu), c3 = c2 + gap_of(w1 >> 16);
            double v[4][kRelRows], xv[4][kRelRows];
            load_x4(x, c0, xv[0]); load_x4(x, c1, xv[1]); load_x4(x, c2, xv[2]); load_x4(x, c3, xv[3]);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int p = 0; p < kRelRows; ++p) v[t][p] = __builtin_nontemporal_load(vp + 64 * (kRelRows * (2 * j + 1 + t) + p));
            c = c3;
            j += 2;
            if (2 * (j + 2) < width) { w0 = __builtin_nontemporal_load(wp + 64 * j); w1 = __builtin_nontemporal_load(wp + 64 * (j + 1)); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int p = 0; p < kRelRows; ++p) acc[p] = __builtin_fma(v[t][p], xv[t][p], acc[p]);
        }
        for (; j < nw; ++j) {
            const uint32_t w0 = __builtin_nontemporal_load(wp + 64 * j);
            const int c0 = c + gap_of(w0 & 0xffffu);
            const int c1 = c0 + gap_of(w0 >> 16);
            double xv[kRelRows];
            load_x4(x, c0, xv);
#pragma unroll
            for (int p = 0; p < kRelRows; ++p)
                acc[p] = __builtin_fma(__builtin_nontemporal_load(vp + 64 * (kRelRows * (2 * j + 1) + p)), xv[p], acc[p]);
            if (2 * j + 2 < width) {
                load_x4(x, c1, xv);
#pragma unroll
                for (int p = 0; p < kRelRows; ++p)
                    acc[p] = __builtin_fma(__builtin_nontemporal_load(vp + 64 * (kRelRows * (2 * j + 2) + p)), xv[p], acc[p]);
            }
            c = c1;
        }
        const int64_t r0 = ((gs << 6) + lane) * kRelRows;
#pragma unroll
        for (int p = 0; p < kRelRows; ++p)
            if (r0 + p < n_rows) {
                y[r0 + p] = acc[p];
                if (WITH_DOT && r0 + p < n_dot) dot = __builtin_fma(x[r0 + p], acc[p], dot);
            }
    }
    if (WITH_DOT) {
        const double t = block_sum(dot, sm);
        if (threadIdx.x == 0) partial[blockIdx.x] = t;
    }
}

// The same kernel for offsets further apart than 65535 (400^3: the z-neighbour is 159 201 rows away): one 32-bit gap per
// entry, 8 + 1 B per nonzero instead of 8 + 1/2 (the int32 row form costs 8 + 4).  Four entries per trip: 4 gap words one
// trip ahead, 16 value loads + 4 x quads in flight.
template <bool WITH_DOT>
__global__ void __launch_bounds__(kBlock) k_spmvr32(SellRDev G, int64_t n_rows, const double *__restrict__ x,
                                                     double *__restrict__ y, int64_t n_dot, double *partial, const CgCtl *ctl,
                                                     SliceSel sel)
{
    __shared__ double sm[4];
    if (WITH_DOT && ctl->flag != 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t gs = pick_slice(sel, (static_cast<int64_t>(blockIdx.x) << 2) + wave, G.n_gslices);
    double dot = 0.0;
    if (gs < G.n_gslices) {
        const int64_t off = G.gslice_off[gs];
        const int width = static_cast<int>((G.gslice_off[gs + 1] - off) >> 6);
        const double *__restrict__ vp = G.vals + kRelRows * off + lane;
        const uint32_t *__restrict__ wp = G.dwords + G.gslice_doff[gs] + lane;      // word k-1 = gap before entry k
        int c = __builtin_nontemporal_load(G.col0 + (gs << 6) + lane);
        double acc[kRelRows] = {0.0, 0.0, 0.0, 0.0};
        if (width > 0) {
            double xv[kRelRows];
            load_x4(x, c, xv);
#pragma unroll
            for (int p = 0; p < kRelRows; ++p) acc[p] = __builtin_nontemporal_load(vp + 64 * p) * xv[p];
        }
        int k = 1;
        uint32_t g0 = 0, g1 = 0, g2 = 0, g3 = 0;
        if (k + 4 <= width) {
            g0 = __builtin_nontemporal_load(wp + 64 * (k - 1)); g1 = __builtin_nontemporal_load(wp + 64 * k);
            g2 = __builtin_nontemporal_load(wp + 64 * (k + 1)); g3 = __builtin_nontemporal_load(wp + 64 * (k + 2));
        }
        while (k + 4 <= width) {
            const int c0 = c + static_cast<int>(g0), c1 = c0 + static_cast<int>(g1);
            const int c2 = c1 + static_cast<int>(g2), c3 = c2 + static_cast<int>(g3);
            double v[4][kRelRows], xv[4][kRelRows];
            load_x4(x, c0, xv[0]); load_x4(x, c1, xv[1]); load_x4(x, c2, xv[2]); load_x4(x, c3, xv[3]);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int p = 0; p < kRelRows; ++p) v[t][p] = __builtin_nontemporal_load(vp + 64 * (kRelRows * (k + t) + p));
            c = c3;
            k += 4;
            if (k + 4 <= width) {
                g0 = __builtin_nontemporal_load(wp + 64 * (k - 1)); g1 = __builtin_nontemporal_load(wp + 64 * k);
                g2 = __builtin_nontemporal_load(wp + 64 * (k + 1)); g3 = __builtin_nontemporal_load(wp + 64 * (k + 2));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int p = 0; p < kRelRows; ++p) acc[p] = __builtin_fma(v[t][p], xv[t][p], acc[p]);
        }
        for (; k < width; ++k) {
            c += static_cast<int>(__builtin_nontemporal_load(wp + 64 * (k - 1)));
            double xv[kRelRows];
            load_x4(x, c, xv);
#pragma unroll
            for (int p = 0; p < kRelRows; ++p)
                acc[p] = __builtin_fma(__builtin_nontemporal_load(vp + 64 * (kRelRows * k + p)), xv[p], acc[p]);
        }
        const int64_t r0 = ((gs << 6) + lane) * kRelRows;
#pragma unroll
        for (int p = 0; p < kRelRows; ++p)
            if (r0 + p < n_rows) {
                y[r0 + p] = acc[p];
                if (WITH_DOT && r0 + p < n_dot) dot = __builtin_fma(x[r0 + p], acc[p], dot);
            }
    }
    if (WITH_DOT) {
        const double t = block_sum(dot, sm);
        if (threadIdx.x == 0) partial[blockIdx.x] = t;
    }
}

// ---------------------------------------------------------------------------
// Node-block Jacobi (PETSc: -pc_type pbjacobi): M = the diagonal blocks of A over the
// row groups of k_spmvg (the dof rows of a node, 1..3 rows).  binv_q[i] holds row (i - r0) of the
// inverse block, column q; z_i = sum_q binv_q[i] * r[r0 + q].  One thread per GROUP in the vector
// kernels, so a block's residual is read and written by one thread.
// ---------------------------------------------------------------------------
// b_q[i] = A(i, r0 + q) for the rows of every group (0 outside the block)
__global__ void __launch_bounds__(kBlock) k_extract_blocks(SellDev A, const int32_t *group_row0, int64_t n_groups, double *b0,
                                                            double *b1, double *b2)
{
    const int64_t g = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (g >= n_groups) return;
    const int r0 = group_row0[g], sz = group_row0[g + 1] - r0;
    double *b[kGroupRows] = {b0, b1, b2};
    for (int p = 0; p < sz; ++p)
        for (int q = 0; q < kGroupRows; ++q) {
            double v = 0.0;
            if (q < sz) {
                const int64_t sl = find_slot(A, r0 + p, r0 + q);
                if (sl >= 0) v = A.vals[sl];
            }
            b[q][r0 + p] = v;
        }
}

// in place: block -> inverse block (adjugate / determinant, fixed operation order)
__global__ void __launch_bounds__(kBlock) k_invert_blocks(const int32_t *group_row0, int64_t n_groups, double *b0, double *b1,
                                                           double *b2)
{
    const int64_t g = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (g >= n_groups) return;
    const int r0 = group_row0[g], sz = group_row0[g + 1] - r0;
    if (sz == 1) {
        b0[r0] = 1.0 / b0[r0];
    } else if (sz == 2) {
        const double a = b0[r0], b = b1[r0], c = b0[r0 + 1], d = b1[r0 + 1];
        const double di = 1.0 / (a * d - b * c);
        b0[r0] = d * di; b1[r0] = -b * di;
        b0[r0 + 1] = -c * di; b1[r0 + 1] = a * di;
    } else if (sz == 3) {
        const double a00 = b0[r0], a01 = b1[r0], a02 = b2[r0];
        const double a10 = b0[r0 + 1], a11 = b1[r0 + 1], a12 = b2[r0 + 1];
        const double a20 = b0[r0 + 2], a21 = b1[r0 + 2], a22 = b2[r0 + 2];
        const double c00 = a11 * a22 - a12 * a21, c01 = a12 * a20 - a10 * a22, c02 = a10 * a21 - a11 * a20;
        const double di = 1.0 / (a00 * c00 + a01 * c01 + a02 * c02);
        b0[r0] = c00 * di; b1[r0] = (a02 * a21 - a01 * a22) * di; b2[r0] = (a01 * a12 - a02 * a11) * di;
        b0[r0 + 1] = c01 * di; b1[r0 + 1] = (a00 * a22 - a02 * a20) * di; b2[r0 + 1] = (a02 * a10 - a00 * a12) * di;
        b0[r0 + 2] = c02 * di; b1[r0 + 2] = (a01 * a20 - a00 * a21) * di; b2[r0 + 2] = (a00 * a11 - a01 * a10) * di;
    }
}

// row_grp[i] = first row of i's group | (group size << 30): lets the vector kernels run one thread per ROW
// (coalesced streams) and still see the whole block
__global__ void __launch_bounds__(kBlock) k_fill_row_groups(const int32_t *group_row0, int64_t n_groups, uint32_t *row_grp)
{
    const int64_t g = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (g >= n_groups) return;
    const int r0 = group_row0[g], sz = group_row0[g + 1] - r0;
    for (int q = 0; q < sz; ++q) row_grp[r0 + q] = static_cast<uint32_t>(r0) | (static_cast<uint32_t>(sz) << 30);
}

// Multi-rank consistency of the row groups: sig_i = 1 + (i - r0) + 4 * size must be the same on every rank
// that holds dof i.  After interface sums S1 = sum sig, S2 = sum sig^2 (small integers: exact), all ranks
// agree on dof i iff S2 = sig_i * S1 on every one of them.
__global__ void __launch_bounds__(kBlock) k_group_sig(const uint32_t *row_grp, int64_t n, double *s1, double *s2)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= n) return;
    const uint32_t rg = row_grp[i];
    const double sig = 1.0 + static_cast<double>(i - static_cast<int64_t>(rg & 0x3fffffffu)) + 4.0 * static_cast<double>(rg >> 30);
    s1[i] = sig;
    s2[i] = sig * sig;
}

__global__ void __launch_bounds__(kBlock) k_group_sig_check(const uint32_t *row_grp, int64_t n, const double *s1, const double *s2,
                                                             double *bad)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= n) return;
    const uint32_t rg = row_grp[i];
    const double sig = 1.0 + static_cast<double>(i - static_cast<int64_t>(rg & 0x3fffffffu)) + 4.0 * static_cast<double>(rg >> 30);
    if (s2[i] != sig * s1[i]) bad[0] = 1.0;     // benign race: every writer stores the same value
}

// z_i = sum_q binv_q[i] * rr[q], q ascending
__device__ __forceinline__ double block_row_apply(double bi0, double bi1, double bi2, int sz, const double (&rr)[kGroupRows])
{
    double t = bi0 * rr[0];
    if (sz > 1) t = __builtin_fma(bi1, rr[1], t);
    if (sz > 2) t = __builtin_fma(bi2, rr[2], t);
    return t;
}

__global__ void __launch_bounds__(kBlock) k_cg_init_b(int64_t n, const uint32_t *__restrict__ row_grp, int64_t n_owned,
                                                       const double *__restrict__ b, const double *__restrict__ b0,
                                                       const double *__restrict__ b1, const double *__restrict__ b2,
                                                       double *__restrict__ x, double *__restrict__ r, double *__restrict__ p,
                                                       double *part_rz, double *part_zz)
{
    __shared__ double sm[4];
    double rz = 0.0, zz = 0.0;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n;
         i += static_cast<int64_t>(gridDim.x) * kBlock) {
        const uint32_t rg = row_grp[i];
        const int r0 = static_cast<int>(rg & 0x3fffffffu), sz = static_cast<int>(rg >> 30);
        double rr[kGroupRows] = {0.0, 0.0, 0.0};
        for (int q = 0; q < sz; ++q) rr[q] = b[r0 + q];
        const double ri = b[i], zi = block_row_apply(b0[i], b1[i], b2[i], sz, rr);
        x[i] = 0.0;
        r[i] = ri;
        p[i] = zi;
        if (i < n_owned) { rz = __builtin_fma(ri, zi, rz); zz = __builtin_fma(zi, zi, zz); }
    }
    const double a = block_sum(rz, sm), c = block_sum(zz, sm);
    if (threadIdx.x == 0) { part_rz[blockIdx.x] = a; part_zz[blockIdx.x] = c; }
}

// the residual is ping-ponged (r_old -> r_new): a thread recomputes the new residual of its block mates
// from r_old instead of waiting for them
__global__ void __launch_bounds__(kBlock) k_cg_update_b(CgCtl *ctl, int it, int64_t n, const uint32_t *__restrict__ row_grp,
                                                         int64_t n_owned, const double *part_pw, int nparts,
                                                         const double *reduced_pw, const double *__restrict__ p,
                                                         const double *__restrict__ w, const double *__restrict__ b0,
                                                         const double *__restrict__ b1, const double *__restrict__ b2,
                                                         double *__restrict__ x, const double *__restrict__ r_old,
                                                         double *__restrict__ r_new, double *__restrict__ z_out, double *part_rz,
                                                         double *part_zz)
{
    __shared__ double sm[4];
    if (ctl->flag != 0) return;
    const double pw = reduced_pw ? *reduced_pw : sum_partials(part_pw, nparts, sm);
    if (!(pw > 0.0)) {                      // KSP_DIVERGED_INDEFINITE_MAT
        if (blockIdx.x == 0 && threadIdx.x == 0) { ctl->its = it; }
        part_rz[blockIdx.x] = 0.0; part_zz[blockIdx.x] = -1.0;
        return;
    }
    const double alpha = ctl->beta[it & 1] / pw;
    double rz = 0.0, zz = 0.0;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n;
         i += static_cast<int64_t>(gridDim.x) * kBlock) {
        const uint32_t rg = row_grp[i];
        const int r0 = static_cast<int>(rg & 0x3fffffffu), sz = static_cast<int>(rg >> 30);
        double rr[kGroupRows] = {0.0, 0.0, 0.0};
        for (int q = 0; q < sz; ++q) rr[q] = __builtin_fma(-alpha, w[r0 + q], r_old[r0 + q]);
        const double ri = __builtin_fma(-alpha, w[i], r_old[i]);
        x[i] = __builtin_fma(alpha, p[i], x[i]);
        r_new[i] = ri;
        const double zi = block_row_apply(b0[i], b1[i], b2[i], sz, rr);
        z_out[i] = zi;                          // stored: re-applying the block in k_cg_direction_b would cost 36 B/row
        if (i < n_owned) { rz = __builtin_fma(ri, zi, rz); zz = __builtin_fma(zi, zi, zz); }
    }
    const double a = block_sum(rz, sm), c = block_sum(zz, sm);
    if (threadIdx.x == 0) { part_rz[blockIdx.x] = a; part_zz[blockIdx.x] = c; }
}

__global__ void __launch_bounds__(kBlock) k_cg_direction_b(CgCtl *ctl, int it, int64_t n, const double *part_rz,
                                                            const double *part_zz, int nparts, const double *reduced,
                                                            const double *__restrict__ z, double *__restrict__ p, double *hist,
                                                            int hist_cap, int maxits)
{
    __shared__ double sm[4];
    if (ctl_finished_before(ctl, it)) return;
    double rz, zz;
    if (reduced) { rz = reduced[0]; zz = reduced[1]; }
    else { rz = sum_partials(part_rz, nparts, sm); zz = sum_partials(part_zz, nparts, sm); }
    const bool lead = blockIdx.x == 0 && threadIdx.x == 0;
    if (zz < 0.0) {
        if (lead) ctl_publish(ctl, -10, it + 1);
        return;
    }
    const double rn = sqrt(zz);
    const double beta_old = ctl->beta[it & 1];
    int flag = 0;
    if (rn <= ctl->ttol) flag = 2;
    else if (rn >= ctl->dtol * ctl->rn0) flag = -4;
    else if (rz < 0.0) flag = -8;
    else if (it + 1 >= maxits) flag = -3;
    if (lead) {
        ctl->beta[(it + 1) & 1] = rz;
        ctl->rn = rn;
        if (it + 1 < hist_cap) hist[it + 1] = rn;
        ctl_publish(ctl, flag, it + 1);
    }
    if (flag != 0) return;
    const double bb = rz / beta_old;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n;
         i += static_cast<int64_t>(gridDim.x) * kBlock)
        p[i] = __builtin_fma(bb, p[i], z[i]);
}

// ---------------------------------------------------------------------------
// neighbour exchange (multi-GPU, sub-assembled rows)
// ---------------------------------------------------------------------------
// send[i] = v[send_lidx[i]]: the partials of the shared dofs, one segment per neighbour (ascending rank)
__global__ void __launch_bounds__(kBlock) k_pack_send(const double *__restrict__ v, const int32_t *__restrict__ send_lidx,
                                                       int64_t n_send, double *__restrict__ send, const CgCtl *ctl)
{
    if (ctl && ctl->flag != 0) return;
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i < n_send) send[i] = v[send_lidx[i]];
}

// v[dof] = sum of the partials of all ranks that hold the dof, ADDED IN ASCENDING RANK ORDER on every rank (floating
// point addition is not associative: a fixed order gives every rank the same bits, so the replicated ghost entries
// stay identical).  Contributions of shared dof j: src[ptr[j]..ptr[j+1]), a position in the receive buffer, or -1
// for this rank's own partial.
__global__ void __launch_bounds__(kBlock) k_unpack_sum(double *__restrict__ v, const int32_t *__restrict__ sh_lidx,
                                                        const int32_t *__restrict__ sh_ptr, const int32_t *__restrict__ sh_src,
                                                        int64_t n_sh, const double *__restrict__ recv, const CgCtl *ctl)
{
    if (ctl && ctl->flag != 0) return;
    const int64_t j = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (j >= n_sh) return;
    const int32_t l = sh_lidx[j];
    const double own = v[l];
    double acc = 0.0;
    for (int32_t k = sh_ptr[j]; k < sh_ptr[j + 1]; ++k) {
        const int32_t q = sh_src[k];
        acc += q < 0 ? own : recv[q];
    }
    v[l] = acc;
}

// slice_flag[slice of row] = 1 for every shared row (rows_per_slice = 64 rows, or 256 for the relative row groups;
// row_grp != nullptr: the row-grouped form, slice = group / 64 with the group looked up per row)
__global__ void __launch_bounds__(kBlock) k_mark_boundary_slices(const int32_t *__restrict__ sh_lidx, int64_t n_sh,
                                                                  int shift, const int32_t *__restrict__ row_group,
                                                                  char *slice_flag)
{
    const int64_t j = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (j >= n_sh) return;
    const int64_t r = sh_lidx[j];
    const int64_t unit = row_group ? static_cast<int64_t>(row_group[r]) : r;
    slice_flag[unit >> shift] = 1;
}

// row -> index of its group (k_spmvg's lanes), from the group start rows
__global__ void __launch_bounds__(kBlock) k_row_group_index(const int32_t *group_row0, int64_t n_groups, int32_t *row_group)
{
    const int64_t g = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (g >= n_groups) return;
    for (int32_t r = group_row0[g]; r < group_row0[g + 1]; ++r) row_group[r] = static_cast<int32_t>(g);
}

}  // namespace pfem
