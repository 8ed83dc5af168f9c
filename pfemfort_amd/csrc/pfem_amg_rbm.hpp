// pfem_amg_rbm.hpp -- gfx950 kernels of the rigid-body-mode coarse space of -pc_type gamg for problems with DIM displacement
// dofs per node (tetraelasticityparallelimpl1.F:894-902, 993: the reference's beam; PETSc reaches the same space through
// MatSetNearNullSpace / PCSetCoordinates).  Host side and the why: pfem_amg.inc ("rigid-body modes").
//
// A level whose dofs come in nodes of FB dofs each (FB = DIM on the assembled matrix, FB = CB below it) is coarsened into
// nodes of CB = DIM + NR dofs: DIM translations T and NR rotations W (3 in space, 1 in the plane) about the aggregate's
// centroid.  For a fine node at offset r from the centroid of its aggregate
//      u = T + W x r          (plane: u = T + Wz (-ry, rx))            and, when the fine node carries rotations itself,  w = W,
// i.e. P_i = [ I  S(r_i) ; 0  I ] with S(r) W = W x r.  Everything is node-block arithmetic on the wave-sliced scalar
// storage: a level is "block regular" (k_rbm_check_blocks): the FB rows of a node have the same length, and their columns
// come in runs of FB consecutive dofs of one node, so entry (c, d) of node block (i, jj) sits at a closed-form slot.
#pragma once

namespace pfem {

template <int DIM> struct RbmDims { static constexpr int NR = DIM == 3 ? 3 : 1; static constexpr int CB = DIM + NR; };

// g[0..NR) += S(r)^T f  (f: DIM forces on a node at offset r -> moments about the centroid: r x f)
template <int DIM>
__device__ __forceinline__ void rbm_moment(const double *r, const double *f, double *g)
{
    if (DIM == 3) {
        g[0] += r[1] * f[2] - r[2] * f[1];
        g[1] += r[2] * f[0] - r[0] * f[2];
        g[2] += r[0] * f[1] - r[1] * f[0];
    } else {
        g[0] += r[0] * f[1] - r[1] * f[0];
    }
}
// u[0..DIM) = W x r
template <int DIM>
__device__ __forceinline__ void rbm_spin(const double *r, const double *w, double *u)
{
    if (DIM == 3) {
        u[0] = w[1] * r[2] - w[2] * r[1];
        u[1] = w[2] * r[0] - w[0] * r[2];
        u[2] = w[0] * r[1] - w[1] * r[0];
    } else {
        u[0] = -w[0] * r[1];
        u[1] = w[0] * r[0];
    }
}

__device__ __forceinline__ int64_t rbm_row_base(const SellDev &A, int64_t row) { return A.slice_off[row >> 6] + (row & 63); }

// ---- symbolic phase ------------------------------------------------------------------------------------------------
// Is the level block regular?  One thread per node: the bs rows of node i have one length, a multiple of bs, and entry
// bs*jj + d of row bs*i + c is dof d of the node whose dof 0 entry bs*jj of row bs*i names.  deg[i] = blocks in node row i.
__global__ void __launch_bounds__(kBlock) k_rbm_check_blocks(SellDev A, int bs, int64_t n_nodes, int64_t *__restrict__ deg, int *__restrict__ bad)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i > n_nodes) return;
    if (i == n_nodes) { deg[i] = 0; return; }
    const int64_t row0 = static_cast<int64_t>(bs) * i;
    const int len = A.rowlen[row0];
    bool ok = len % bs == 0;
    for (int c = 1; c < bs; ++c) ok = ok && A.rowlen[row0 + c] == len;
    if (ok) {
        for (int c = 0; c < bs; ++c) {
            const int64_t base = rbm_row_base(A, row0 + c);
            for (int k = 0; k < len; k += bs) {
                const int32_t c0 = A.cols[base + 64LL * k];
                ok = ok && c0 % bs == 0 && c0 < bs * n_nodes;
                for (int d = 1; d < bs; ++d) ok = ok && A.cols[base + 64LL * (k + d)] == c0 + d;
            }
        }
        if (ok && bs > 1) {       // the rows of a node name the same nodes
            const int64_t b0 = rbm_row_base(A, row0), b1 = rbm_row_base(A, row0 + 1);
            for (int k = 0; k < len; k += bs) ok = ok && A.cols[b0 + 64LL * k] == A.cols[b1 + 64LL * k];
        }
    }
    deg[i] = ok ? len / bs : 0;
    if (!ok) *bad = 1;
}

// the same check with one thread per ROW (consecutive lanes read consecutive rows of the wave-sliced storage: the per-node
// form above walks bs rows per lane, i.e. every line bs times with 1/bs of its lanes -- 1.1 + 1.8 ms on the beam's first two levels)
__global__ void __launch_bounds__(kBlock) k_rbm_check_rows(SellDev A, int bs, int64_t n_nodes, int64_t *__restrict__ deg, int *__restrict__ bad)
{
    const int64_t r = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    const int64_t n_rows = static_cast<int64_t>(bs) * n_nodes;
    if (r > n_rows) return;
    if (r == n_rows) { deg[n_nodes] = 0; return; }
    const int64_t node = r / bs;
    const int c = static_cast<int>(r - node * bs);
    const int64_t row0 = node * bs;
    const int len = A.rowlen[r];
    bool ok = len % bs == 0 && len == A.rowlen[row0];
    if (ok) {
        const int64_t base = rbm_row_base(A, r), base0 = rbm_row_base(A, row0);
        int32_t c0 = 0;
        for (int k = 0; k < len; ++k) {
            const int32_t col = A.cols[base + 64LL * k];
            const int d = k % bs;
            if (d == 0) {
                c0 = col;
                ok = ok && c0 % bs == 0 && c0 < n_rows && (c == 0 || A.cols[base0 + 64LL * k] == c0);       // the rows of a node name the same nodes
            } else {
                ok = ok && col == c0 + d;
            }
        }
    }
    if (c == 0) deg[node] = ok ? len / bs : 0;
    if (!ok) *bad = 1;
}
// ... and the graph's weights with DIM lanes per node (the DIM displacement rows of the node: adjacent lanes, adjacent rows),
// 64 / DIM nodes per wave, the rows' partial sums of squares combined by shuffles
template <int DIM>
__global__ void __launch_bounds__(kBlock) k_rbm_graph_fill_rows(SellDev A, int bs, int64_t n_nodes, const int64_t *__restrict__ gptr,
                                                                 int32_t *__restrict__ gcol, int32_t *__restrict__ brow, double *__restrict__ gw,
                                                                 double *__restrict__ gdiag)
{
    constexpr int NPW = 64 / DIM;                        // nodes per wave
    const int64_t wave = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const int64_t i = wave * NPW + lane / DIM;
    const int c = lane % DIM;
    if (lane >= NPW * DIM || i >= n_nodes) return;
    const int64_t base = rbm_row_base(A, static_cast<int64_t>(bs) * i + c);
    const int64_t q0 = gptr[i];
    const int deg = static_cast<int>(gptr[i + 1] - q0);
    bool have_diag = false;
    for (int jj = 0; jj < deg; ++jj) {
        double s2 = 0.0;
#pragma unroll
        for (int d = 0; d < DIM; ++d) {
            const double v = A.vals[base + 64LL * (bs * jj + d)];
            s2 += v * v;
        }
        const double t1 = __shfl_down(s2, 1, 64), t2 = DIM == 3 ? __shfl_down(s2, 2, 64) : 0.0;
        if (c == 0) {
            s2 += t1 + t2;
            const int32_t j = A.cols[base + 64LL * (bs * jj)] / bs;
            gcol[q0 + jj] = j;
            brow[q0 + jj] = static_cast<int32_t>(i);
            if (j == i) { gdiag[i] = sqrt(s2); gw[q0 + jj] = 0.0; have_diag = true; }
            else gw[q0 + jj] = -sqrt(s2);
        }
    }
    if (c == 0 && !have_diag) gdiag[i] = 0.0;
}
// node graph of a block-regular level: columns, the node row of every block, and the strength weights of the pairing
// (Frobenius norm of the DIM x DIM displacement part of a node block: -norm off the diagonal, norm on it -- what
// k_amg_graph_from_keys makes of the squared entries)
__global__ void __launch_bounds__(kBlock) k_rbm_graph_fill(SellDev A, int bs, int dim, int64_t n_nodes, const int64_t *__restrict__ gptr,
                                                            int32_t *__restrict__ gcol, int32_t *__restrict__ brow, double *__restrict__ gw,
                                                            double *__restrict__ gdiag)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= n_nodes) return;
    const int64_t row0 = static_cast<int64_t>(bs) * i;
    const int64_t q0 = gptr[i];
    const int deg = static_cast<int>(gptr[i + 1] - q0);
    bool have_diag = false;
    for (int jj = 0; jj < deg; ++jj) {
        double s2 = 0.0;
        int32_t j = 0;
        for (int c = 0; c < dim; ++c) {
            const int64_t base = rbm_row_base(A, row0 + c);
            if (c == 0) j = A.cols[base + 64LL * (bs * jj)] / bs;
            for (int d = 0; d < dim; ++d) {
                const double v = A.vals[base + 64LL * (bs * jj + d)];
                s2 += v * v;
            }
        }
        gcol[q0 + jj] = j;
        brow[q0 + jj] = static_cast<int32_t>(i);
        if (j == i) { gdiag[i] = sqrt(s2); gw[q0 + jj] = 0.0; have_diag = true; }
        else gw[q0 + jj] = -sqrt(s2);
    }
    if (!have_diag) gdiag[i] = 0.0;
}

// ---- node bricks in one step (amg_node_bricks) ------------------------------------------------------------------------------
// The pairing passes on a lattice whose lines are full end in bricks that a table per axis describes: position -> brick
// coordinate (pairs 2k | 2k + 1; the node an odd line leaves over joins the pair next to it, at either end).  One check on
// the level's node graph replaces the passes and their sorted aggregate graphs: every coupling to a neighbour one position
// away inside the own brick is at least a quarter of the node's strongest (what a pass asks of a pair, k_amg_lat_pick /
// k_amg_lat_absorb); else *fail and the passes take the level.
__global__ void __launch_bounds__(kBlock) k_rbm_lat_check(int64_t nn, const int64_t *__restrict__ gptr, const int32_t *__restrict__ gcol,
                                                           const double *__restrict__ gw, const double *__restrict__ gdiag,
                                                           const int32_t *__restrict__ pos, const int32_t *__restrict__ table, int *__restrict__ fail)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= nn) return;
    const int32_t pi = pos[i];
    const int px = pi & 0x3ff, py = (pi >> 10) & 0x3ff, pz = (pi >> 20) & 0x3ff;
    const int bx = table[px], by = table[1024 + py], bz = table[2048 + pz];
    const double di = gdiag[i];
    double smax = 0.0, smin_in = 1e300;
    for (int64_t q = gptr[i]; q < gptr[i + 1]; ++q) {
        const int32_t j = gcol[q];
        if (j == static_cast<int32_t>(i)) continue;
        const double sij = amg_strength(gw[q], di, gdiag[j]);
        smax = fmax(smax, sij);
        const int32_t pj = pos[j];
        const int qx = pj & 0x3ff, qy = (pj >> 10) & 0x3ff, qz = (pj >> 20) & 0x3ff;
        const int dist = abs(qx - px) + abs(qy - py) + abs(qz - pz);
        if (dist == 1 && table[qx] == bx && table[1024 + qy] == by && table[2048 + qz] == bz) smin_in = fmin(smin_in, sij);
    }
    if (smin_in < 1e300 && !(smin_in > 0.0 && smin_in >= 0.25 * smax)) atomicAdd(fail, 1);       // (*fail: the number of such nodes)
}
// aggregate of every node = its brick, numbered in z, y, x order inside the box of bricks (the lines are full: every brick is
// occupied); blo / nbk: first brick and number of bricks along every axis
__global__ void __launch_bounds__(kBlock) k_rbm_lat_assign(int64_t nn, const int32_t *__restrict__ pos, const int32_t *__restrict__ table, int blo0,
                                                            int blo1, int blo2, int nbk0, int nbk1, int32_t *__restrict__ node_agg)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= nn) return;
    const int32_t p = pos[i];
    const int bx = table[p & 0x3ff] - blo0, by = table[1024 + ((p >> 10) & 0x3ff)] - blo1, bz = table[2048 + ((p >> 20) & 0x3ff)] - blo2;
    node_agg[i] = bx + nbk0 * (by + nbk1 * bz);
}
__global__ void __launch_bounds__(kBlock) k_rbm_lat_coarse_pos(int64_t na, int blo0, int blo1, int blo2, int nbk0, int nbk1, int32_t *__restrict__ pos_c)
{
    const int64_t a = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (a >= na) return;
    const int bx = static_cast<int>(a % nbk0) + blo0, by = static_cast<int>((a / nbk0) % nbk1) + blo1, bz = static_cast<int>(a / (static_cast<int64_t>(nbk0) * nbk1)) + blo2;
    pos_c[a] = bx | (by << 10) | (bz << 20);
}

// The coarse block pattern of a node-brick level without a sort (the analogue of k_lat_codes_*): the aggregates are the bricks of
// a box, numbered bx + nbk0 (by + nbk1 bz), so the coarse column of a fine block is one of the 27 bricks around its row's brick.
// One thread per coarse node walks its member nodes' block rows twice: counts per offset code (LDS, [code][thread]) -> the row's
// coarse blocks and fine blocks; after two scans the keys of the coarse blocks (ascending code = ascending coarse column), the
// start of every coarse block's list and the fine blocks themselves, member nodes ascending, blocks of a node row ascending --
// the order the sorted form gives (stable sort of the block index by (coarse row, coarse column)).
constexpr int kRbmCodes = 27;
__device__ __forceinline__ int rbm_brick_code(int32_t I, int32_t J, int nbk0, int nbk1)
{
    const int ix = I % nbk0, iy = (I / nbk0) % nbk1, iz = I / (nbk0 * nbk1);
    const int jx = J % nbk0, jy = (J / nbk0) % nbk1, jz = J / (nbk0 * nbk1);
    const int dx = jx - ix, dy = jy - iy, dz = jz - iz;
    if (dx < -1 || dx > 1 || dy < -1 || dy > 1 || dz < -1 || dz > 1) return -1;
    return (dx + 1) + 3 * (dy + 1) + 9 * (dz + 1);
}
__global__ void __launch_bounds__(kBlock) k_rbm_codes_count(int64_t na, const int32_t *__restrict__ mem_ptr, const int32_t *__restrict__ mem_idx,
                                                             const int64_t *__restrict__ gptr, const int32_t *__restrict__ gcol,
                                                             const int32_t *__restrict__ node_agg, int nbk0, int nbk1, uint16_t *__restrict__ code_cnt,
                                                             int32_t *__restrict__ n_entries, int32_t *__restrict__ n_pairs, int *__restrict__ fail)
{
    __shared__ uint16_t cnt[kRbmCodes][kBlock];
    const int64_t I = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (I == na) { n_entries[na] = 0; n_pairs[na] = 0; }
    if (I >= na) return;
    const int t = threadIdx.x;
#pragma unroll
    for (int c = 0; c < kRbmCodes; ++c) cnt[c][t] = 0;
    int pairs = 0;
    for (int32_t m = mem_ptr[I]; m < mem_ptr[I + 1]; ++m) {
        const int64_t i = mem_idx[m];
        for (int64_t q = gptr[i]; q < gptr[i + 1]; ++q) {
            const int c = rbm_brick_code(static_cast<int32_t>(I), node_agg[gcol[q]], nbk0, nbk1);
            if (c < 0) { *fail = 1; continue; }           // a coupling beyond the neighbouring bricks: the sorted form takes the level
            ++cnt[c][t];
            ++pairs;
        }
    }
    int entries = 0;
#pragma unroll
    for (int c = 0; c < kRbmCodes; ++c) {
        const uint16_t v = cnt[c][t];
        code_cnt[static_cast<int64_t>(c) * na + I] = v;
        entries += v != 0;
    }
    n_entries[I] = entries;
    n_pairs[I] = pairs;
    if (pairs > 0xffff) *fail = 1;
}
__global__ void __launch_bounds__(kBlock) k_rbm_codes_fill(int64_t na, const int32_t *__restrict__ mem_ptr, const int32_t *__restrict__ mem_idx,
                                                            const int64_t *__restrict__ gptr, const int32_t *__restrict__ gcol,
                                                            const int32_t *__restrict__ node_agg, int nbk0, int nbk1,
                                                            const uint16_t *__restrict__ code_cnt, const int32_t *__restrict__ entry_off,
                                                            const int32_t *__restrict__ pair_off, uint64_t *__restrict__ ukeys,
                                                            int64_t *__restrict__ src_ptr, int32_t *__restrict__ src_slot)
{
    __shared__ uint16_t at[kRbmCodes][kBlock];
    const int64_t I = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (I >= na) return;
    const int t = threadIdx.x;
    const int ix = static_cast<int>(I % nbk0), iy = static_cast<int>((I / nbk0) % nbk1), iz = static_cast<int>(I / (static_cast<int64_t>(nbk0) * nbk1));
    const int64_t p0 = pair_off[I];
    int64_t e = entry_off[I];
    int run = 0;
#pragma unroll
    for (int c = 0; c < kRbmCodes; ++c) {
        const int v = code_cnt[static_cast<int64_t>(c) * na + I];
        at[c][t] = static_cast<uint16_t>(run);
        if (v != 0) {
            const int32_t J = (ix + c % 3 - 1) + nbk0 * ((iy + (c / 3) % 3 - 1) + nbk1 * (iz + c / 9 - 1));
            ukeys[e] = (static_cast<uint64_t>(I) << 32) | static_cast<uint32_t>(J);
            src_ptr[e] = p0 + run;
            ++e;
        }
        run += v;
    }
    if (I == na - 1) src_ptr[e] = p0 + run;
    for (int32_t m = mem_ptr[I]; m < mem_ptr[I + 1]; ++m) {
        const int64_t i = mem_idx[m];
        for (int64_t q = gptr[i]; q < gptr[i + 1]; ++q) {
            const int c = rbm_brick_code(static_cast<int32_t>(I), node_agg[gcol[q]], nbk0, nbk1);
            src_slot[p0 + at[c][t]++] = static_cast<int32_t>(q);
        }
    }
}

// dof -> (node, component) of a level whose dofs come bs to the node
__global__ void __launch_bounds__(kBlock) k_rbm_node_comp(int64_t n, int bs, int32_t *__restrict__ node_of, int32_t *__restrict__ comp_of)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= n) return;
    node_of[i] = static_cast<int32_t>(i / bs);
    comp_of[i] = static_cast<int32_t>(i % bs);
}
// coarse dof that carries the same component of the same aggregate (the translation part of P: what pfem_solver_amg_aggregates reports)
__global__ void __launch_bounds__(kBlock) k_rbm_dof_agg(int64_t n, int bs, int cb, const int32_t *__restrict__ node_agg, int32_t *__restrict__ agg)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i < n) agg[i] = cb * node_agg[i / bs] + static_cast<int32_t>(i % bs);
}

// coordinates of the nodes of the assembled matrix: node of dof l = l / ndof (checked by the caller through the element
// dof array: a node's dofs are consecutive, all free or all constrained)
__global__ void __launch_bounds__(kBlock) k_rbm_mesh_xyz(MeshDev m, int64_t n_nodes, double *__restrict__ xyz, int *__restrict__ bad)
{
    const int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (t >= m.nElem * m.npe) return;
    const int64_t e = t % m.nElem;
    const int a = static_cast<int>(t / m.nElem);
    const int32_t nd = m.conn[a * m.nElem + e];
    const int32_t l0 = m.edof[(a * m.ndof) * m.nElem + e];
    bool ok = true;
    for (int d = 1; d < m.ndof; ++d) {
        const int32_t l = m.edof[(a * m.ndof + d) * m.nElem + e];
        ok = ok && (l0 < 0 ? l < 0 : l == l0 + d);
    }
    if (l0 >= 0) ok = ok && l0 % m.ndof == 0 && l0 / m.ndof < n_nodes;
    if (!ok) { *bad = 1; return; }
    if (l0 < 0) return;
    const int64_t node = l0 / m.ndof;
    for (int d = 0; d < 3; ++d) xyz[d * n_nodes + node] = d < m.ndim ? m.xyz[static_cast<int64_t>(d) * m.nNode + nd] : 0.0;
}

// centroid of every aggregate (members in ascending order: one fixed sum) and whether its nodes span enough space for the
// rotations to be independent of the translations and of each other: in space the nodes must not be collinear (the sum of
// the principal 2x2 minors of sum r r^T must not vanish), in the plane there must be two of them.  A degenerate aggregate
// keeps its translations only (its nodes get the offset 0, so its rotation columns vanish; k_rbm_galerkin puts 1 on the
// diagonal of the empty rows).
__global__ void __launch_bounds__(kBlock) k_rbm_centroids(int64_t na, int dim, int64_t nn, const int32_t *__restrict__ mem_ptr,
                                                           const int32_t *__restrict__ mem_idx, const double *__restrict__ xyz, int64_t nn_c,
                                                           double *__restrict__ cen, int32_t *__restrict__ rot_ok, int check)
{
    const int64_t a = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (a >= na) return;
    const int q0 = mem_ptr[a], q1 = mem_ptr[a + 1];
    double c[3] = {0.0, 0.0, 0.0};
    for (int q = q0; q < q1; ++q) {
        const int64_t i = mem_idx[q];
        for (int d = 0; d < 3; ++d) c[d] += xyz[d * nn + i];
    }
    const double inv = q1 > q0 ? 1.0 / static_cast<double>(q1 - q0) : 0.0;
    for (int d = 0; d < 3; ++d) { c[d] *= inv; cen[d * nn_c + a] = c[d]; }
    int ok = 1;
    if (check) {
        double sxx = 0, syy = 0, szz = 0, sxy = 0, sxz = 0, syz = 0;
        for (int q = q0; q < q1; ++q) {
            const int64_t i = mem_idx[q];
            const double x = xyz[i] - c[0], y = xyz[nn + i] - c[1], z = xyz[2 * nn + i] - c[2];
            sxx += x * x; syy += y * y; szz += z * z; sxy += x * y; sxz += x * z; syz += y * z;
        }
        const double tr = sxx + syy + szz;
        if (dim == 3) {
            const double m2 = (sxx * syy - sxy * sxy) + (sxx * szz - sxz * sxz) + (syy * szz - syz * syz);
            ok = tr > 0.0 && m2 > 1e-8 * tr * tr;
        } else {
            ok = tr > 0.0;
        }
    }
    rot_ok[a] = ok;
}
// offset of every node from the centroid of its aggregate (0 in a degenerate aggregate)
__global__ void __launch_bounds__(kBlock) k_rbm_offsets(int64_t nn, const int32_t *__restrict__ node_agg, const double *__restrict__ xyz,
                                                         int64_t nn_c, const double *__restrict__ cen, const int32_t *__restrict__ rot_ok,
                                                         double *__restrict__ roff)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= nn) return;
    const int32_t a = node_agg[i];
    const bool ok = rot_ok[a] != 0;
    for (int d = 0; d < 3; ++d) roff[d * nn + i] = ok ? xyz[d * nn + i] - cen[d * nn_c + a] : 0.0;
}

// ---- the same on several ranks (one hierarchy across the ranks): a rank forms aggregates inside its owned nodes; the ghost
// nodes learn theirs through the level's sum-exchanges (the dof-level aggregate numbers as ever, then the centroid and the
// rotation verdict of the aggregate: the owner sends them on the node's displacement dofs, the ghosts send zeros)
__global__ void __launch_bounds__(kBlock) k_rbm_node_agg_from_dofs(int64_t nn, int bs, int cb, const int32_t *__restrict__ agg, int32_t *__restrict__ node_agg)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i < nn) node_agg[i] = agg[bs * i] / cb;
}
// v[bs*i + d] = centroid coordinate d of the aggregate of owned node i (d < dim), w[bs*i] = its rotation verdict; 0 elsewhere
__global__ void __launch_bounds__(kBlock) k_rbm_centroid_vectors(int64_t nn, int64_t nn_own, int bs, int dim, const int32_t *__restrict__ node_agg,
                                                                  int64_t nn_c, const double *__restrict__ cen, const int32_t *__restrict__ rot_ok,
                                                                  double *__restrict__ v, double *__restrict__ w)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= nn) return;
    const bool own = i < nn_own;
    const int32_t a = own ? node_agg[i] : 0;
    for (int d = 0; d < bs; ++d) {
        v[bs * i + d] = (own && d < dim) ? cen[d * nn_c + a] : 0.0;
        w[bs * i + d] = (own && d == 0) ? static_cast<double>(rot_ok[a]) : 0.0;
    }
}
// offsets of all local nodes; a ghost node takes the centroid it received and hands it on to its (ghost) aggregate
__global__ void __launch_bounds__(kBlock) k_rbm_offsets_coupled(int64_t nn, int64_t nn_own, int bs, int dim, const int32_t *__restrict__ node_agg,
                                                                 const double *__restrict__ xyz, int64_t nn_c, double *__restrict__ cen,
                                                                 const int32_t *__restrict__ rot_ok, const double *__restrict__ v,
                                                                 const double *__restrict__ w, double undo, double *__restrict__ roff)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= nn) return;
    const int32_t a = node_agg[i];
    if (i < nn_own) {
        const bool ok = rot_ok[a] != 0;
        for (int d = 0; d < 3; ++d) roff[d * nn + i] = ok ? xyz[d * nn + i] - cen[d * nn_c + a] : 0.0;
        return;
    }
    const bool ok = undo * w[bs * i] > 0.5;
    for (int d = 0; d < 3; ++d) {
        const double c = d < dim ? undo * v[bs * i + d] : 0.0;
        cen[d * nn_c + a] = c;                        // (every ghost member of the aggregate writes the same value)
        roff[d * nn + i] = ok ? xyz[d * nn + i] - c : 0.0;
    }
}
// strength graph of the pairing on several ranks: the owned nodes' rows of the node graph without their ghost columns
__global__ void __launch_bounds__(kBlock) k_rbm_owned_degree(int64_t nn_own, const int64_t *__restrict__ gptr, const int32_t *__restrict__ gcol,
                                                              int64_t *__restrict__ deg)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i > nn_own) return;
    int64_t n = 0;
    if (i < nn_own)
        for (int64_t q = gptr[i]; q < gptr[i + 1]; ++q) n += gcol[q] < nn_own;
    deg[i] = n;
}
__global__ void __launch_bounds__(kBlock) k_rbm_owned_graph(int64_t nn_own, const int64_t *__restrict__ gptr, const int32_t *__restrict__ gcol,
                                                             const double *__restrict__ gw, const int64_t *__restrict__ optr,
                                                             int32_t *__restrict__ ocol, double *__restrict__ ow)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= nn_own) return;
    int64_t o = optr[i];
    for (int64_t q = gptr[i]; q < gptr[i + 1]; ++q)
        if (gcol[q] < nn_own) { ocol[o] = gcol[q]; ow[o] = gw[q]; ++o; }
}
// owned coarse nodes' centroids by global node number into the array every rank gets through an all-reduce (zeroed before)
__global__ void __launch_bounds__(kBlock) k_rbm_emit_global_cen(int64_t na_own, int64_t node_off, int64_t nn_c, const double *__restrict__ cen,
                                                                 int64_t n_glob, double *__restrict__ out)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= na_own) return;
    for (int d = 0; d < 3; ++d) out[d * n_glob + node_off + i] = cen[d * nn_c + i];
}
// |.| row sums of the symmetrically scaled operator from the ranks' shares (summed over the holders afterwards), block maximum
__global__ void __launch_bounds__(kBlock) k_rbm_block_max(int64_t n, const double *__restrict__ v, double *__restrict__ part_max)
{
    __shared__ double sm[4];
    const int64_t r = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    double s = r < n ? v[r] : 0.0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s = fmax(s, __shfl_xor(s, o, 64));
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part_max[blockIdx.x] = fmax(fmax(sm[0], sm[1]), fmax(sm[2], sm[3]));
}

// coarse node block of every fine node block: key (agg(i) << 32 | agg(j)), payload the block's index (sorted afterwards;
// the radix sort is stable, so the blocks of one coarse block stay in ascending order: one fixed sum)
__global__ void __launch_bounds__(kBlock) k_rbm_emit_block_keys(int64_t nblk, const int32_t *__restrict__ brow, const int32_t *__restrict__ gcol,
                                                                 const int32_t *__restrict__ node_agg, uint64_t *__restrict__ keys,
                                                                 int32_t *__restrict__ idx)
{
    const int64_t q = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (q >= nblk) return;
    keys[q] = (static_cast<uint64_t>(static_cast<uint32_t>(node_agg[brow[q]])) << 32) | static_cast<uint32_t>(node_agg[gcol[q]]);
    idx[q] = static_cast<int32_t>(q);
}
__global__ void __launch_bounds__(kBlock) k_rbm_split_keys(int64_t n, const uint64_t *__restrict__ ukeys, int32_t *__restrict__ brow,
                                                            int32_t *__restrict__ gcol)
{
    const int64_t q = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (q >= n) return;
    brow[q] = static_cast<int32_t>(ukeys[q] >> 32);
    gcol[q] = static_cast<int32_t>(ukeys[q] & 0xffffffffu);
}
// scalar row pointer of a block-regular level from its node graph
__global__ void __launch_bounds__(kBlock) k_rbm_rowptr(int64_t n_nodes, int bs, const int64_t *__restrict__ gptr, int64_t *__restrict__ rowptr)
{
    const int64_t r = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    const int64_t n = n_nodes * bs;
    if (r > n) return;
    if (r == n) { rowptr[r] = gptr[n_nodes] * bs * bs; return; }
    const int64_t i = r / bs;
    const int a = static_cast<int>(r % bs);
    rowptr[r] = gptr[i] * bs * bs + static_cast<int64_t>(a) * bs * (gptr[i + 1] - gptr[i]);
}
// its columns in the wave-sliced storage (padding: the own row, as everywhere)
__global__ void __launch_bounds__(kBlock) k_rbm_fill_cols(int64_t n_rows, int64_t n_slices, int bs, const int64_t *__restrict__ slice_off,
                                                           const int64_t *__restrict__ gptr, const int32_t *__restrict__ gcol,
                                                           int32_t *__restrict__ cols)
{
    const int64_t r = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    const int64_t s = r >> 6;
    if (s >= n_slices) return;
    const int64_t off = slice_off[s] + (r & 63);
    const int width = static_cast<int>((slice_off[s + 1] - slice_off[s]) >> 6);
    int64_t q0 = 0;
    int len = 0;
    if (r < n_rows) { q0 = gptr[r / bs]; len = static_cast<int>(gptr[r / bs + 1] - q0) * bs; }
    const int32_t pad = r < n_rows ? static_cast<int32_t>(r) : 0;
    for (int k = 0; k < width; ++k) cols[off + 64LL * k] = k < len ? bs * gcol[q0 + k / bs] + k % bs : pad;
}

// ---- numeric phase: Galerkin product with the rigid-body prolongator, one thread per coarse node block ------------------
// C(I,J) = sum over the fine blocks (i,j), i in I, j in J, of  P_i^T F_ij P_j.  A fine block is read row by row:
// row m of G = F P_j is formed from row m of F, then added to the rows of C that row m of P_i^T feeds.
constexpr int kRbmLanes = 8;          // lanes that share one coarse block (its fine blocks dealt out round-robin, sums combined by a fixed
                                      // butterfly): a diagonal block sums 64 fine blocks, a corner block one -- with one lane per coarse
                                      // block a wave waited for its longest list (k_rbm_galerkin<3,3> 1.29 ms at config 4)
template <int FB, int DIM>
__global__ void __launch_bounds__(kBlock) k_rbm_galerkin(int64_t nblk_c, const int64_t *__restrict__ src_ptr, const int32_t *__restrict__ src_blk,
                                                          SellDev F, const int64_t *__restrict__ f_gptr, const int32_t *__restrict__ f_gcol,
                                                          const int32_t *__restrict__ f_brow, const double *__restrict__ roff, int64_t nn,
                                                          SellDev C, const int64_t *__restrict__ c_gptr, const int32_t *__restrict__ c_brow,
                                                          const int32_t *__restrict__ c_gcol, int64_t c_own_nodes)
{
    constexpr int NR = RbmDims<DIM>::NR, CB = RbmDims<DIM>::CB;
    const int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    const int64_t c = t / kRbmLanes;
    const int g = static_cast<int>(t % kRbmLanes);
    const bool live = c < nblk_c;
    double acc[CB][CB];
#pragma unroll
    for (int a = 0; a < CB; ++a)
#pragma unroll
        for (int b = 0; b < CB; ++b) acc[a][b] = 0.0;
    const int64_t p1 = live ? src_ptr[c + 1] : 0;
    for (int64_t p = live ? src_ptr[c] + g : 0; p < p1; p += kRbmLanes) {
        const int64_t q = src_blk[p];
        const int64_t i = f_brow[q];
        const int64_t j = f_gcol[q];
        const int jj = static_cast<int>(q - f_gptr[i]);
        double ri[3], rj[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) { ri[d] = roff[d * nn + i]; rj[d] = roff[d * nn + j]; }
        double f[FB][FB];
#pragma unroll
        for (int m = 0; m < FB; ++m) {
            const int64_t base = rbm_row_base(F, FB * i + m) + 64LL * (FB * jj);
#pragma unroll
            for (int d = 0; d < FB; ++d) f[m][d] = F.vals[base + 64LL * d];
        }
#pragma unroll
        for (int m = 0; m < FB; ++m) {
            double gr[CB];
            // gr = (row m of F) P_j :  translations as they are, rotations  f . (e_k x r_j) = (r_j x f)_k  [+ the fine rotations]
#pragma unroll
            for (int d = 0; d < DIM; ++d) gr[d] = f[m][d];
            {
                double mom[NR];
#pragma unroll
                for (int k = 0; k < NR; ++k) mom[k] = 0.0;
                rbm_moment<DIM>(rj, f[m], mom);
#pragma unroll
                for (int k = 0; k < NR; ++k) gr[DIM + k] = FB > DIM ? mom[k] + f[m][DIM + k] : mom[k];
            }
            // acc += (row m of P_i)^T gr
            if (m < DIM) {
#pragma unroll
                for (int b = 0; b < CB; ++b) acc[m][b] += gr[b];
                if (DIM == 3) {
                    // P_i[m, DIM + k] = (e_k x r_i)_m:  k = m+1 -> +r[m+2],  k = m+2 -> -r[m+1]  (indices mod 3)
                    const int m1 = (m + 1) % 3, m2 = (m + 2) % 3;
#pragma unroll
                    for (int b = 0; b < CB; ++b) {
                        acc[DIM + m1][b] += ri[m2] * gr[b];
                        acc[DIM + m2][b] -= ri[m1] * gr[b];
                    }
                } else {
                    const double w = m == 0 ? -ri[1] : ri[0];
#pragma unroll
                    for (int b = 0; b < CB; ++b) acc[DIM][b] += w * gr[b];
                }
            } else {
#pragma unroll
                for (int b = 0; b < CB; ++b) acc[m][b] += gr[b];
            }
        }
    }
    // the lanes of a coarse block combine their sums (fixed butterfly: the same bits in every run)
#pragma unroll
    for (int a = 0; a < CB; ++a)
#pragma unroll
        for (int b = 0; b < CB; ++b) {
            double v = acc[a][b];
#pragma unroll
            for (int o = kRbmLanes / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            acc[a][b] = v;
        }
    if (!live) return;
    const int64_t I = c_brow[c];
    const int jc = static_cast<int>(c - c_gptr[I]);
    const bool diag_block = c_gcol[c] == I;
#pragma unroll
    for (int a = 0; a < CB; ++a) {
        if (a % kRbmLanes != g) continue;                   // lane g writes row g of the block
        const int64_t base = rbm_row_base(C, CB * I + a) + 64LL * (CB * jc);
#pragma unroll
        for (int b = 0; b < CB; ++b) {
            double v = acc[a][b];
            // rotation of a degenerate aggregate: an idle dof with a unit diagonal (several ranks: the owner's share carries it)
            if (FB == DIM && diag_block && a == b && a >= DIM && v == 0.0 && I < c_own_nodes) v = 1.0;
            C.vals[base + 64LL * b] = v;
        }
    }
}

// ---- cycle: the coarse-level SpMV + vector step (k_amg_spmv_ep) on a block-regular level.  The column of entry CB*kk + d of
// a row is CB * (kk-th neighbour of its node) + d: one node index per CB x CB values instead of one column per value
// (8.1 instead of 12 bytes per nonzero), and the CB values of x it multiplies are contiguous.  Same products in the same
// order as k_amg_spmv_ep / k_spmv: same bits.
template <int MODE, int CB>
__global__ void __launch_bounds__(kBlock) k_rbm_spmv_ep(SellDev A, const int64_t *__restrict__ gptr, const int32_t *__restrict__ gcol,
                                                         const double *__restrict__ xin, const double *r_in, const double *__restrict__ dinv,
                                                         const double *__restrict__ lam, double ratio, int step, int add_dd0, double *r_out,
                                                         double *dd_out, double *x, const CgCtl *ctl)
{
    if (ctl && ctl->flag != 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t s = (static_cast<int64_t>(blockIdx.x) << 2) + wave;
    if (s >= A.n_slices) return;
    const int64_t off = A.slice_off[s];
    const int nb = static_cast<int>((A.slice_off[s + 1] - off) >> 6) / CB;          // node blocks in the longest row of the slice
    const int64_t i = (s << 6) + lane;
    const bool live = i < A.n_rows;
    const int64_t node = live ? i / CB : 0;
    const int64_t q0 = live ? gptr[node] : 0;
    const int deg = live ? static_cast<int>(gptr[node + 1] - q0) : 0;
    const double *__restrict__ vp = A.vals + off + lane;
    const int32_t own = static_cast<int32_t>(node * CB);
    double acc = 0.0;
    int kk = 0;
    for (; kk + 2 <= nb; kk += 2) {
        const int32_t c0 = kk < deg ? CB * gcol[q0 + kk] : own;           // (beyond the row's end the values are the padding zeros)
        const int32_t c1 = kk + 1 < deg ? CB * gcol[q0 + kk + 1] : own;
        double v[2 * CB], xv[2 * CB];
#pragma unroll
        for (int d = 0; d < 2 * CB; ++d) v[d] = vp[64 * (CB * kk + d)];
#pragma unroll
        for (int d = 0; d < CB; ++d) { xv[d] = xin[c0 + d]; xv[CB + d] = xin[c1 + d]; }
#pragma unroll
        for (int d = 0; d < 2 * CB; ++d) acc = __builtin_fma(v[d], xv[d], acc);
    }
    for (; kk < nb; ++kk) {
        const int32_t c0 = kk < deg ? CB * gcol[q0 + kk] : own;
#pragma unroll
        for (int d = 0; d < CB; ++d) acc = __builtin_fma(vp[64 * (CB * kk + d)], xin[c0 + d], acc);
    }
    if (!live) return;
    if (MODE == kEpResid) {
        r_out[i] = r_in[i] - acc;
    } else if (MODE == kEpFirstRes) {
        const double ri = r_in[i] - acc;
        r_out[i] = ri;
        dd_out[i] = cheb_coef(lam[0], ratio, 0).c_first * dinv[i] * ri;
    } else if (MODE == kEpNextLast) {
        const ChebCoef c = cheb_coef(lam[0], ratio, step);
        const double ri = r_in[i] - acc;
        const double d0 = xin[i];
        const double di = __builtin_fma(c.c_dd, d0, c.c_r * dinv[i] * ri);
        double xv = x[i];
        if (add_dd0) xv += d0;
        x[i] = xv + di;
    } else {                  // plain product
        r_out[i] = acc;
    }
}

// second bound of lambda_max(D^-1 A): Gershgorin on the symmetrically scaled matrix, max_i sum_j |a_ij| sqrt(dinv_i dinv_j).
// Unlike the row sums of D^-1 A it does not depend on the scaling of the basis -- the rotation dofs of a coarse node carry a
// length^2 against the translations, which makes the plain bound several times too large there (35 against 3.4 on the
// second level of the beam) and the Chebyshev interval useless.  Both are bounds; the smaller one is taken.
__global__ void __launch_bounds__(kBlock) k_rbm_bound_sym(SellDev A, int32_t col_limit, const double *__restrict__ dinv, double *__restrict__ rowsum,
                                                           double *__restrict__ part_max)
{
    __shared__ double sm[4];
    const int64_t r = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    double s = 0.0;
    if (r < A.n_rows) {
        const int64_t base = rbm_row_base(A, r);
        const int len = A.rowlen[r];
        for (int k = 0; k < len; ++k) {
            const int32_t c = A.cols[base + 64LL * k];
            if (c < col_limit) s += fabs(A.vals[base + 64LL * k]) * sqrt(dinv[c]);
        }
        s *= sqrt(dinv[r]);
        if (rowsum) rowsum[r] = s;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s = fmax(s, __shfl_xor(s, o, 64));
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part_max[blockIdx.x] = fmax(fmax(sm[0], sm[1]), fmax(sm[2], sm[3]));
}
__global__ void k_rbm_min2(const double *__restrict__ a, double *__restrict__ lam)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) lam[0] = fmin(lam[0], a[0]);
}

// ---- cycle: restriction and prolongation with the rigid-body prolongator ----------------------------------------------
// one coarse node: bc = sum over its member nodes of P_i^T res_i, members in ascending order
template <int FB, int DIM>
__device__ __forceinline__ void rbm_restrict_node(int64_t a, const int32_t *__restrict__ mem_ptr, const int32_t *__restrict__ mem_idx,
                                                  const double *__restrict__ roff, int64_t nn, const double *__restrict__ b, const double *__restrict__ t,
                                                  int64_t n_own_nodes, double *out)
{
    constexpr int NR = RbmDims<DIM>::NR, CB = RbmDims<DIM>::CB;
#pragma unroll
    for (int k = 0; k < CB; ++k) out[k] = 0.0;
    const int q0 = mem_ptr[a], q1 = mem_ptr[a + 1];
    for (int q = q0; q < q1; ++q) {
        const int64_t i = mem_idx[q];
        double f[FB], r[3];
#pragma unroll
        for (int d = 0; d < FB; ++d) {
            const double bv = i < n_own_nodes ? b[FB * i + d] : 0.0;          // (coupled hierarchy: b counts where the rank owns the node)
            f[d] = t ? bv - t[FB * i + d] : bv;
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) r[d] = roff[d * nn + i];
#pragma unroll
        for (int d = 0; d < DIM; ++d) out[d] += f[d];
        double mom[NR];
#pragma unroll
        for (int k = 0; k < NR; ++k) mom[k] = 0.0;
        rbm_moment<DIM>(r, f, mom);
#pragma unroll
        for (int k = 0; k < NR; ++k) out[DIM + k] += FB > DIM ? mom[k] + f[DIM + k] : mom[k];
    }
}
template <int FB, int DIM>
__global__ void __launch_bounds__(kBlock) k_rbm_restrict(int64_t nc_nodes, const int32_t *__restrict__ mem_ptr, const int32_t *__restrict__ mem_idx,
                                                          const double *__restrict__ roff, int64_t nn, const double *__restrict__ b,
                                                          const double *__restrict__ t, double *__restrict__ bc, const double *__restrict__ dinv_c,
                                                          const double *__restrict__ lam_c, double ratio, double *__restrict__ dd_c,
                                                          double *__restrict__ x_c, const CgCtl *ctl, int64_t n_own_nodes)
{
    constexpr int CB = RbmDims<DIM>::CB;
    if (ctl && ctl->flag != 0) return;
    const int64_t a = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (a >= nc_nodes) return;
    double out[CB];
    rbm_restrict_node<FB, DIM>(a, mem_ptr, mem_idx, roff, nn, b, t, n_own_nodes, out);
    const double c_first = dd_c ? cheb_coef(lam_c[0], ratio, 0).c_first : 0.0;
#pragma unroll
    for (int k = 0; k < CB; ++k) {
        bc[CB * a + k] = out[k];
        if (dd_c) {          // step 0 of the next level's pre-smoothing (zero guess), as k_amg_restrict does it
            const double di = c_first * dinv_c[CB * a + k] * out[k];
            dd_c[CB * a + k] = di;
            x_c[CB * a + k] = di;
        }
    }
}
// The same restriction for LARGE aggregates (level 0 of a displacement problem: bricks of 4x4x4 = 64 nodes): kRbmWide lanes share
// one coarse node.  With a thread per coarse node the 12 675 threads of config 4's level 1 walked 64 members each, one dependent
// round trip to memory after the other: 97 us per cycle for 37 MB, a sixth of the iteration (profiles/r06/
// rocprofv3_kernel_stats_beam.txt).  Lane l takes members l, l + kRbmWide, ... in ascending order, the lanes' sums are combined
// by a fixed butterfly: deterministic, the same on every run and in every caller (the association differs from the one-thread
// form's: a level uses one form or the other throughout -- amg_rbm_restrict_launch decides from the level's sizes alone).
constexpr int kRbmWide = 16;
template <int FB, int DIM>
__global__ void __launch_bounds__(kBlock) k_rbm_restrict_wide(int64_t nc_nodes, const int32_t *__restrict__ mem_ptr, const int32_t *__restrict__ mem_idx,
                                                               const double *__restrict__ roff, int64_t nn, const double *__restrict__ b,
                                                               const double *__restrict__ t, double *__restrict__ bc, const double *__restrict__ dinv_c,
                                                               const double *__restrict__ lam_c, double ratio, double *__restrict__ dd_c,
                                                               double *__restrict__ x_c, const CgCtl *ctl, int64_t n_own_nodes)
{
    constexpr int NR = RbmDims<DIM>::NR, CB = RbmDims<DIM>::CB;
    if (ctl && ctl->flag != 0) return;
    const int64_t gt = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    const int64_t a = gt / kRbmWide;
    const int lane = static_cast<int>(gt % kRbmWide);
    const bool live = a < nc_nodes;
    double out[CB];
#pragma unroll
    for (int k = 0; k < CB; ++k) out[k] = 0.0;
    if (live) {
        const int q0 = mem_ptr[a], q1 = mem_ptr[a + 1];
        for (int q = q0 + lane; q < q1; q += kRbmWide) {
            const int64_t i = mem_idx[q];
            double f[FB], r[3];
#pragma unroll
            for (int d = 0; d < FB; ++d) {
                const double bv = i < n_own_nodes ? b[FB * i + d] : 0.0;
                f[d] = t ? bv - t[FB * i + d] : bv;
            }
#pragma unroll
            for (int d = 0; d < 3; ++d) r[d] = roff[d * nn + i];
#pragma unroll
            for (int d = 0; d < DIM; ++d) out[d] += f[d];
            double mom[NR];
#pragma unroll
            for (int k = 0; k < NR; ++k) mom[k] = 0.0;
            rbm_moment<DIM>(r, f, mom);
#pragma unroll
            for (int k = 0; k < NR; ++k) out[DIM + k] += FB > DIM ? mom[k] + f[DIM + k] : mom[k];
        }
    }
#pragma unroll
    for (int o = kRbmWide / 2; o > 0; o >>= 1)
#pragma unroll
        for (int k = 0; k < CB; ++k) out[k] += __shfl_xor(out[k], o, 64);
    if (!live || lane != 0) return;
    const double c_first = dd_c ? cheb_coef(lam_c[0], ratio, 0).c_first : 0.0;
#pragma unroll
    for (int k = 0; k < CB; ++k) {
        bc[CB * a + k] = out[k];
        if (dd_c) {
            const double di = c_first * dinv_c[CB * a + k] * out[k];
            dd_c[CB * a + k] = di;
            x_c[CB * a + k] = di;
        }
    }
}
// x_i += scale * P_i xc_I, one fine node
template <int FB, int DIM>
__device__ __forceinline__ void rbm_prolong_node(int64_t i, const int32_t *__restrict__ node_agg, const double *__restrict__ roff, int64_t nn,
                                                 const double *__restrict__ xc, double scale, double *__restrict__ x)
{
    constexpr int NR = RbmDims<DIM>::NR, CB = RbmDims<DIM>::CB;
    const int64_t a = node_agg[i];
    double r[3], w[NR], u[DIM];
#pragma unroll
    for (int d = 0; d < 3; ++d) r[d] = roff[d * nn + i];
#pragma unroll
    for (int k = 0; k < NR; ++k) w[k] = xc[CB * a + DIM + k];
    rbm_spin<DIM>(r, w, u);
#pragma unroll
    for (int d = 0; d < DIM; ++d) x[FB * i + d] = __builtin_fma(scale, xc[CB * a + d] + u[d], x[FB * i + d]);
    if (FB > DIM) {
#pragma unroll
        for (int k = 0; k < NR; ++k) x[FB * i + DIM + k] = __builtin_fma(scale, w[k], x[FB * i + DIM + k]);
    }
}
template <int FB, int DIM>
__global__ void __launch_bounds__(kBlock) k_rbm_prolong(int64_t nn_active, const int32_t *__restrict__ node_agg, const double *__restrict__ roff,
                                                         int64_t nn, const double *__restrict__ xc, double scale, double *__restrict__ x,
                                                         const CgCtl *ctl)
{
    if (ctl && ctl->flag != 0) return;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < nn_active; i += static_cast<int64_t>(gridDim.x) * kBlock)
        rbm_prolong_node<FB, DIM>(i, node_agg, roff, nn, xc, scale, x);
}

}  // namespace pfem
