// pfem_amg_kernels.hpp -- gfx950 kernels of the aggregation-multigrid preconditioner (-pc_type gamg; pfem_amg.inc holds the
// host side and the design notes).  Included by pfem_kernels.hpp's users after it; everything lives in namespace pfem.
#pragma once

namespace pfem {

// ---------------------------------------------------------------------------
// symbolic phase: strength graph, pairwise matching, aggregates, Galerkin maps
// ---------------------------------------------------------------------------
// Symmetric priority of the edge {i,j} of the strength graph: a node proposes to its best free neighbour and two nodes
// that propose to each other are matched (locally dominant edges: the maximal edge is always mutual, so every round
// makes progress).  Priority, most significant first: (1) the coupling strength s = -w_ij / sqrt(w_ii w_jj) in eighths
// of an octave -- couplings that differ by rounding only tie; (2) index distance, closer first: on a mesh numbered
// line by line this makes ties pair along the numbering direction, so three passes build 2x2x2 bricks instead of random
// octets; (3) parity of the lower index: of the two neighbours of a node on a line exactly one forms an "even" edge, so
// the whole line pairs up in ONE round; (4) a hash of the pair.  Positive couplings (w >= 0) are not eligible.
__device__ __forceinline__ int64_t amg_edge_key(int32_t i, int32_t j, int32_t hi_, int32_t hj_, double w, double di, double dj, int flat)
{
    if (!(w < 0.0) || !(di > 0.0) || !(dj > 0.0)) return -1;
    const double s = -w / sqrt(di * dj);
    int b = static_cast<int>(floor(8.0 * log2(s))) + 2048;
    b = b < 1 ? 1 : (b > 4095 ? 4095 : b);
    if (flat) b = 1;                     // lab knob PFEM_AMG_NO_STRENGTH: every negative coupling equally strong
    // (2), (3) on the nodes' PLACE ALONG A SPACE-FILLING CURVE when the mesh came with coordinates (hi_, hj_: Morton ranks,
    // halved from pass to pass and from level to level), else on their indices
    const int64_t lo = hi_ < hj_ ? hi_ : hj_, hi = hi_ < hj_ ? hj_ : hi_;
    const int64_t close = 0x7fffffffLL - (hi - lo);
    const int64_t par = (lo & 1) == 0;
    const int64_t a = i < j ? i : j, c = i < j ? j : i;
    const int64_t h = ((a * 2654435761LL + c * 40503LL) >> 7) & 0x7ffffLL;
    return (static_cast<int64_t>(b) << 51) | (close << 20) | (par << 19) | h;
}

// every free node proposes to its best free neighbour
__global__ void __launch_bounds__(kBlock) k_amg_match_pick(int64_t n, const int64_t *__restrict__ gptr, const int32_t *__restrict__ gcol,
                                                            const double *__restrict__ gw, const double *__restrict__ gdiag,
                                                            const int32_t *__restrict__ hint /* null: the index */, int flat,
                                                            const int32_t *__restrict__ match, int32_t *__restrict__ cand)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= n) return;
    int32_t best = -1;
    if (match[i] < 0) {
        int64_t bk = -1;
        const double di = gdiag[i];
        const int32_t hi_ = hint ? hint[i] : static_cast<int32_t>(i);
        for (int64_t q = gptr[i]; q < gptr[i + 1]; ++q) {
            const int32_t j = gcol[q];
            if (j == i || match[j] >= 0) continue;
            const int64_t k = amg_edge_key(static_cast<int32_t>(i), j, hi_, hint ? hint[j] : j, gw[q], di, gdiag[j], flat);
            if (k > bk) { bk = k; best = j; }
        }
    }
    cand[i] = best;
}

// mutual proposals become pairs (root = the lower index)
__global__ void __launch_bounds__(kBlock) k_amg_match_commit(int64_t n, const int32_t *__restrict__ cand, int32_t *__restrict__ match)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= n) return;
    const int32_t j = cand[i];
    if (j >= 0 && cand[j] == static_cast<int32_t>(i)) match[i] = j < i ? j : static_cast<int32_t>(i);
}

// nodes left over stay alone; flag the roots
__global__ void __launch_bounds__(kBlock) k_amg_match_finish(int64_t n, int32_t *__restrict__ match, int32_t *__restrict__ is_root)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= n) return;
    if (match[i] < 0) match[i] = static_cast<int32_t>(i);
    is_root[i] = match[i] == static_cast<int32_t>(i);
}

// aggregate id of a node = rank of its root among the roots; composed with the map of the earlier passes
__global__ void __launch_bounds__(kBlock) k_amg_agg_ids(int64_t n, const int32_t *__restrict__ match, const int32_t *__restrict__ root_rank,
                                                         int32_t *__restrict__ agg)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i < n) agg[i] = root_rank[match[i]];
}
__global__ void __launch_bounds__(kBlock) k_amg_compose(int64_t n, int32_t *__restrict__ total, const int32_t *__restrict__ step)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i < n) total[i] = step[total[i]];
}
// place of an aggregate along the curve = the lowest place of its members, halved (a pair of curve neighbours 2k, 2k+1 -> k)
__global__ void __launch_bounds__(kBlock) k_amg_hint_coarsen(int64_t n, const int32_t *__restrict__ hint, const int32_t *__restrict__ agg,
                                                              int32_t *__restrict__ hint_c)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i < n) atomicMin(&hint_c[agg[i]], (hint ? hint[i] : static_cast<int32_t>(i)) >> 1);
}
// ---- roots + neighbours: aggregates from an independent set of the strength graph (round 6) ---------------------------------
// WHY.  Where a displacement problem's mesh has no lattice, five passes of pairwise matching gave the rigid-body transfer lopsided
// aggregates (chains of pairs of pairs): the beam with its nodes moved off the lattice took 61 iterations where 4x4x4 bricks take
// 22, and every pass cost a sorted aggregate graph.  The classical remedy (Vanek, Mandel, Brezina 1996: aggregation by roots)
// gives compact ones in a few sweeps over the graph and no sort: a ROOT is a node whose priority is the highest within TWO
// strong hops among the undecided (so roots end up at least three hops apart), its strong neighbours join it at once, nodes two
// hops from a root wait; rounds repeat until nobody is undecided; then the waiting nodes join the aggregate of their strongest
// decided neighbour (two sweeps), and whoever is still alone becomes a root.  j is a strong neighbour of i when
// s_ij >= 1/4 max_k s_ik (the threshold of the pairing).  Priorities are a hash of the node's place along the curve / its index,
// unique by construction: the outcome depends on nothing but the graph.  state: 0 undecided, 1 root, 2 member, 3 waiting.
__device__ __forceinline__ double amg_strength(double w, double di, double dj);
__device__ __forceinline__ unsigned long long mis_prio(int32_t place, int64_t i)
{
    unsigned long long h = static_cast<unsigned long long>(static_cast<uint32_t>(place)) * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29;
    h *= 0xBF58476D1CE4E5B9ull;
    h ^= h >> 32;
    return ((h & 0x7fffffffull) << 32) | static_cast<unsigned long long>(i + 1);
}
__global__ void __launch_bounds__(kBlock) k_mis_init(int64_t n, const int64_t *__restrict__ gptr, const int32_t *__restrict__ gcol,
                                                      const double *__restrict__ gw, const double *__restrict__ gdiag, const int32_t *__restrict__ hint,
                                                      double *__restrict__ smax, unsigned long long *__restrict__ prio, int32_t *__restrict__ state,
                                                      int32_t *__restrict__ match)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= n) return;
    const double di = gdiag[i];
    double m = 0.0;
    for (int64_t q = gptr[i]; q < gptr[i + 1]; ++q) {
        const int32_t j = gcol[q];
        if (j != i) m = fmax(m, amg_strength(gw[q], di, gdiag[j]));
    }
    smax[i] = m;
    prio[i] = mis_prio(hint ? hint[i] : static_cast<int32_t>(i), i);
    state[i] = 0;
    match[i] = -1;
}
// out[i] = max over i and its strong neighbours of (FIRST: the priority of the undecided / else: in[j])
template <bool FIRST>
__global__ void __launch_bounds__(kBlock) k_mis_spread(int64_t n, const int64_t *__restrict__ gptr, const int32_t *__restrict__ gcol,
                                                        const double *__restrict__ gw, const double *__restrict__ gdiag, const double *__restrict__ smax,
                                                        const int32_t *__restrict__ state, const unsigned long long *__restrict__ in,
                                                        unsigned long long *__restrict__ out)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= n) return;
    const double di = gdiag[i], thr = 0.25 * smax[i];
    unsigned long long m = FIRST ? (state[i] == 0 ? in[i] : 0ull) : in[i];
    for (int64_t q = gptr[i]; q < gptr[i + 1]; ++q) {
        const int32_t j = gcol[q];
        if (j == i) continue;
        const double sij = amg_strength(gw[q], di, gdiag[j]);
        if (!(sij > 0.0 && sij >= thr)) continue;
        const unsigned long long v = FIRST ? (state[j] == 0 ? in[j] : 0ull) : in[j];
        m = v > m ? v : m;
    }
    out[i] = m;
}
__global__ void __launch_bounds__(kBlock) k_mis_roots(int64_t n, const unsigned long long *__restrict__ prio, const unsigned long long *__restrict__ t2,
                                                       int32_t *__restrict__ state, int32_t *__restrict__ match)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i < n && state[i] == 0 && prio[i] == t2[i]) { state[i] = 1; match[i] = static_cast<int32_t>(i); }
}
// an undecided node next to a root joins it (the strongest such root, the lower index on a tie); cand = the root, else -1
__global__ void __launch_bounds__(kBlock) k_mis_join_pick(int64_t n, const int64_t *__restrict__ gptr, const int32_t *__restrict__ gcol,
                                                           const double *__restrict__ gw, const double *__restrict__ gdiag, const double *__restrict__ smax,
                                                           const int32_t *__restrict__ state, int32_t *__restrict__ cand)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= n) return;
    int32_t best = -1;
    if (state[i] == 0) {
        const double di = gdiag[i], thr = 0.25 * smax[i];
        double sb = 0.0;
        for (int64_t q = gptr[i]; q < gptr[i + 1]; ++q) {
            const int32_t j = gcol[q];
            if (j == i || state[j] != 1) continue;
            const double sij = amg_strength(gw[q], di, gdiag[j]);
            if (sij > 0.0 && sij >= thr && (sij > sb || (sij == sb && j < best))) { sb = sij; best = j; }
        }
    }
    cand[i] = best;
}
__global__ void __launch_bounds__(kBlock) k_mis_join_commit(int64_t n, const int32_t *__restrict__ cand, int32_t *__restrict__ state, int32_t *__restrict__ match)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i < n && cand[i] >= 0) { state[i] = 2; match[i] = cand[i]; }
}
// an undecided node next to a MEMBER is two hops from a root: it waits (state 3); *left counts the nodes still undecided
__global__ void __launch_bounds__(kBlock) k_mis_wait(int64_t n, const int64_t *__restrict__ gptr, const int32_t *__restrict__ gcol,
                                                      const double *__restrict__ gw, const double *__restrict__ gdiag, const double *__restrict__ smax,
                                                      const int32_t *__restrict__ state, int32_t *__restrict__ cand)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= n) return;
    int32_t w = 0;
    if (state[i] == 0) {
        const double di = gdiag[i], thr = 0.25 * smax[i];
        for (int64_t q = gptr[i]; q < gptr[i + 1] && !w; ++q) {
            const int32_t j = gcol[q];
            if (j == i || state[j] != 2) continue;
            const double sij = amg_strength(gw[q], di, gdiag[j]);
            if (sij > 0.0 && sij >= thr) w = 1;
        }
    }
    cand[i] = w;
}
__global__ void __launch_bounds__(kBlock) k_mis_wait_commit(int64_t n, const int32_t *__restrict__ cand, int32_t *__restrict__ state, int *__restrict__ left)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= n) return;
    if (cand[i]) state[i] = 3;
    if (left && state[i] == 0) atomicAdd(left, 1);
}
// the waiting (and any still undecided) nodes: the aggregate of the strongest decided neighbour (any negative coupling counts here:
// nobody stays alone for a threshold); cand = that aggregate's root, else -1
__global__ void __launch_bounds__(kBlock) k_mis_adopt_pick(int64_t n, const int64_t *__restrict__ gptr, const int32_t *__restrict__ gcol,
                                                            const double *__restrict__ gw, const double *__restrict__ gdiag,
                                                            const int32_t *__restrict__ state, const int32_t *__restrict__ match, int32_t *__restrict__ cand)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= n) return;
    int32_t best = -1;
    if (state[i] == 0 || state[i] == 3) {
        const double di = gdiag[i];
        double sb = 0.0;
        int32_t bj = -1;
        for (int64_t q = gptr[i]; q < gptr[i + 1]; ++q) {
            const int32_t j = gcol[q];
            if (j == i || (state[j] != 1 && state[j] != 2)) continue;
            const double sij = amg_strength(gw[q], di, gdiag[j]);
            if (sij > 0.0 && (sij > sb || (sij == sb && j < bj))) { sb = sij; bj = j; }
        }
        if (bj >= 0) best = match[bj];
    }
    cand[i] = best;
}
// ---- pairing on a tensor-product lattice -----------------------------------------------------------------------------
// When the nodes of the mesh sit on a lattice (every coordinate takes few distinct values: the generated boxes, the
// reference's tet10 / tet100 files) a node's hint is its lattice position, 10 bits per axis (x | y << 10 | z << 20), and a
// pass pairs along ONE axis: node and partner agree in the other two positions and differ in bit 0 of this one -- exact
// 2x2x2 bricks after three passes, whatever the number of nodes per line.  The node a line of odd length leaves over joins
// the pair next to it on the same line (a brick of 3 in that direction) instead of pairing across lines, which is what
// produced plates of half thickness along the boundary of every level whose lines are odd (199 -> 100 -> 50 -> 25 -> 13 -> 7)
// and cost config 3 a third of its iterations (128^3 and 256^3, whose lines are 2^k - 1, never had them).  A pair is made
// only across a coupling at least a quarter as strong as the node's strongest (else the node sits this pass out: the
// classical strength threshold, i.e. semi-coarsening where the operator is anisotropic).
__device__ __forceinline__ double amg_strength(double w, double di, double dj)
{
    return (w < 0.0 && di > 0.0 && dj > 0.0) ? -w / sqrt(di * dj) : 0.0;
}
__global__ void __launch_bounds__(kBlock) k_amg_lat_pick(int64_t n, const int64_t *__restrict__ gptr, const int32_t *__restrict__ gcol,
                                                          const double *__restrict__ gw, const double *__restrict__ gdiag,
                                                          const int32_t *__restrict__ pos, int shift, int32_t *__restrict__ cand)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= n) return;
    const int32_t want = pos[i] ^ (1 << shift);          // the sibling's position
    const double di = gdiag[i];
    double smax = 0.0, ssib = 0.0;
    int32_t sib = -1;
    for (int64_t q = gptr[i]; q < gptr[i + 1]; ++q) {
        const int32_t j = gcol[q];
        if (j == i) continue;
        const double sij = amg_strength(gw[q], di, gdiag[j]);
        smax = fmax(smax, sij);
        if (pos[j] == want) { sib = j; ssib = sij; }
    }
    cand[i] = (sib >= 0 && ssib > 0.0 && ssib >= 0.25 * smax) ? sib : -1;
}
// a node left over joins the pair of its line neighbour (same position but +-1 along the axis of the pass)
__global__ void __launch_bounds__(kBlock) k_amg_lat_absorb(int64_t n, const int64_t *__restrict__ gptr, const int32_t *__restrict__ gcol,
                                                            const double *__restrict__ gw, const double *__restrict__ gdiag,
                                                            const int32_t *__restrict__ pos, int shift, const int32_t *__restrict__ match,
                                                            int32_t *__restrict__ join)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= n) return;
    int32_t to = -1;
    if (match[i] < 0) {
        const int32_t field = 0x3ff << shift, pi = pos[i], ci = (pi >> shift) & 0x3ff;
        const double di = gdiag[i];
        double smax = 0.0, sbest = 0.0;
        for (int64_t q = gptr[i]; q < gptr[i + 1]; ++q) {
            const int32_t j = gcol[q];
            if (j == i) continue;
            const double sij = amg_strength(gw[q], di, gdiag[j]);
            smax = fmax(smax, sij);
            const int32_t pj = pos[j], cj = (pj >> shift) & 0x3ff;
            if (((pi ^ pj) & ~field) == 0 && (cj == ci + 1 || cj == ci - 1) && match[j] >= 0 && sij > sbest) { sbest = sij; to = match[j]; }
        }
        if (!(sbest > 0.0 && sbest >= 0.25 * smax)) to = -1;
    }
    join[i] = to;
}
__global__ void __launch_bounds__(kBlock) k_amg_lat_join(int64_t n, const int32_t *__restrict__ join, int32_t *__restrict__ match)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i < n && join[i] >= 0) match[i] = join[i];
}
// position of an aggregate = its root's, halved along the axis of the pass
__global__ void __launch_bounds__(kBlock) k_amg_lat_coarsen(int64_t n, const int32_t *__restrict__ pos, const int32_t *__restrict__ match,
                                                             const int32_t *__restrict__ agg, int shift, int32_t *__restrict__ pos_c)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= n || match[i] != static_cast<int32_t>(i)) return;
    const int32_t field = 0x3ff << shift, p = pos[i];
    pos_c[agg[i]] = (p & ~field) | ((((p >> shift) & 0x3ff) >> 1) << shift);
}
// ---- bricks in one step (scalar problems on a lattice whose couplings along the axes are all strong) ---------------------
// The three passes above end in exact bricks when every sibling pair is accepted: brick = position >> shift per axis.  This
// path forms them directly from the positions -- no aggregate graphs, no sort of them -- after ONE check on the level's own
// matrix that the passes would indeed accept every pair: the coupling to the sibling along each halved axis, where the
// sibling exists, is at least a quarter of the node's strongest (else the pass-by-pass pairing takes the level as before:
// anisotropic operators keep their semi-coarsening).  Couplings must stay inside neighbouring bricks (|offset| <= 1).
struct LatBricks {
    int shift[3];          // halvings of every axis on this level (the passes' axis sequence, simulated on the host)
    int nb[3];             // bricks along every axis of the box the flag / rank arrays span
    int lo[3];             // first brick of that box along every axis (one rank: 0; several ranks: the box of this rank's local nodes)
};
__device__ __forceinline__ int lat_linear(const LatBricks &B, int bx, int by, int bz)
{
    return (bx - B.lo[0]) + B.nb[0] * ((by - B.lo[1]) + B.nb[1] * (bz - B.lo[2]));
}
__device__ __forceinline__ int lat_brick_linear(const LatBricks &B, int32_t pos)
{
    const int bx = (pos & 0x3ff) >> B.shift[0], by = ((pos >> 10) & 0x3ff) >> B.shift[1], bz = ((pos >> 20) & 0x3ff) >> B.shift[2];
    return lat_linear(B, bx, by, bz);
}
// one thread per row: sibling couplings strong enough?  every coupling within reach of the 27 neighbouring bricks?  marks the row's brick
__global__ void __launch_bounds__(kBlock) k_lat_check(SellDev A, const int32_t *__restrict__ pos, const double *__restrict__ diag, LatBricks B,
                                                       int32_t *__restrict__ brick_flag, int *__restrict__ fail)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= A.n_rows) return;
    const int64_t base = A.slice_off[i >> 6] + (i & 63);
    const int len = A.rowlen[i];
    const int32_t pi = pos[i];
    const double di = diag[i];
    const int bix = (pi & 0x3ff) >> B.shift[0], biy = ((pi >> 10) & 0x3ff) >> B.shift[1], biz = ((pi >> 20) & 0x3ff) >> B.shift[2];
    double smax = 0.0, ssib[3] = {-1.0, -1.0, -1.0};
    bool far = false, partial = false;
    for (int k = 0; k < len; ++k) {
        const int32_t j = A.cols[base + 64LL * k];
        if (j == static_cast<int32_t>(i)) continue;
        const int32_t pj = pos[j];
        // (a dof at the node's OWN position: the other owners' parts of the same brick of the level above, on a level that every
        // rank holds whole after bricks were split between their owners -- amg_split_bricks.  It lands in the node's brick
        // whatever its coupling, and that coupling -- the strongest a sliver has -- is no yardstick for the siblings')
        if (pj == pi) { partial = true; continue; }
        const double sij = amg_strength(A.vals[base + 64LL * k], di, diag[j]);
        smax = fmax(smax, sij);
        const int x = pi ^ pj;            // (several dofs may sit at the sibling's position too: the strongest of them speaks for it)
        if (x == 1) ssib[0] = fmax(ssib[0], sij);
        else if (x == (1 << 10)) ssib[1] = fmax(ssib[1], sij);
        else if (x == (1 << 20)) ssib[2] = fmax(ssib[2], sij);
        const int dx = ((pj & 0x3ff) >> B.shift[0]) - bix, dy = (((pj >> 10) & 0x3ff) >> B.shift[1]) - biy, dz = (((pj >> 20) & 0x3ff) >> B.shift[2]) - biz;
        far = far || dx < -1 || dx > 1 || dy < -1 || dy > 1 || dz < -1 || dz > 1;
    }
    bool weak = false;
    for (int a = 0; a < 3; ++a)
        if (B.shift[a] > 0 && ssib[a] >= 0.0 && !(ssib[a] > 0.0 && ssib[a] >= 0.25 * smax)) weak = true;
    // (a PART of a brick -- somebody else sits at its position -- is a thin cell by construction, and thin cells couple weakly
    // across their thin side: that says nothing about the brick it completes together with the other parts)
    if (partial) weak = false;
    if (weak || far) {
        *fail = 1;
        if (weak) atomicAdd(fail + 1, 1);          // (how many rows, and why: PFEM_AMG_VERBOSE prints them; rows that refuse are few or all)
        if (far) atomicAdd(fail + 2, 1);
    }
    brick_flag[lat_brick_linear(B, pi)] = 1;
}
// The same question for bricks that may hold dofs of several owners (amg_split_bricks: a rank's aggregate = ITS part of a brick):
// only the owned rows [0, A.n_rows) and their owned columns count -- a sibling another rank owns is not in the aggregate --, and
// couplings may reach anywhere (the Galerkin maps of such a level come from the sorted keys, not from the 27 offset codes).
__global__ void __launch_bounds__(kBlock) k_lat_check_split(SellDev A, const int32_t *__restrict__ pos, const double *__restrict__ diag, LatBricks B,
                                                             int32_t *__restrict__ brick_flag, int *__restrict__ fail)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= A.n_rows) return;
    const int64_t base = A.slice_off[i >> 6] + (i & 63);
    const int len = A.rowlen[i];
    const int32_t pi = pos[i];
    const double di = diag[i];
    double smax = 0.0, ssib[3] = {-1.0, -1.0, -1.0};
    for (int k = 0; k < len; ++k) {
        const int32_t j = A.cols[base + 64LL * k];
        if (j == static_cast<int32_t>(i) || j >= A.n_rows) continue;
        const int32_t pj = pos[j];
        const double sij = amg_strength(A.vals[base + 64LL * k], di, diag[j]);
        smax = fmax(smax, sij);
        const int x = pi ^ pj;
        if (x == 1) ssib[0] = sij;
        else if (x == (1 << 10)) ssib[1] = sij;
        else if (x == (1 << 20)) ssib[2] = sij;
    }
    bool weak = false;
    for (int a = 0; a < 3; ++a)
        if (B.shift[a] > 0 && ssib[a] >= 0.0 && !(ssib[a] > 0.0 && ssib[a] >= 0.25 * smax)) weak = true;
    if (weak) {
        *fail = 1;
        atomicAdd(fail + 1, 1);
    }
    brick_flag[lat_brick_linear(B, pi)] = 1;
}
// aggregate of every node = rank of its brick among the occupied ones (ascending in z, y, x: coarse columns come out ascending
// by offset code); an aggregate's position on the coarser lattice = its brick's coordinates
__global__ void __launch_bounds__(kBlock) k_lat_assign(int64_t n, const int32_t *__restrict__ pos, LatBricks B, const int32_t *__restrict__ brick_rank,
                                                        int32_t *__restrict__ agg, int32_t *__restrict__ pos_c)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= n) return;
    const int32_t p = pos[i];
    const int bx = (p & 0x3ff) >> B.shift[0], by = ((p >> 10) & 0x3ff) >> B.shift[1], bz = ((p >> 20) & 0x3ff) >> B.shift[2];
    const int32_t a = brick_rank[lat_linear(B, bx, by, bz)];
    agg[i] = a;
    pos_c[a] = bx | (by << 10) | (bz << 20);                // (every member writes the same value)
}
// ---- bricks across ranks (one hierarchy over several ranks, amg_bricks_level with `coupled`) -------------------------------
// Positions are GLOBAL lattice positions, the same on every rank, padded so that the planes where the ownership of the
// nodes changes sit on multiples of the level's brick size: no brick holds nodes of two owners, and everything above works
// unchanged on a rank's local nodes (owned first, ghosts after them) once the ghosts' bricks are in the rank table too.
// box of the positions pos[0..n): out = {min x, min y, min z, max x, max y, max z} (preset to 1023.. / 0..)
constexpr int kLatBoxBlocks = 512;
__global__ void __launch_bounds__(kBlock) k_lat_bbox(int64_t n, const int32_t *__restrict__ pos, int *__restrict__ out)
{
    // (grid-stride, one set of atomics per wave of a small grid: 7.9 M positions through per-wave atomics on six words took 17 ms)
    int lo[3] = {1023, 1023, 1023}, hi[3] = {0, 0, 0};
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * kBlock) {
        const int32_t p = pos[i];
        for (int d = 0; d < 3; ++d) {
            const int v = (p >> (10 * d)) & 0x3ff;
            lo[d] = min(lo[d], v);
            hi[d] = max(hi[d], v);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        for (int d = 0; d < 3; ++d) {
            lo[d] = min(lo[d], __shfl_xor(lo[d], o, 64));
            hi[d] = max(hi[d], __shfl_xor(hi[d], o, 64));
        }
    if ((threadIdx.x & 63) == 0)
        for (int d = 0; d < 3; ++d) {
            atomicMin(&out[d], lo[d]);
            atomicMax(&out[3 + d], hi[d]);
        }
}
// per-axis renumbering of positions through a table [3][1024] (the padding that aligns the ownership planes)
__global__ void __launch_bounds__(kBlock) k_lat_remap(int64_t n, const int32_t *__restrict__ in, const int32_t *__restrict__ table,
                                                       int32_t *__restrict__ out)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= n) return;
    const int32_t p = in[i];
    out[i] = table[p & 0x3ff] | (table[1024 + ((p >> 10) & 0x3ff)] << 10) | (table[2048 + ((p >> 20) & 0x3ff)] << 20);
}
// corner coordinates of the nodes from their positions (coord: [3][1024], the coordinate of every position)
__global__ void __launch_bounds__(kBlock) k_lat_xyz_from_pos(int64_t nn, const int32_t *__restrict__ pos, const double *__restrict__ coord,
                                                              double *__restrict__ xyz)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= nn) return;
    const int32_t p = pos[i];
    for (int d = 0; d < 3; ++d) xyz[d * nn + i] = coord[1024 * d + ((p >> (10 * d)) & 0x3ff)];
}
// owned dofs that got no position (pos < 0: lattice_assign with fill 0xff): *count > 0
__global__ void __launch_bounds__(kBlock) k_lat_count_unplaced(int64_t n, const int32_t *__restrict__ pos, int *__restrict__ count)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i < n && pos[i] < 0) atomicAdd(count, 1);
}
// which positions of every axis are occupied (used: [3][1024], zeroed)
__global__ void __launch_bounds__(kBlock) k_lat_mark_used(int64_t n, const int32_t *__restrict__ pos, int32_t *__restrict__ used)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= n) return;
    const int32_t p = pos[i];
    used[p & 0x3ff] = 1;
    used[1024 + ((p >> 10) & 0x3ff)] = 1;
    used[2048 + ((p >> 20) & 0x3ff)] = 1;
}
// owned dofs' positions into the global array of the replicated level (zeroed before, summed over the ranks after)
__global__ void __launch_bounds__(kBlock) k_lat_emit_global_pos(int64_t n_own, const int32_t *__restrict__ gid, const int32_t *__restrict__ pos,
                                                                 double *__restrict__ out)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i < n_own) out[gid[i]] = static_cast<double>(pos[i]);
}
// ghost nodes [n0, n1): their aggregates came from the owners (amg_couple_level); the bricks they sit in get those numbers
// in the rank table, and the ghost aggregates their brick coordinates (every member writes the same value)
__global__ void __launch_bounds__(kBlock) k_lat_ghost_bricks(int64_t n0, int64_t n1, const int32_t *__restrict__ pos, LatBricks B,
                                                              const int32_t *__restrict__ agg, int32_t *__restrict__ brick_rank,
                                                              int32_t *__restrict__ pos_c)
{
    const int64_t i = n0 + static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= n1) return;
    const int32_t p = pos[i];
    const int bx = (p & 0x3ff) >> B.shift[0], by = ((p >> 10) & 0x3ff) >> B.shift[1], bz = ((p >> 20) & 0x3ff) >> B.shift[2];
    const int32_t a = agg[i];
    brick_rank[lat_linear(B, bx, by, bz)] = a;
    pos_c[a] = bx | (by << 10) | (bz << 20);
}
// diagonal of a matrix whose rows are not in ascending column order (coarse levels of a hierarchy across ranks: ghost
// columns are numbered behind the owned ones, whatever their brick)
__global__ void __launch_bounds__(kBlock) k_extract_diag_scan(SellDev A, double *__restrict__ diag)
{
    const int64_t r = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (r >= A.n_rows) return;
    const int64_t base = A.slice_off[r >> 6] + (r & 63);
    const int len = A.rowlen[r];
    double d = 0.0;
    for (int k = 0; k < len; ++k)
        if (A.cols[base + 64LL * k] == static_cast<int32_t>(r)) d += A.vals[base + 64LL * k];
    diag[r] = d;
}
// Galerkin maps without 64-bit keys: coarse entry of a stored fine entry = 27 * agg(row) + offset code of the column's brick
// (25 bits at config 3: four radix passes instead of eight); padding slots get the sentinel n_keys
__global__ void __launch_bounds__(kBlock) k_lat_emit_keys(SellDev A, const int32_t *__restrict__ pos, const int32_t *__restrict__ agg, LatBricks B,
                                                           uint32_t sentinel, uint32_t *__restrict__ keys, int32_t *__restrict__ slots)
{
    const int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (t >= A.n_slices * 64) return;
    const int64_t sl = t >> 6;
    const int64_t off = A.slice_off[sl];
    const int width = static_cast<int>((A.slice_off[sl + 1] - off) >> 6);
    const bool live = t < A.n_rows;
    const int len = live ? A.rowlen[t] : 0;
    const int32_t pi = live ? pos[t] : 0;
    const uint32_t a27 = live ? 27u * static_cast<uint32_t>(agg[t]) : 0u;
    const int bix = (pi & 0x3ff) >> B.shift[0], biy = ((pi >> 10) & 0x3ff) >> B.shift[1], biz = ((pi >> 20) & 0x3ff) >> B.shift[2];
    for (int k = 0; k < width; ++k) {
        const int64_t q = off + 64LL * k + (t & 63);
        uint32_t key = sentinel;
        if (k < len) {
            const int32_t pj = pos[A.cols[q]];
            const int dx = ((pj & 0x3ff) >> B.shift[0]) - bix, dy = (((pj >> 10) & 0x3ff) >> B.shift[1]) - biy, dz = (((pj >> 20) & 0x3ff) >> B.shift[2]) - biz;
            key = a27 + static_cast<uint32_t>((dx + 1) + 3 * (dy + 1) + 9 * (dz + 1));
        }
        keys[q] = key;
        slots[q] = static_cast<int32_t>(q);
    }
}
__global__ void __launch_bounds__(kBlock) k_lat_run_heads(int64_t n, const uint32_t *__restrict__ skeys, int32_t *__restrict__ head)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i < n) head[i] = (i == 0 || skeys[i] != skeys[i - 1]) ? 1 : 0;
    if (i == n) head[n] = 0;
}
// run starts + the coarse entry's (row << 32 | col) key: col = the aggregate of the brick at the coded offset from the row's
__global__ void __launch_bounds__(kBlock) k_lat_run_scatter(int64_t n, const uint32_t *__restrict__ skeys, const int32_t *__restrict__ head,
                                                             const int32_t *__restrict__ rank, uint32_t sentinel, const int32_t *__restrict__ pos_c,
                                                             LatBricks B, const int32_t *__restrict__ brick_rank, uint64_t *__restrict__ ukeys,
                                                             int64_t *__restrict__ start)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i < n && head[i]) {
        const uint32_t key = skeys[i];
        start[rank[i]] = i;
        if (key == sentinel) {
            ukeys[rank[i]] = ~0ull;
        } else {
            const uint32_t I = key / 27u, code = key % 27u;
            const int32_t pc = pos_c[I];
            const int bx = (pc & 0x3ff) + static_cast<int>(code % 3u) - 1, by = ((pc >> 10) & 0x3ff) + static_cast<int>((code / 3u) % 3u) - 1,
                      bz = ((pc >> 20) & 0x3ff) + static_cast<int>(code / 9u) - 1;
            const int32_t J = brick_rank[lat_linear(B, bx, by, bz)];
            ukeys[rank[i]] = (static_cast<uint64_t>(I) << 32) | static_cast<uint32_t>(J);
        }
    }
    if (i == n) start[rank[n]] = n;
}
// The same maps without any sort, from the member lists: one thread per COARSE row walks its (at most 8) member rows twice.
// First walk (k_lat_codes_count): how many stored entries fall into each of the 27 neighbouring bricks -- per-thread counters in
// LDS, [code][thread] -- giving the row's number of coarse entries and of (entry, slot) pairs; after two scans over the coarse
// rows the second walk (k_lat_codes_fill) writes the coarse entries' keys (ascending offset code = ascending coarse column: the
// aggregates are numbered in the bricks' z, y, x order), the start of every entry's slot list, and the slots themselves --
// member rows ascending, entries of a row ascending: a fixed summation order for the Galerkin product.
constexpr int kLatCodes = 27;
__device__ __forceinline__ int lat_code(const LatBricks &B, int32_t pj, int bix, int biy, int biz)
{
    const int dx = ((pj & 0x3ff) >> B.shift[0]) - bix, dy = (((pj >> 10) & 0x3ff) >> B.shift[1]) - biy, dz = (((pj >> 20) & 0x3ff) >> B.shift[2]) - biz;
    return (dx + 1) + 3 * (dy + 1) + 9 * (dz + 1);
}
__global__ void __launch_bounds__(kBlock) k_lat_codes_count(SellDev A, const int32_t *__restrict__ pos, LatBricks B, int64_t na,
                                                             const int32_t *__restrict__ mem_ptr, const int32_t *__restrict__ mem_idx,
                                                             const int32_t *__restrict__ pos_c, uint16_t *__restrict__ code_cnt,
                                                             int32_t *__restrict__ n_entries, int32_t *__restrict__ n_pairs, int *__restrict__ too_long)
{
    __shared__ uint16_t cnt[kLatCodes][kBlock];
    const int64_t I = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (I == na) { n_entries[na] = 0; n_pairs[na] = 0; }
    if (I >= na) return;
    const int t = threadIdx.x;
#pragma unroll
    for (int c = 0; c < kLatCodes; ++c) cnt[c][t] = 0;
    const int32_t pc = pos_c[I];
    const int bix = pc & 0x3ff, biy = (pc >> 10) & 0x3ff, biz = (pc >> 20) & 0x3ff;
    int pairs = 0;
    for (int32_t m = mem_ptr[I]; m < mem_ptr[I + 1]; ++m) {
        const int64_t i = mem_idx[m];
        const int64_t base = A.slice_off[i >> 6] + (i & 63);
        const int len = A.rowlen[i];
        pairs += len;
        for (int k = 0; k < len; ++k) ++cnt[lat_code(B, pos[A.cols[base + 64LL * k]], bix, biy, biz)][t];
    }
    int entries = 0;
#pragma unroll
    for (int c = 0; c < kLatCodes; ++c) {
        const uint16_t v = cnt[c][t];
        code_cnt[static_cast<int64_t>(c) * na + I] = v;
        entries += v != 0;
    }
    n_entries[I] = entries;
    n_pairs[I] = pairs;
    if (pairs > 0xffff) *too_long = 1;           // (16-bit counters: the sorted form takes the level)
}
template <bool MAPS>          // MAPS: also the (entry -> slots) lists of the map-driven product (A/B runs)
__global__ void __launch_bounds__(kBlock) k_lat_codes_fill(SellDev A, const int32_t *__restrict__ pos, LatBricks B, int64_t na,
                                                            const int32_t *__restrict__ mem_ptr, const int32_t *__restrict__ mem_idx,
                                                            const int32_t *__restrict__ pos_c, const int32_t *__restrict__ brick_rank,
                                                            const uint16_t *__restrict__ code_cnt, const int32_t *__restrict__ entry_off,
                                                            const int32_t *__restrict__ pair_off, uint64_t *__restrict__ ukeys,
                                                            int64_t *__restrict__ src_ptr, int32_t *__restrict__ src_slot,
                                                            uint8_t *__restrict__ code_of, uint32_t *__restrict__ code_mask)
{
    __shared__ uint16_t at[kLatCodes][kBlock];          // next place of every code's run, relative to the row's first pair
    const int64_t I = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (I >= na) return;
    const int t = threadIdx.x;
    const int32_t pc = pos_c[I];
    const int bix = pc & 0x3ff, biy = (pc >> 10) & 0x3ff, biz = (pc >> 20) & 0x3ff;
    const int64_t p0 = pair_off[I];
    int64_t e = entry_off[I];
    int run = 0;
#pragma unroll
    for (int c = 0; c < kLatCodes; ++c) {
        const int v = code_cnt[static_cast<int64_t>(c) * na + I];
        at[c][t] = static_cast<uint16_t>(run);
        if (v != 0) {
            const int bx = bix + c % 3 - 1, by = biy + (c / 3) % 3 - 1, bz = biz + c / 9 - 1;
            const int32_t J = brick_rank[lat_linear(B, bx, by, bz)];
            ukeys[e] = (static_cast<uint64_t>(I) << 32) | static_cast<uint32_t>(J);
            if (MAPS) src_ptr[e] = p0 + run;
            ++e;
        }
        run += v;
    }
    if (MAPS && I == na - 1) src_ptr[e] = p0 + run;
    uint32_t mask = 0;
    for (int32_t m = mem_ptr[I]; m < mem_ptr[I + 1]; ++m) {
        const int64_t i = mem_idx[m];
        const int64_t base = A.slice_off[i >> 6] + (i & 63);
        const int len = A.rowlen[i];
        for (int k = 0; k < len; ++k) {
            const int64_t q = base + 64LL * k;
            const int c = lat_code(B, pos[A.cols[q]], bix, biy, biz);
            if (MAPS) src_slot[p0 + at[c][t]++] = static_cast<int32_t>(q);
            code_of[q] = static_cast<uint8_t>(c);
            mask |= 1u << c;
        }
    }
    code_mask[I] = mask;
}
// Numeric Galerkin product of a brick level, by coarse ROW: the member rows are walked as in the symbolic phase, every stored
// value added to its code's accumulator in LDS ([code][thread]), the occupied codes written out in ascending order = the
// coarse row's entries.  Same additions in the same order as the map-driven k_amg_galerkin (member rows ascending, entries of
// a row ascending): the same bits, read as 8 + 1 bytes per fine entry along the rows instead of slot gathers per coarse entry.
constexpr int kLatGalThreads = 128;
// Round 6.  (1) The thread was a chain of dependent loads -- member -> row length / slice offset -> entries, member after
// member, eight of them, ten waves to the CU (the accumulators' LDS): level 0's product ran at 2.3 TB/s, 460 us of a solve's
// 1.05 ms of numeric set-up.  Now the members' indices, row lengths and slice offsets are requested together (three round trips
// for all eight), and a row's entries sixteen at a time; the additions keep their order.  (2) While the coarse row is in the
// thread's hands it also leaves behind what the rest of the numeric phase would read the coarse matrix for again: the inverse
// diagonal and the row's share of the Gershgorin bound (k_amg_diag_bound's sums, in slot order: dinv_out / ratio_out; the
// diagonal is offset code 13, the brick itself), and the row's value CODES against the level's dictionary (k_vd_encode16's
// look-up through the dictionary's hash table, pfem_vdhash.hpp: vhash / codes / vstate; a value the dictionary lacks raises miss).
struct LatGalExtra {
    double *dinv_out, *ratio_out;          // [na] or null
    const VdHashEntry *vhash;              // or null
    uint16_t *codes;                       // [coarse stored]
    VdState *vstate;
};
// (Tried and dropped, round 6: the 27 accumulators in REGISTERS -- the code of "entry k of member j" is the same in nearly every lane
// of a wave on a lattice, so it was read from one lane and the lanes that agree added under a wave-uniform switch, the others in
// further turns.  Occupancy 10 -> 16 waves a CU and no LDS round trips, yet level 0 took 677 us against 545 and the small levels --
// all boundary, up to 27 turns an entry -- 110-150 us against 25-35.)
__global__ void __launch_bounds__(kLatGalThreads) k_lat_galerkin(SellDev A, int64_t na, const int32_t *__restrict__ mem_ptr,
                                                                  const int32_t *__restrict__ mem_idx, const uint8_t *__restrict__ code_of,
                                                                  const uint32_t *__restrict__ code_mask, const int64_t *__restrict__ c_slice_off,
                                                                  double *__restrict__ coarse_vals, LatGalExtra X)
{
    __shared__ double acc[kLatCodes][kLatGalThreads];
    const int64_t I = static_cast<int64_t>(blockIdx.x) * kLatGalThreads + threadIdx.x;
    if (I >= na) return;
    const int t = threadIdx.x;
#pragma unroll
    for (int c = 0; c < kLatCodes; ++c) acc[c][t] = 0.0;
    const int32_t m0 = mem_ptr[I], m1 = mem_ptr[I + 1];
    for (int32_t mb = m0; mb < m1; mb += 8) {
        int32_t mi[8];
        int len[8];
        int64_t base[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) mi[j] = mb + j < m1 ? __builtin_nontemporal_load(mem_idx + mb + j) : -1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            len[j] = mi[j] >= 0 ? A.rowlen[mi[j]] : 0;
            base[j] = mi[j] >= 0 ? A.slice_off[mi[j] >> 6] + (mi[j] & 63) : 0;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            for (int k0 = 0; k0 < len[j]; k0 += 16) {
                double v[16];
                uint8_t c[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const bool in = k0 + e < len[j];
                    const int64_t q = base[j] + 64LL * (k0 + e);
                    v[e] = in ? __builtin_nontemporal_load(A.vals + q) : 0.0;
                    c[e] = in ? __builtin_nontemporal_load(code_of + q) : static_cast<uint8_t>(0);
                }
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (k0 + e < len[j]) acc[c[e]][t] += v[e];
            }
        }
    }
    const int64_t out0 = c_slice_off[I >> 6] + (I & 63);
    double *out = coarse_vals + out0;
    const uint32_t mask = code_mask[I];
    int r = 0;
    double sabs = 0.0;
    bool missed = false;
#pragma unroll
    for (int c = 0; c < kLatCodes; ++c)
        if (mask & (1u << c)) {
            const double v = acc[c][t];
            out[64LL * r] = v;
            sabs += fabs(v);
            if (X.vhash) {
                int code = vd_hash_find(X.vhash, static_cast<unsigned long long>(__double_as_longlong(v)));
                if (code < 0) { missed = true; code = 0; }
                X.codes[out0 + 64LL * r] = static_cast<uint16_t>(code);
            }
            ++r;
        }
    if (X.dinv_out) {
        const double d = (mask & (1u << 13)) ? acc[13][t] : 0.0;
        const bool pos = d > 0.0;
        X.dinv_out[I] = pos ? 1.0 / d : 1.0;
        X.ratio_out[I] = pos ? sabs / d : 1.0;
    }
    if (missed) X.vstate->miss = 1;
}
// distinct values of one coordinate: every node drops its value into a small open-addressing table (a lattice has a few
// hundred distinct values per axis, so almost every probe finds its value already there); *overflow when the table fills
constexpr int kLatticeTable = 4096;
__global__ void __launch_bounds__(kBlock) k_amg_distinct(const double *__restrict__ c, int64_t n, unsigned long long *__restrict__ table,
                                                          int *__restrict__ overflow)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    const double v = (i < n ? c[i] : 0.0) + 0.0;                  // -0.0 -> +0.0
    const unsigned long long key = static_cast<unsigned long long>(__double_as_longlong(v));
    // a lane whose left neighbour holds the same value leaves the table to it (along a line of a lattice two of the three
    // coordinates are the same for the whole wave)
    const unsigned long long left = __shfl_up(key, 1, 64);
    if (i >= n || ((threadIdx.x & 63) != 0 && left == key && i - 1 < n)) return;
    if (key == ~0ull) { *overflow = 1; return; }                   // (the empty marker is a NaN pattern: not a coordinate)
    unsigned h = static_cast<unsigned>((key * 0x9E3779B97F4A7C15ull) >> 52) & (kLatticeTable - 1);
    for (int probe = 0; probe < kLatticeTable; ++probe) {
        // (a mesh without a lattice fills the table at once: nobody needs to walk it to the end to learn that again)
        if ((probe & 63) == 0 && __hip_atomic_load(overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
        const unsigned long long cur = table[h];
        if (cur == key) return;
        if (cur == ~0ull) {
            const unsigned long long old = atomicCAS(&table[h], ~0ull, key);
            if (old == ~0ull || old == key) return;
        }
        h = (h + 1) & (kLatticeTable - 1);
    }
    *overflow = 1;
}

// ---- coordinates handed down a coupled hierarchy: a coarse node sits at the minimum corner of its aggregate (exact and
// independent of any order of evaluation -- the level that is replicated on every rank finds ITS lattice from them)
__device__ __forceinline__ unsigned long long amg_sortable(double v)
{
    const unsigned long long b = static_cast<unsigned long long>(__double_as_longlong(v + 0.0));
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double amg_unsortable(unsigned long long k)
{
    const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double(static_cast<long long>(b));
}
__global__ void __launch_bounds__(kBlock) k_amg_node_xyz(MeshDev m, int64_t n_owned, const int32_t *__restrict__ node_of, int64_t n_nodes,
                                                          double *__restrict__ xyz)
{
    const int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (t >= m.nElem * m.npe) return;
    const int64_t e = t % m.nElem;
    const int a = static_cast<int>(t / m.nElem);
    const int32_t nd = m.conn[a * m.nElem + e];
    const int32_t l = m.edof[(a * m.ndof) * m.nElem + e];          // first dof of the node
    if (l < 0 || l >= n_owned) return;
    const int64_t node = node_of ? node_of[l] : l;
    if (node >= n_nodes) return;
    for (int d = 0; d < 3; ++d) xyz[d * n_nodes + node] = d < m.ndim ? m.xyz[static_cast<int64_t>(d) * m.nNode + nd] : 0.0;
}
__global__ void __launch_bounds__(kBlock) k_amg_xyz_min(int64_t nn, const int32_t *__restrict__ node_agg, const double *__restrict__ xyz, int64_t na,
                                                         unsigned long long *__restrict__ keys)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= nn) return;
    for (int d = 0; d < 3; ++d) atomicMin(&keys[d * na + node_agg[i]], amg_sortable(xyz[d * nn + i]));
}
__global__ void __launch_bounds__(kBlock) k_amg_xyz_decode(int64_t n, const unsigned long long *__restrict__ keys, double *__restrict__ xyz)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i < n) xyz[i] = amg_unsortable(keys[i]);
}
// owned nodes' coordinates into the global array of the replicated level (zeroed before, summed over the ranks after)
__global__ void __launch_bounds__(kBlock) k_amg_emit_global_xyz(int64_t n_nodes_own, const int32_t *__restrict__ gid /* null: node_off + i */,
                                                                 int64_t node_off, const double *__restrict__ xyz, int64_t n_glob,
                                                                 double *__restrict__ out)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= n_nodes_own) return;
    const int64_t g = gid ? gid[i] : node_off + i;
    for (int d = 0; d < 3; ++d) out[d * n_glob + g] = xyz[d * n_nodes_own + i];
}

// lattice position of a node from its coordinates and the sorted distinct values of every axis (exact matches)
__device__ __forceinline__ int amg_lattice_index(const double *__restrict__ u, int n, double v)
{
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (u[mid] < v) lo = mid + 1; else hi = mid; }
    return lo < n ? lo : n - 1;
}
__global__ void __launch_bounds__(kBlock) k_amg_lattice_pos(MeshDev m, int64_t n_owned, const double *__restrict__ ux, int nx,
                                                             const double *__restrict__ uy, int ny, const double *__restrict__ uz, int nz,
                                                             int32_t *__restrict__ pos)
{
    const int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (t >= m.nElem * m.npe) return;
    const int64_t e = t % m.nElem;
    const int a = static_cast<int>(t / m.nElem);
    const int32_t nd = m.conn[a * m.nElem + e];
    const int ix = amg_lattice_index(ux, nx, m.xyz[nd]), iy = amg_lattice_index(uy, ny, m.xyz[m.nNode + nd]);
    const int iz = m.ndim > 2 ? amg_lattice_index(uz, nz, m.xyz[2 * m.nNode + nd]) : 0;
    const int32_t p = ix | (iy << 10) | (iz << 20);
    for (int d = 0; d < m.ndof; ++d) {
        const int32_t l = m.edof[(a * m.ndof + d) * m.nElem + e];
        if (l >= 0 && l < n_owned) pos[l] = p;              // every visit of a dof writes the same value
    }
}

// lattice position of every owned dof from the node -> row table of the assembly (one thread per mesh node)
__global__ void __launch_bounds__(kBlock) k_amg_lattice_pos_nodes(int64_t nNode, int ndim, int ndof, const double *__restrict__ xyz,
                                                                   const int32_t *__restrict__ node_row, int64_t n_owned, const double *__restrict__ ux,
                                                                   int nx, const double *__restrict__ uy, int ny, const double *__restrict__ uz, int nz,
                                                                   int32_t *__restrict__ pos)
{
    const int64_t nd = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (nd >= nNode) return;
    int32_t p = -1;
    for (int d = 0; d < ndof; ++d) {
        const int32_t l = node_row[nd * ndof + d];
        if (l < 0 || l >= n_owned) continue;
        if (p < 0) {
            const int ix = amg_lattice_index(ux, nx, xyz[nd]), iy = amg_lattice_index(uy, ny, xyz[nNode + nd]);
            const int iz = ndim > 2 ? amg_lattice_index(uz, nz, xyz[2 * nNode + nd]) : 0;
            p = ix | (iy << 10) | (iz << 20);
        }
        pos[l] = p;
    }
}

// ---- a lattice by NUMBERING: nodes moved off their lattice sites (a mapped or graded block, a mesh after a moving-mesh step),
// but numbered like a box -- node n at (n % a, (n / a) % (b / a), n / b).  The strides a, b come from the incidence patterns
// (lattice_positions_by_numbering); this pass holds every element to them: the nodes of an element lie within one cell of each
// other along every axis (a stride that only looks right wraps around somewhere and is caught here)
__global__ void __launch_bounds__(kBlock) k_latnum_check(int64_t nNode, int ndof, int a, int b, int n2, const int64_t *__restrict__ inc_ptr,
                                                          const int32_t *__restrict__ inc_cnt, const int4 *__restrict__ inc_rec,
                                                          const int32_t *__restrict__ node_row, int *bad)
{
    const int64_t n = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (n >= nNode) return;
    bool has_row = false;
    for (int p = 0; p < ndof; ++p) has_row |= node_row[n * ndof + p] >= 0;
    if (!has_row) return;
    const int i0 = static_cast<int>(n % a), j0 = static_cast<int>((n % b) / a), k0 = static_cast<int>(n / b);
    bool ok = k0 < n2;
    const int cnt = inc_cnt[n];
    const int64_t beg = inc_ptr[n >> 6] + (n & 63);
    for (int t = 0; t < cnt && ok; ++t) {
        const int4 rc = inc_rec[beg + 64LL * t];
        const int o[3] = {rc.x & 0x7fffffff, rc.y & 0x7fffffff, rc.z};
        for (int q = 0; q < 3; ++q) {
            const int di = o[q] % a - i0, dj = (o[q] % b) / a - j0, dk = o[q] / b - k0;
            ok = ok && di >= -1 && di <= 1 && dj >= -1 && dj <= 1 && dk >= -1 && dk <= 1 && o[q] / b < n2;
        }
    }
    if (!ok) *bad = 1;
}
__global__ void __launch_bounds__(kBlock) k_latnum_pos(int64_t nNode, int ndof, int a, int b, const int32_t *__restrict__ node_row, int64_t n_owned,
                                                        int32_t *__restrict__ pos)
{
    const int64_t n = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (n >= nNode) return;
    const int32_t p = static_cast<int32_t>(n % a) | (static_cast<int32_t>((n % b) / a) << 10) | (static_cast<int32_t>(n / b) << 20);
    for (int d = 0; d < ndof; ++d) {
        const int32_t l = node_row[n * ndof + d];
        if (l >= 0 && l < n_owned) pos[l] = p;
    }
}

// place of a node = place of its first dof / dofs per node
__global__ void __launch_bounds__(kBlock) k_amg_node_hint(int64_t n, const int32_t *__restrict__ node_of, const int32_t *__restrict__ comp_of,
                                                           const int32_t *__restrict__ dof_rank, int bs, int32_t *__restrict__ hint)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= n) return;
    if (!node_of) hint[i] = dof_rank[i];
    else if (comp_of[i] == 0) hint[node_of[i]] = dof_rank[i] / bs;
}
__global__ void __launch_bounds__(kBlock) k_amg_iota(int64_t n, int32_t *__restrict__ v)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i < n) v[i] = static_cast<int32_t>(i);
}

// first-pass strength graph of a matrix whose dofs are grouped in nodes (node_of, up to `bs` dofs each): one key per
// stored entry, (node(row) << 32 | node(col)) with the SQUARED entry (reduced by key, then -sqrt: the Frobenius norm of
// the node block).  Padding entries of the wave-sliced storage get the sentinel key.
__global__ void __launch_bounds__(kBlock) k_amg_emit_node_keys(SellDev A, const int32_t *__restrict__ node_of, int32_t col_limit, int squared,
                                                                uint64_t *__restrict__ keys, double *__restrict__ vals)
{
    const int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;      // one thread per row, entries strided by 64
    if (t >= A.n_slices * 64) return;
    const int64_t sl = t >> 6;
    const int64_t off = A.slice_off[sl];
    const int width = static_cast<int>((A.slice_off[sl + 1] - off) >> 6);
    const int len = t < A.n_rows ? A.rowlen[t] : 0;
    const int32_t nr = t < A.n_rows ? (node_of ? node_of[t] : static_cast<int32_t>(t)) : 0;
    for (int k = 0; k < width; ++k) {
        const int64_t q = off + 64LL * k + (t & 63);
        const int32_t c = k < len ? A.cols[q] : col_limit;
        if (c < col_limit) {                 // columns from col_limit on are another rank's dofs (ghosts): not in this block
            const double v = A.vals[q];
            keys[q] = (static_cast<uint64_t>(static_cast<uint32_t>(nr)) << 32) | static_cast<uint32_t>(node_of ? node_of[c] : c);
            vals[q] = squared ? v * v : v;
        } else {
            keys[q] = ~0ull;
            vals[q] = 0.0;
        }
    }
}

// runs of equal keys in a sorted array: flag the run heads, (scan), scatter head key + head position, sum each run in order
__global__ void __launch_bounds__(kBlock) k_amg_run_heads(int64_t n, const uint64_t *__restrict__ skeys, int32_t *__restrict__ head)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i < n) head[i] = (i == 0 || skeys[i] != skeys[i - 1]) ? 1 : 0;
    if (i == n) head[n] = 0;
}
__global__ void __launch_bounds__(kBlock) k_amg_run_scatter(int64_t n, const uint64_t *__restrict__ skeys, const int32_t *__restrict__ head,
                                                             const int32_t *__restrict__ rank, uint64_t *__restrict__ ukeys, int64_t *__restrict__ start)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i < n && head[i]) { ukeys[rank[i]] = skeys[i]; start[rank[i]] = i; }
    if (i == n) start[rank[n]] = n;
}
__global__ void __launch_bounds__(kBlock) k_amg_run_sums(int64_t m, const int64_t *__restrict__ start, const double *__restrict__ svals,
                                                          const uint64_t *__restrict__ ukeys, double *__restrict__ sums)
{
    const int64_t u = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (u >= m) return;
    if (ukeys[u] == ~0ull) { sums[u] = 0.0; return; }      // the run of the padding entries (it can be millions long) is dropped anyway
    double a = 0.0;
    for (int64_t q = start[u]; q < start[u + 1]; ++q) a += svals[q];
    sums[u] = a;
}

// reduced node keys -> graph arrays: columns, weights (-norm off the diagonal), diagonal norms
__global__ void __launch_bounds__(kBlock) k_amg_graph_from_keys(int64_t m, const uint64_t *__restrict__ ukeys, const double *__restrict__ sums,
                                                                 int squared, int32_t *__restrict__ gcol, double *__restrict__ gw,
                                                                 double *__restrict__ gdiag)
{
    const int64_t q = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (q >= m) return;
    const uint64_t k = ukeys[q];
    const int32_t r = static_cast<int32_t>(k >> 32), c = static_cast<int32_t>(k & 0xffffffffu);
    const double v = sums[q];
    gcol[q] = c;
    if (r == c) {
        gdiag[r] = squared ? sqrt(v) : v;
        gw[q] = 0.0;
    } else {
        gw[q] = squared ? -sqrt(v) : v;
    }
}

// the graph of the next matching pass: edge {i,j} -> {agg(i), agg(j)}, weights summed by key afterwards.  Entry q of the
// CSR graph; the diagonal rides along as one more key per node (index nnz + i).  An edge INSIDE an aggregate adds to the
// aggregate's diagonal (Galerkin) for scalar problems; for node blocks, whose weights are norms, it is dropped.
__global__ void __launch_bounds__(kBlock) k_amg_emit_coarse_graph(int64_t n, int64_t nnz, const int64_t *__restrict__ gptr,
                                                                   const int32_t *__restrict__ gcol, const double *__restrict__ gw,
                                                                   const double *__restrict__ gdiag, const int32_t *__restrict__ agg,
                                                                   int blocks, uint64_t *__restrict__ keys, double *__restrict__ vals)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= n) return;
    const uint64_t ai = static_cast<uint32_t>(agg[i]);
    for (int64_t q = gptr[i]; q < gptr[i + 1]; ++q) {
        const int32_t j = gcol[q];
        const uint64_t aj = static_cast<uint32_t>(agg[j]);
        keys[q] = (ai << 32) | aj;
        vals[q] = (j == i) ? 0.0 : ((ai == aj && blocks) ? 0.0 : gw[q]);
    }
    keys[nnz + i] = (ai << 32) | ai;
    vals[nnz + i] = gdiag[i];
}

// coarse dof of a fine dof: the (aggregate of its node, its component) pairs that occur, numbered in ascending order
__global__ void __launch_bounds__(kBlock) k_amg_mark_cdofs(int64_t n, const int32_t *__restrict__ node_of, const int32_t *__restrict__ comp_of,
                                                            const int32_t *__restrict__ node_agg, int bs, int32_t *__restrict__ flag)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= n) return;
    const int64_t nd = node_of ? node_of[i] : i;
    flag[static_cast<int64_t>(node_agg[nd]) * bs + (comp_of ? comp_of[i] : 0)] = 1;
}
__global__ void __launch_bounds__(kBlock) k_amg_assign_cdofs(int64_t n, const int32_t *__restrict__ node_of, const int32_t *__restrict__ comp_of,
                                                              const int32_t *__restrict__ node_agg, int bs, const int32_t *__restrict__ rank,
                                                              int32_t *__restrict__ agg)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= n) return;
    const int64_t nd = node_of ? node_of[i] : i;
    agg[i] = rank[static_cast<int64_t>(node_agg[nd]) * bs + (comp_of ? comp_of[i] : 0)];
}
// node / component of every coarse dof (the next level's node_of / comp_of)
__global__ void __launch_bounds__(kBlock) k_amg_coarse_nodes(int64_t nslots, int bs, const int32_t *__restrict__ flag, const int32_t *__restrict__ rank,
                                                              int32_t *__restrict__ node_of_c, int32_t *__restrict__ comp_of_c)
{
    const int64_t q = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (q >= nslots || !flag[q]) return;
    node_of_c[rank[q]] = static_cast<int32_t>(q / bs);
    comp_of_c[rank[q]] = static_cast<int32_t>(q % bs);
}

// component of a dof inside its node (row - first row of its group)
__global__ void __launch_bounds__(kBlock) k_amg_comp_of(int64_t n, const int32_t *__restrict__ node_of, const int32_t *__restrict__ group_row0,
                                                         int32_t *__restrict__ comp_of)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i < n) comp_of[i] = static_cast<int32_t>(i - group_row0[node_of[i]]);
}
// int32 run lengths -> int64 (one extra zero at the end), scanned in place afterwards
__global__ void __launch_bounds__(kBlock) k_amg_widen(int64_t n_out, int64_t n_in, const int32_t *__restrict__ in, int64_t *__restrict__ out)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i < n_out) out[i] = i < n_in ? in[i] : 0;
}

// members of every coarse dof: histogram, then (after the scan) the sorted member list comes from a stable sort by agg
__global__ void __launch_bounds__(kBlock) k_amg_count(int64_t n, const int32_t *__restrict__ agg, int32_t *__restrict__ cnt)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i < n) atomicAdd(&cnt[agg[i]], 1);
}

// Galerkin product with piecewise-constant prolongation: coarse entry (agg(r), agg(c)) = sum of the fine entries that map
// to it.  One key per stored fine entry with its storage slot as payload (sorted by key afterwards; the radix sort is
// stable, so the slots of one coarse entry stay in ascending order: the numeric sum has a fixed order).
__global__ void __launch_bounds__(kBlock) k_amg_emit_rap_keys(SellDev A, const int32_t *__restrict__ agg, int32_t col_limit,
                                                               uint64_t *__restrict__ keys, int32_t *__restrict__ slots)
{
    const int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (t >= A.n_slices * 64) return;
    const int64_t sl = t >> 6;
    const int64_t off = A.slice_off[sl];
    const int width = static_cast<int>((A.slice_off[sl + 1] - off) >> 6);
    const int len = t < A.n_rows ? A.rowlen[t] : 0;
    const uint64_t ar = t < A.n_rows ? static_cast<uint32_t>(agg[t]) : 0u;
    for (int k = 0; k < width; ++k) {
        const int64_t q = off + 64LL * k + (t & 63);
        const int32_t c = k < len ? A.cols[q] : col_limit;
        keys[q] = c < col_limit ? ((ar << 32) | static_cast<uint32_t>(agg[c])) : ~0ull;     // ghost columns: not in this rank's block
        slots[q] = static_cast<int32_t>(q);
    }
}

// storage slot of every coarse entry (unique keys in ascending order = row-major order of the coarse matrix)
__global__ void __launch_bounds__(kBlock) k_amg_dst_slots(int64_t nnz_c, const uint64_t *__restrict__ ukeys, const int64_t *__restrict__ rowptr,
                                                           const int64_t *__restrict__ slice_off, int64_t *__restrict__ dst)
{
    const int64_t c = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (c >= nnz_c) return;
    const int64_t row = static_cast<int64_t>(ukeys[c] >> 32);
    dst[c] = slice_off[row >> 6] + 64LL * (c - rowptr[row]) + (row & 63);
}

// ---------------------------------------------------------------------------
// numeric phase (every solve): coarse values, smoother data, the dense inverse of the coarsest operator
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_amg_galerkin(int64_t nnz_c, const int64_t *__restrict__ src_ptr, const int32_t *__restrict__ src_slot,
                                                          const int64_t *__restrict__ dst, const double *__restrict__ fine_vals,
                                                          double *__restrict__ coarse_vals)
{
    const int64_t c = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (c >= nnz_c) return;
    // (a coarse entry sums 4 fine entries on average and ~46 on the diagonal of a 2x2x2 brick: slots first, then all
    // values -- eight gathers in flight instead of a chain of dependent pairs; added in the same ascending order.
    // A wave-cooperative form -- the 64 entries' contiguous stretch of slots fetched by all lanes into LDS, every lane
    // then summing its part from there -- was built and measured no faster: 1.73 against 1.66 ms for the phase.)
    double a = 0.0;
    int64_t q = src_ptr[c];
    const int64_t q1 = src_ptr[c + 1];
    for (; q + 8 <= q1; q += 8) {
        int32_t sl[8];
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) sl[k] = src_slot[q + k];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = fine_vals[sl[k]];
#pragma unroll
        for (int k = 0; k < 8; ++k) a += v[k];
    }
    if (q < q1) {
        int32_t sl[8];
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) sl[k] = q + k < q1 ? src_slot[q + k] : -1;
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = sl[k] >= 0 ? fine_vals[sl[k]] : 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (sl[k] >= 0) a += v[k];
    }
    coarse_vals[dst[c]] = a;
}

// inverse diagonal and the Gershgorin bound max_i sum_j |a_ij| / a_ii >= lambda_max(D^-1 A), one pass over the matrix
__global__ void __launch_bounds__(kBlock) k_amg_diag_bound(SellDev A, int32_t col_limit, double *__restrict__ dinv, double *__restrict__ part_max)
{
    __shared__ double sm[4];
    const int64_t r = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    double ratio = 0.0;
    if (r < A.n_slices * 64) {
        const int64_t sl = r >> 6;
        const int64_t off = A.slice_off[sl];
        const int width = static_cast<int>((A.slice_off[sl + 1] - off) >> 6);
        double d = 0.0, s = 0.0;
        for (int k = 0; k < width; ++k) {
            const int64_t q = off + 64LL * k + (r & 63);
            const double v = A.vals[q];
            const int32_t c = A.cols[q];
            if (c == static_cast<int32_t>(r) && v != 0.0) d += v;      // padding points at the own row with value 0
            if (c < col_limit) s += fabs(v);                           // ghost columns are not part of this rank's block
        }
        if (r < A.n_rows) {
            const bool ok = d > 0.0;
            dinv[r] = ok ? 1.0 / d : 1.0;
            ratio = ok ? s / d : 1.0;
        }
    }
    // block maximum (wave shuffles, then the four waves)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ratio = fmax(ratio, __shfl_xor(ratio, o, 64));
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = ratio;
    __syncthreads();
    if (threadIdx.x == 0) part_max[blockIdx.x] = fmax(fmax(sm[0], sm[1]), fmax(sm[2], sm[3]));
}
// part_max[block] = max of v[0..n) over the block's grid-stride share (v >= 0)
__global__ void __launch_bounds__(kBlock) k_amg_max_rows(const double *__restrict__ v, int64_t n, double *__restrict__ part_max)
{
    __shared__ double sm[4];
    double a = 0.0;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * kBlock)
        a = fmax(a, __builtin_nontemporal_load(v + i));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a = fmax(a, __shfl_xor(a, o, 64));
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) part_max[blockIdx.x] = fmax(fmax(sm[0], sm[1]), fmax(sm[2], sm[3]));
}
__global__ void __launch_bounds__(1024) k_amg_max(const double *__restrict__ part, int64_t n, double *out)
{
    __shared__ double sm[16];
    double a = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) a = fmax(a, part[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a = fmax(a, __shfl_xor(a, o, 64));
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < 16; ++w) t = fmax(t, sm[w]);
        out[0] = t;
    }
}

// coarsest operator (n <= kAmgDense rows) -> dense, inverted in LDS by Gauss-Jordan without pivoting (SPD), one block
constexpr int kAmgDense = 128;
__device__ __forceinline__ void amg_gauss_jordan(double *M, double *colp, int n, double *__restrict__ inv);
__global__ void __launch_bounds__(1024) k_amg_dense_inverse(SellDev A, double *__restrict__ inv)
{
    extern __shared__ double M[];                   // n x n
    __shared__ double colp[kAmgDense];
    const int n = static_cast<int>(A.n_rows);
    for (int q = threadIdx.x; q < n * n; q += 1024) M[q] = 0.0;
    __syncthreads();
    // (round 6: lane = row, wave = entry -- sixteen entries of every row in flight; one thread per row walked its 64 entries one
    // dependent load after the other: 60 of this kernel's 97 us)
    for (int r = threadIdx.x & 63; r < n; r += 64) {
        const int64_t base = A.slice_off[r >> 6] + (r & 63);
        const int len = A.rowlen[r];
        for (int k = threadIdx.x >> 6; k < len; k += 16) M[r * n + A.cols[base + 64LL * k]] = A.vals[base + 64LL * k];
    }
    __syncthreads();
    amg_gauss_jordan(M, colp, n, inv);
}
// the same from a dense n x n matrix in memory (several ranks: the last level's operator summed over the ranks)
__global__ void __launch_bounds__(1024) k_amg_dense_inverse_of(const double *__restrict__ dense, int n, double *__restrict__ inv)
{
    extern __shared__ double M[];
    __shared__ double colp[kAmgDense];
    for (int q = threadIdx.x; q < n * n; q += 1024) M[q] = dense[q];
    __syncthreads();
    amg_gauss_jordan(M, colp, n, inv);
}
__device__ __forceinline__ void amg_gauss_jordan(double *M, double *colp, int n, double *__restrict__ inv)
{
    // (round 6: every element of a pivot step depends on the OLD pivot row and column only, so a thread reads what its elements
    // need, all threads meet, and then it writes them: two barriers a step instead of five -- 96 -> ~45 us for the 64-row level of
    // config 3 in every solve; the same operations on the same operands: M[p][j] piv, fma(-M[i][p], M[p][j] piv, M[i][j]), -M[i][p] piv)
    (void)colp;
    constexpr int kPer = (kAmgDense * kAmgDense + 1023) / 1024;
    const int per = (n * n + 1023) / 1024;                  // elements a thread owns (uniform): 4 for 64 rows
    int qi[kPer], qj[kPer];                                  // ... their row and column, found once
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
        const int q = threadIdx.x + 1024 * u;
        qi[u] = q < n * n ? q / n : -1;
        qj[u] = q < n * n ? q - qi[u] * n : 0;
    }
    for (int p = 0; p < n; ++p) {
        double nv[kPer];
        const double piv = 1.0 / M[p * n + p];
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            nv[u] = 0.0;
            if (u < per && qi[u] >= 0) {
                const int i = qi[u], j = qj[u];
                const double cp = M[i * n + p], rp = M[p * n + j] * piv;
                nv[u] = (i == p) ? (j == p ? piv : rp) : (j == p ? -cp * piv : __builtin_fma(-cp, rp, M[i * n + j]));
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < kPer; ++u)
            if (u < per && qi[u] >= 0) M[qi[u] * n + qj[u]] = nv[u];
        __syncthreads();
    }
    for (int q = threadIdx.x; q < n * n; q += 1024) inv[q] = M[q];
}
// ---- coupled hierarchy on several ranks: pieces of the numeric phase that need the neighbours' shares ----
// this rank's share of the diagonal and of the absolute row sums (both are summed over the holders afterwards; the sum of
// the shares' absolute values bounds the absolute value of the summed entries, so the bound stays a bound)
__global__ void __launch_bounds__(kBlock) k_amg_diag_abs(SellDev A, double *__restrict__ diag, double *__restrict__ rowabs)
{
    const int64_t r = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (r >= A.n_rows) return;
    const int64_t base = A.slice_off[r >> 6] + (r & 63);
    const int len = A.rowlen[r];
    double d = 0.0, s = 0.0;
    for (int k = 0; k < len; ++k) {
        const double v = A.vals[base + 64LL * k];
        if (A.cols[base + 64LL * k] == static_cast<int32_t>(r)) d += v;
        s += fabs(v);
    }
    diag[r] = d;
    rowabs[r] = s;
}
__global__ void __launch_bounds__(kBlock) k_amg_dinv_ratio(int64_t n, const double *__restrict__ diag, const double *__restrict__ rowabs,
                                                            double *__restrict__ dinv, double *__restrict__ part_max)
{
    __shared__ double sm[4];
    const int64_t r = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    double ratio = 0.0;
    if (r < n) {
        const double d = diag[r];
        const bool ok = d > 0.0;
        dinv[r] = ok ? 1.0 / d : 1.0;
        ratio = ok ? rowabs[r] / d : 1.0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ratio = fmax(ratio, __shfl_xor(ratio, o, 64));
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = ratio;
    __syncthreads();
    if (threadIdx.x == 0) part_max[blockIdx.x] = fmax(fmax(sm[0], sm[1]), fmax(sm[2], sm[3]));
}
// v[i] = global number of the coarse dof of owned dof i, 0 on the ghosts: after the level's sum-exchange every holder
// of a dof knows its aggregate (numbers stay far below 2^53: exact in a double)
__global__ void __launch_bounds__(kBlock) k_amg_gid_vector(int64_t n, int64_t n_loc, const int32_t *__restrict__ agg, int64_t off, double *__restrict__ v)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i < n_loc) v[i] = i < n ? static_cast<double>(off + agg[i]) : 0.0;
}
// w[i] = holder set (bit q = rank q holds it, as a double: at most 52 ranks) of the coarse dof of owned dof i, 0 on the
// ghosts: after the level's sum-exchange every holder of a dof knows all holders of its aggregate
__global__ void __launch_bounds__(kBlock) k_amg_mask_vector(int64_t n, int64_t n_loc, const int32_t *__restrict__ agg, const double *__restrict__ mask_c,
                                                             double *__restrict__ w)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i < n_loc) w[i] = i < n ? mask_c[agg[i]] : 0.0;
}
// the ghosts' side of amg_couple_level on the device: sort keys = the aggregates' global numbers the owners sent (v holds
// them as doubles; undo = 0.5 in the self-peer timing probe), the local number of every ghost dof's aggregate from the run
// it falls into (table: whole coarse nodes), and one holder-set mask per run
__global__ void __launch_bounds__(kBlock) k_amg_ghost_keys(int64_t ng, const double *__restrict__ vg, double undo, uint64_t *__restrict__ keys,
                                                            int32_t *__restrict__ idx)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= ng) return;
    keys[i] = static_cast<uint64_t>(static_cast<int64_t>(undo * vg[i] + 0.5));
    idx[i] = static_cast<int32_t>(i);
}
__global__ void __launch_bounds__(kBlock) k_amg_ghost_agg(int64_t ng, const int32_t *__restrict__ sidx, const int32_t *__restrict__ head,
                                                           const int32_t *__restrict__ rank, const int32_t *__restrict__ table, int32_t nc,
                                                           int32_t *__restrict__ agg_g)
{
    const int64_t p = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (p >= ng) return;
    const int32_t run = rank[p] + head[p] - 1;
    agg_g[sidx[p]] = table ? table[run] : nc + run;
}
__global__ void __launch_bounds__(kBlock) k_amg_ghost_masks(int64_t m, const int64_t *__restrict__ start, const int32_t *__restrict__ sidx,
                                                             const double *__restrict__ wg, double *__restrict__ out)
{
    const int64_t k = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (k < m) out[k] = wg[sidx[start[k]]];
}
// this rank's share of the last level's operator into the dense global matrix (zeroed before; summed over the ranks after)
__global__ void __launch_bounds__(kBlock) k_amg_dense_scatter(SellDev A, const int32_t *__restrict__ gid, int n_glob, double *__restrict__ dense)
{
    const int64_t r = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (r >= A.n_rows) return;
    const int64_t base = A.slice_off[r >> 6] + (r & 63);
    const int len = A.rowlen[r];
    for (int k = 0; k < len; ++k) dense[static_cast<int64_t>(gid[r]) * n_glob + gid[A.cols[base + 64LL * k]]] = A.vals[base + 64LL * k];
}
// ---- coupled hierarchy, the step to the replicated levels: this rank's entries of the level with GLOBAL row / column
// numbers (as doubles: they travel through the all-reduce that concatenates the ranks' lists), row by row
__global__ void __launch_bounds__(kBlock) k_amg_emit_global_entries(SellDev A, const int64_t *__restrict__ rowptr, const int32_t *__restrict__ gid,
                                                                     double *__restrict__ rows, double *__restrict__ cols, int32_t *__restrict__ slot)
{
    const int64_t r = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (r >= A.n_rows) return;
    const int64_t base = A.slice_off[r >> 6] + (r & 63);
    const int len = A.rowlen[r];
    const int64_t p0 = rowptr[r];
    for (int k = 0; k < len; ++k) {
        rows[p0 + k] = static_cast<double>(gid[r]);
        cols[p0 + k] = static_cast<double>(gid[A.cols[base + 64LL * k]]);
        slot[p0 + k] = static_cast<int32_t>(base + 64LL * k);
    }
}
__global__ void __launch_bounds__(kBlock) k_amg_keys_from_pairs(int64_t n, const double *__restrict__ rows, const double *__restrict__ cols,
                                                                 uint64_t *__restrict__ keys, int32_t *__restrict__ pos)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= n) return;
    keys[i] = (static_cast<uint64_t>(rows[i] + 0.5) << 32) | static_cast<uint64_t>(cols[i] + 0.5);
    pos[i] = static_cast<int32_t>(i);
}
__global__ void __launch_bounds__(kBlock) k_amg_pack_values(int64_t n, const double *__restrict__ vals, const int32_t *__restrict__ slot,
                                                             double *__restrict__ out)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i < n) out[i] = vals[slot[i]];
}
// node and component of every owned dof of the level, by global dof number (summed over the ranks afterwards)
__global__ void __launch_bounds__(kBlock) k_amg_emit_global_nodes(int64_t n_own, const int32_t *__restrict__ gid, const int32_t *__restrict__ node_of,
                                                                   const int32_t *__restrict__ comp_of, int64_t node_off, double *__restrict__ node_g,
                                                                   double *__restrict__ comp_g)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= n_own) return;
    node_g[gid[i]] = static_cast<double>(node_off + node_of[i]);
    comp_g[gid[i]] = static_cast<double>(comp_of[i]);
}
__global__ void __launch_bounds__(kBlock) k_amg_round_to_int(int64_t n, const double *__restrict__ in, int32_t *__restrict__ out)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i < n) out[i] = static_cast<int32_t>(in[i] + 0.5);
}
__global__ void __launch_bounds__(kBlock) k_amg_scatter_gid(int64_t n_own, const double *__restrict__ b, const int32_t *__restrict__ gid,
                                                             double *__restrict__ out, const CgCtl *ctl)
{
    if (ctl && ctl->flag != 0) return;
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i < n_own) out[gid[i]] = b[i];
}
__global__ void __launch_bounds__(kBlock) k_amg_gather_gid(int64_t n_loc, const double *__restrict__ xg, const int32_t *__restrict__ gid,
                                                            double *__restrict__ x, const CgCtl *ctl)
{
    if (ctl && ctl->flag != 0) return;
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i < n_loc) x[i] = xg[gid[i]];
}
__global__ void __launch_bounds__(kBlock) k_amg_zero(int64_t n, double *__restrict__ v, const CgCtl *ctl)
{
    if (ctl && ctl->flag != 0) return;
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i < n) v[i] = 0.0;
}

// x = inv * b (one block, one thread per row; the inverse of an SPD matrix is symmetric: column access is coalesced)
__global__ void __launch_bounds__(kAmgDense) k_amg_dense_apply(int n, const double *__restrict__ inv, const double *__restrict__ b,
                                                                double *__restrict__ x, const CgCtl *ctl)
{
    __shared__ double sb[kAmgDense];
    if (ctl && ctl->flag != 0) return;
    const int i = threadIdx.x;
    if (i < n) sb[i] = b[i];
    __syncthreads();
    if (i >= n) return;
    double a = 0.0;
    for (int j = 0; j < n; ++j) a = __builtin_fma(inv[j * n + i], sb[j], a);
    x[i] = a;
}

// ---------------------------------------------------------------------------
// V-cycle vector kernels.  Chebyshev smoothing on D^-1 A over [lmax/ratio, lmax] (lmax: device scalar `lam`):
//   theta = (lmax+lmin)/2, delta = (lmax-lmin)/2, sigma = theta/delta, rho_0 = 1/sigma, rho_k = 1/(2 sigma - rho_{k-1})
//   d_0 = D^-1 r / theta,  d_k = rho_k rho_{k-1} d_{k-1} + 2 rho_k/delta D^-1 (r - A(d_0+...+d_{k-1}))    x += d_k
// Every kernel leaves at once when the solve has finished (the tail of a chunk of enqueued iterations).
// ---------------------------------------------------------------------------
struct ChebCoef { double c_first, c_dd, c_r; };
__device__ __forceinline__ ChebCoef cheb_coef(double lmax, double ratio, int step)
{
    const double lmin = lmax / ratio, theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin), sigma = theta / delta;
    double rho = 1.0 / sigma, rho_new = rho;
    ChebCoef c{1.0 / theta, 0.0, 0.0};
    for (int k = 1; k <= step; ++k) {
        rho_new = 1.0 / (2.0 * sigma - rho);
        c.c_dd = rho_new * rho;
        c.c_r = 2.0 * rho_new / delta;
        rho = rho_new;
    }
    return c;
}

// One hierarchy across the ranks: a product t = A x is the sum of the holders' shares.  The kernel that USES t takes the
// sum on the fly for the shared dofs (what k_unpack_sum would have written into t first: own share and received shares
// added in ascending rank order, same bits) -- one launch less per exchange.  row_sh[i] = index of dof i among the shared
// dofs of the level's plan or -1; sh_ptr / sh_src as in k_unpack_sum.
struct ShareSum {
    const int32_t *row_sh, *sh_ptr, *sh_src;
    const double *recv;
};
__device__ __forceinline__ double share_sum(const ShareSum &S, int64_t i, double own)
{
    if (!S.row_sh) return own;
    const int32_t j = S.row_sh[i];
    if (j < 0) return own;
    double acc = 0.0;
    for (int32_t k = S.sh_ptr[j]; k < S.sh_ptr[j + 1]; ++k) {
        const int32_t q = S.sh_src[k];
        acc += q < 0 ? own : S.recv[q];
    }
    return acc;
}
// the other end: a shared dof's value goes to every position of the send buffer that names it (the receive positions of
// sh_src ARE the send positions: both buffers have one segment per neighbour with the same lists)
__device__ __forceinline__ void share_pack(const ShareSum &S, double *__restrict__ send, int64_t i, double v)
{
    if (!S.row_sh) return;
    const int32_t j = S.row_sh[i];
    if (j < 0) return;
    for (int32_t k = S.sh_ptr[j]; k < S.sh_ptr[j + 1]; ++k) {
        const int32_t q = S.sh_src[k];
        if (q >= 0) send[q] = v;
    }
}
__global__ void __launch_bounds__(kBlock) k_amg_fill_row_sh(int64_t n_sh, const int32_t *__restrict__ sh_lidx, int32_t *__restrict__ row_sh)
{
    const int64_t j = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (j < n_sh) row_sh[sh_lidx[j]] = static_cast<int32_t>(j);
}

// step 0.  zero_guess: r = b, x = dd = D^-1 b / theta.  Else t = A x on entry: r = b - t (stored when more steps follow),
// dd = D^-1 r / theta, x += dd.
__global__ void __launch_bounds__(kBlock) k_amg_cheb_first(int64_t n, const double *__restrict__ b, const double *__restrict__ t,
                                                            const double *__restrict__ dinv, const double *__restrict__ lam, double ratio,
                                                            double *__restrict__ r_out, double *__restrict__ dd, double *__restrict__ x,
                                                            const CgCtl *ctl, ShareSum S = ShareSum{nullptr, nullptr, nullptr, nullptr})
{
    if (ctl && ctl->flag != 0) return;
    const ChebCoef c = cheb_coef(lam[0], ratio, 0);
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * kBlock) {
        const double ri = t ? b[i] - share_sum(S, i, t[i]) : b[i];
        const double di = c.c_first * dinv[i] * ri;
        if (r_out) r_out[i] = ri;
        if (dd) dd[i] = di;                  // null when no step follows (degree 1): nobody reads it
        x[i] = t ? x[i] + di : di;
    }
}
// step k >= 1.  t = A dd on entry: r -= t, dd = c_dd dd + c_r D^-1 r, x += dd.  `r_in` may be the right-hand side itself
// (zero-guess pre-smoothing: the residual before this step is b); r_out / dd_out null when no step follows.
__global__ void __launch_bounds__(kBlock) k_amg_cheb_next(int64_t n, int step, const double *r_in, const double *__restrict__ t,
                                                           const double *__restrict__ dinv, const double *__restrict__ lam, double ratio,
                                                           double *r_out, const double *dd_in, double *dd_out,      // in / out may be one array
                                                           double *__restrict__ x, const CgCtl *ctl, ShareSum S = ShareSum{nullptr, nullptr, nullptr, nullptr})
{
    if (ctl && ctl->flag != 0) return;
    const ChebCoef c = cheb_coef(lam[0], ratio, step);
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * kBlock) {
        const double ri = r_in[i] - share_sum(S, i, t[i]);
        const double di = __builtin_fma(c.c_dd, dd_in[i], c.c_r * dinv[i] * ri);
        if (r_out) r_out[i] = ri;
        if (dd_out) dd_out[i] = di;
        x[i] += di;
    }
}

// restriction of the residual b - t (t = A x): the right-hand side of the next level, one thread per coarse dof.  The
// member indices of an aggregate (up to 8 from three passes of pairing) are fetched first, then all values: 16 loads in
// flight per lane instead of a chain of dependent pairs; summed in ascending member order either way.
__global__ void __launch_bounds__(kBlock) k_amg_restrict(int64_t nc, const int32_t *__restrict__ mem_ptr, const int32_t *__restrict__ mem_idx,
                                                          const double *__restrict__ b, const double *__restrict__ t /* null: b IS the residual */,
                                                          double *__restrict__ bc, const double *__restrict__ dinv_c, const double *__restrict__ lam_c,
                                                          double ratio, double *__restrict__ dd_c, double *__restrict__ x_c, const CgCtl *ctl,
                                                          int32_t n_own = INT32_MAX, ShareSum S = ShareSum{nullptr, nullptr, nullptr, nullptr},
                                                          double *__restrict__ send = nullptr)
{
    // (coupled hierarchy on several ranks: t is this rank's UN-summed share of A x over all its local dofs and b counts
    // only where the rank owns the dof -- members at or beyond n_own are ghosts --, so that the sum over the ranks of these
    // partial restrictions is the restriction of the assembled residual; one rank: n_own is out of reach, same bits as ever)
    if (ctl && ctl->flag != 0) return;
    const int64_t a = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (a >= nc) return;
    const int q0 = mem_ptr[a], q1 = mem_ptr[a + 1];
    if (n_own != INT32_MAX) {
        double acc = 0.0;
        for (int q = q0; q < q1; ++q) {
            const int i = mem_idx[q];
            acc += (i < n_own ? b[i] : 0.0) - (t ? t[i] : 0.0);
        }
        bc[a] = acc;
        share_pack(S, send, a, acc);         // the coarse vector is what travels (pfem_amg.inc: amg_apply_coupled)
        return;
    }
    double acc = 0.0;
    int q = q0;
    for (; q + 8 <= q1; q += 8) {
        int idx[8];
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) idx[k] = mem_idx[q + k];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = t ? b[idx[k]] - t[idx[k]] : b[idx[k]];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += v[k];
    }
    if (q < q1) {
        int idx[8];
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) idx[k] = q + k < q1 ? mem_idx[q + k] : -1;
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = idx[k] >= 0 ? (t ? b[idx[k]] - t[idx[k]] : b[idx[k]]) : 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (idx[k] >= 0) acc += v[k];
    }
    bc[a] = acc;
    if (dd_c) {          // step 0 of the next level's pre-smoothing (zero guess), what k_amg_cheb_first would do there
        const double di = cheb_coef(lam_c[0], ratio, 0).c_first * dinv_c[a] * acc;
        dd_c[a] = di;
        x_c[a] = di;
    }
}
// coarse-grid correction x += scale * P xc
__global__ void __launch_bounds__(kBlock) k_amg_prolong(int64_t n, const int32_t *__restrict__ agg, const double *__restrict__ xc, double scale,
                                                         double *__restrict__ x, const CgCtl *ctl)
{
    if (ctl && ctl->flag != 0) return;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * kBlock)
        x[i] = __builtin_fma(scale, xc[agg[i]], x[i]);
}

// W-cycle (-pc_mg_cycle_type w): a coarse problem  A_c e = b  gets a second visit.  Between the visits (t = A_c x on entry, x the
// first visit's answer): x is put aside, the right-hand side becomes the residual b - t, and -- where the level's
// pre-smoothing expects its step 0 from the launch above it (fused levels) -- x = dd = D^-1 b / theta of the new right-hand side.
__global__ void __launch_bounds__(kBlock) k_amg_w_between(int64_t n, double *__restrict__ b, const double *__restrict__ t, double *__restrict__ x,
                                                           double *__restrict__ x1, const double *__restrict__ dinv, const double *__restrict__ lam,
                                                           double ratio, double *__restrict__ dd /* null: no step 0 here */, const CgCtl *ctl)
{
    if (ctl && ctl->flag != 0) return;
    const double c_first = dd ? cheb_coef(lam[0], ratio, 0).c_first : 0.0;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * kBlock) {
        const double bi = b[i] - t[i];
        x1[i] = x[i];
        b[i] = bi;
        if (dd) {
            const double di = c_first * dinv[i] * bi;
            dd[i] = di;
            x[i] = di;
        }
    }
}
// ... and after the second: the two answers add up
__global__ void __launch_bounds__(kBlock) k_amg_w_add(int64_t n, double *__restrict__ x, const double *__restrict__ x1, const CgCtl *ctl)
{
    if (ctl && ctl->flag != 0) return;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * kBlock) x[i] += x1[i];
}

// ---------------------------------------------------------------------------
// Coarse levels: the SpMV with the vector step that follows it as its epilogue (row form, one lane per row), so that a
// level of the cycle costs 6 launches instead of 10 -- these levels are too small to fill the chip and every launch has a
// floor of several microseconds.  Same products, same order, same expressions as k_spmv + k_amg_cheb_*: same bits.
//   kEpNextLast: acc = (A dd)_i;  r = r_in - acc;  d = c_dd dd_i + c_r dinv_i r;  x_i = (x_i [+ dd_i]) + d   (last Chebyshev step;
//                add_dd0: step 0 left x untouched because x was the SpMV input of kEpFirstRes)
//   kEpResid:    acc = (A x)_i;   r_i = b_i - acc                                     (residual for the restriction)
//   kEpFirstRes: acc = (A x)_i;   r_i = b_i - acc;  dd_i = c_first dinv_i r_i         (step 0 with a guess; x is added later)
// ---------------------------------------------------------------------------
enum AmgEpilogue { kEpNextLast = 0, kEpResid = 1, kEpFirstRes = 2 };
template <int MODE>
__global__ void __launch_bounds__(kBlock) k_amg_spmv_ep(SellDev A, const double *__restrict__ xin, const double *r_in, const double *__restrict__ dinv,
                                                         const double *__restrict__ lam, double ratio, int step, int add_dd0, double *r_out,
                                                         double *dd_out, double *x, const CgCtl *ctl)
{
    if (ctl && ctl->flag != 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t s = (static_cast<int64_t>(blockIdx.x) << 2) + wave;
    if (s >= A.n_slices) return;
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
        for (int j = 0; j < 4; ++j) { c[j] = cp[64 * (k + j)]; v[j] = vp[64 * (k + j)]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) xv[j] = xin[c[j]];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_fma(v[j], xv[j], acc);
    }
    for (; k < width; ++k) acc = __builtin_fma(vp[64 * k], xin[cp[64 * k]], acc);
    const int64_t i = (s << 6) + lane;
    if (i >= A.n_rows) return;
    if (MODE == kEpResid) {
        r_out[i] = r_in[i] - acc;
    } else if (MODE == kEpFirstRes) {
        const double ri = r_in[i] - acc;
        r_out[i] = ri;
        dd_out[i] = cheb_coef(lam[0], ratio, 0).c_first * dinv[i] * ri;
    } else {
        const ChebCoef c = cheb_coef(lam[0], ratio, step);
        const double ri = r_in[i] - acc;
        const double d0 = xin[i];
        const double di = __builtin_fma(c.c_dd, d0, c.c_r * dinv[i] * ri);
        double xv = x[i];
        if (add_dd0) xv += d0;
        x[i] = xv + di;
    }
}

// the same kernel over 16-bit value codes (pfem_valdict.hpp): v = dict[code], the dictionary in LDS; same products in the same order
template <int MODE>
__global__ void __launch_bounds__(kBlock) k_amg_spmv_ep_vd(SellDev A, const uint16_t *__restrict__ codes, const double *__restrict__ dict, int nd,
                                                            const double *__restrict__ xin, const double *r_in, const double *__restrict__ dinv,
                                                            const double *__restrict__ lam, double ratio, int step, int add_dd0, double *r_out,
                                                            double *dd_out, double *x, const CgCtl *ctl)
{
    extern __shared__ double vd[];
    if (ctl && ctl->flag != 0) return;
    for (int i = threadIdx.x; i < nd; i += kBlock) vd[i] = dict[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t s = (static_cast<int64_t>(blockIdx.x) << 2) + wave;
    if (s >= A.n_slices) return;
    const int64_t off = A.slice_off[s];
    const int width = static_cast<int>((A.slice_off[s + 1] - off) >> 6);
    const int32_t *__restrict__ cp = A.cols + off + lane;
    const uint16_t *__restrict__ qp = codes + off + lane;
    double acc = 0.0;
    int k = 0;
    for (; k + 4 <= width; k += 4) {
        int c[4];
        uint16_t q[4];
        double xv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { c[j] = cp[64 * (k + j)]; q[j] = qp[64 * (k + j)]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) xv[j] = xin[c[j]];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_fma(vd[q[j]], xv[j], acc);
    }
    for (; k < width; ++k) acc = __builtin_fma(vd[qp[64 * k]], xin[cp[64 * k]], acc);
    const int64_t i = (s << 6) + lane;
    if (i >= A.n_rows) return;
    if (MODE == kEpResid) {
        r_out[i] = r_in[i] - acc;
    } else if (MODE == kEpFirstRes) {
        const double ri = r_in[i] - acc;
        r_out[i] = ri;
        dd_out[i] = cheb_coef(lam[0], ratio, 0).c_first * dinv[i] * ri;
    } else {
        const ChebCoef c = cheb_coef(lam[0], ratio, step);
        const double ri = r_in[i] - acc;
        const double d0 = xin[i];
        const double di = __builtin_fma(c.c_dd, d0, c.c_r * dinv[i] * ri);
        double xv = x[i];
        if (add_dd0) xv += d0;
        x[i] = xv + di;
    }
}

// coarse-level product of a hierarchy across the ranks: y = (this rank's share of A) x, the shared rows also into the send buffer
__global__ void __launch_bounds__(kBlock) k_amg_spmv_pack(SellDev A, const double *__restrict__ xin, double *__restrict__ y, ShareSum S,
                                                           double *__restrict__ send, const CgCtl *ctl)
{
    if (ctl && ctl->flag != 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t s = (static_cast<int64_t>(blockIdx.x) << 2) + wave;
    if (s >= A.n_slices) return;
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
        for (int j = 0; j < 4; ++j) { c[j] = cp[64 * (k + j)]; v[j] = vp[64 * (k + j)]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) xv[j] = xin[c[j]];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_fma(v[j], xv[j], acc);
    }
    for (; k < width; ++k) acc = __builtin_fma(vp[64 * k], xin[cp[64 * k]], acc);
    const int64_t i = (s << 6) + lane;
    if (i >= A.n_rows) return;
    y[i] = acc;
    share_pack(S, send, i, acc);
}

// ... the same product over the level's values as 16-bit codes into the dictionary of its distinct ones (amg_value_codes; the
// levels of a hierarchy ACROSS ranks, round 6: the one-rank cycle's fused kernels have had this since round 5): 4 + 2 bytes per
// slot instead of 4 + 8, the same doubles into the same fma chain.  S.row_sh == nullptr: no packing (the product before a restriction).
__global__ void __launch_bounds__(kBlock) k_amg_spmv_pack_vd(SellDev A, const uint16_t *__restrict__ codes, const double *__restrict__ dict, int nd,
                                                              const double *__restrict__ xin, double *__restrict__ y, ShareSum S,
                                                              double *__restrict__ send, const CgCtl *ctl)
{
    extern __shared__ double vd[];
    if (ctl && ctl->flag != 0) return;
    for (int i = threadIdx.x; i < nd; i += kBlock) vd[i] = dict[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t s = (static_cast<int64_t>(blockIdx.x) << 2) + wave;
    if (s >= A.n_slices) return;
    const int64_t off = A.slice_off[s];
    const int width = static_cast<int>((A.slice_off[s + 1] - off) >> 6);
    const int32_t *__restrict__ cp = A.cols + off + lane;
    const uint16_t *__restrict__ qp = codes + off + lane;
    double acc = 0.0;
    int k = 0;
    for (; k + 4 <= width; k += 4) {
        int c[4];
        uint16_t q[4];
        double xv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { c[j] = cp[64 * (k + j)]; q[j] = qp[64 * (k + j)]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) xv[j] = xin[c[j]];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_fma(vd[q[j]], xv[j], acc);
    }
    for (; k < width; ++k) acc = __builtin_fma(vd[qp[64 * k]], xin[cp[64 * k]], acc);
    const int64_t i = (s << 6) + lane;
    if (i >= A.n_rows) return;
    y[i] = acc;
    share_pack(S, send, i, acc);
}

}  // namespace pfem
#include "pfem_amg_rbm.hpp"          // (after cheb_coef, before the tail kernel that uses its transfer)
namespace pfem {

// ---------------------------------------------------------------------------
// The bottom of the cycle in ONE launch: every level of at most kAmgTailRows rows is walked by a single workgroup
// (block-wide barriers instead of kernel boundaries).  At 200^3 that is 3 of the 8 levels (644, 235, 169 rows): 30
// launches of a few microseconds each become one.  Same operations in the same order as the level-by-level kernels:
// same bits.  The limit is low on purpose: one workgroup is latency-bound -- with 4096 rows (a 3350-row level inside)
// the tail took 294 us per cycle, MORE than the 40 launches it replaced (profiles/r03/rocprofv3_kernel_stats_gamg_tail4096.txt).
// ---------------------------------------------------------------------------
constexpr int kAmgTailRows = 1024;
constexpr int64_t kAmgTailNnz = 32768;
constexpr int kAmgTailLevels = 8;
struct AmgTailLevel {
    SellDev A;
    const double *dinv, *lam;
    double *x, *dd, *t, *r, *b;
    const int32_t *agg, *mem_ptr, *mem_idx;
    int64_t n, nc;
    int64_t stored;       // slots of the level's matrix (k_amg_tail stages the first level's matrix in LDS when it fits)
    // transfer with rigid-body modes (pfem_amg_rbm.hpp): mem_* list nodes then
    int rbm_dim, fb;
    int64_t nn;
    const int32_t *node_agg;
    const double *roff;
};
struct AmgTail {
    int nlev, deg, coarsest_deg, dense_n;
    const double *dense_inv;
    double ratio, scale;
    AmgTailLevel lev[kAmgTailLevels];
};

// Round 6: one workgroup has nothing to hide a latency behind, and every operation of the tail was a chain of dependent global
// loads (row -> slice offset -> column -> x): the 343- and 64-row levels of config 3 cost 50 us per cycle, more than level 1 with
// its million rows.  So the tail works out of LDS: every level's vectors (x, dd, t, r, b, dinv: 6 n doubles a level; use_lds) and,
// when it fits beside them, the FIRST level's matrix -- values, columns, slice offsets (mat_slots = its stored slots, else 0) --,
// which the cycle reads four times.  TailView = where a level's data live in this launch.  (The kernel's argument stays
// untouched: a modified copy of it would live in scratch memory, and every field read would be a memory access.)
struct TailView {
    double *x, *dd, *t, *r, *b;
    const double *dinv;
    const int64_t *slice_off;
    const int32_t *cols;
    const double *vals;
};
__device__ inline TailView tail_view(const AmgTail &T, int l, int use_lds, int64_t mat_slots, double *lds)
{
    const AmgTailLevel &L = T.lev[l];
    TailView V{L.x, L.dd, L.t, L.r, L.b, L.dinv, L.A.slice_off, L.A.cols, L.A.vals};
    if (use_lds) {
        int64_t off = 0, tot = 0;
        for (int q = 0; q < T.nlev; ++q) {
            if (q < l) off += 6 * T.lev[q].n;
            tot += 6 * T.lev[q].n;
        }
        double *base = lds + off;
        const int64_t n = L.n;
        V.x = base; V.dd = base + n; V.t = base + 2 * n; V.r = base + 3 * n; V.b = base + 4 * n; V.dinv = base + 5 * n;
        if (l == 0 && mat_slots > 0) {
            double *mv = lds + tot;
            int64_t *so = reinterpret_cast<int64_t *>(mv + mat_slots);
            V.vals = mv;
            V.slice_off = so;
            V.cols = reinterpret_cast<int32_t *>(so + L.A.n_slices + 1);
        }
    }
    return V;
}
__device__ inline void tail_spmv(const TailView &V, int64_t n_rows, const double *x, double *y)
{
    for (int64_t i = threadIdx.x; i < n_rows; i += 1024) {
        const int64_t off = V.slice_off[i >> 6];
        const int width = static_cast<int>((V.slice_off[(i >> 6) + 1] - off) >> 6);
        const int32_t *cp = V.cols + off + (i & 63);
        const double *vp = V.vals + off + (i & 63);
        double acc = 0.0;
        int k = 0;
        for (; k + 8 <= width; k += 8) {          // (columns, values and gathers of eight entries requested together, then the chain in its order)
            int c[8];
            double v[8], xv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { c[j] = cp[64 * (k + j)]; v[j] = vp[64 * (k + j)]; }
#pragma unroll
            for (int j = 0; j < 8; ++j) xv[j] = x[c[j]];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc = __builtin_fma(v[j], xv[j], acc);
        }
        for (; k < width; ++k) acc = __builtin_fma(vp[64 * k], x[cp[64 * k]], acc);
        y[i] = acc;
    }
    __syncthreads();
}
// Chebyshev smoothing exactly as amg_smooth enqueues it (k_amg_cheb_first / k_amg_cheb_next)
__device__ inline void tail_smooth(const AmgTailLevel &L, const TailView &V, const double *b, bool zero_guess, int deg, double ratio, double lmax)
{
    if (!zero_guess) tail_spmv(V, L.n, V.x, V.t);
    const ChebCoef c0 = cheb_coef(lmax, ratio, 0);
    for (int64_t i = threadIdx.x; i < L.n; i += 1024) {
        const double ri = zero_guess ? b[i] : b[i] - V.t[i];
        const double di = c0.c_first * V.dinv[i] * ri;
        if (deg > 1 && !zero_guess) V.r[i] = ri;
        if (deg > 1) V.dd[i] = di;
        V.x[i] = zero_guess ? di : V.x[i] + di;
    }
    __syncthreads();
    const double *r_in = zero_guess ? b : V.r;
    for (int k = 1; k < deg; ++k) {
        tail_spmv(V, L.n, V.dd, V.t);
        const ChebCoef c = cheb_coef(lmax, ratio, k);
        const bool more = k + 1 < deg;
        for (int64_t i = threadIdx.x; i < L.n; i += 1024) {
            const double ri = r_in[i] - V.t[i];
            const double di = __builtin_fma(c.c_dd, V.dd[i], c.c_r * V.dinv[i] * ri);
            if (more) { V.r[i] = ri; V.dd[i] = di; }
            V.x[i] += di;
        }
        __syncthreads();
        r_in = V.r;
    }
}
// transfer with rigid-body modes inside the tail (a tail level has as many dofs per node as the level below it)
__device__ inline void tail_rbm_restrict(const AmgTailLevel &L, const TailView &V, const AmgTailLevel &C, const TailView &VC)
{
    const int64_t nc_nodes = C.n / L.fb;
    for (int64_t a = threadIdx.x; a < nc_nodes; a += 1024) {
        double out[6];
        if (L.rbm_dim == 3) rbm_restrict_node<6, 3>(a, L.mem_ptr, L.mem_idx, L.roff, L.nn, V.b, V.t, L.nn, out);
        else rbm_restrict_node<3, 2>(a, L.mem_ptr, L.mem_idx, L.roff, L.nn, V.b, V.t, L.nn, out);
        for (int k = 0; k < L.fb; ++k) VC.b[L.fb * a + k] = out[k];
    }
}
__device__ inline void tail_rbm_prolong(const AmgTailLevel &L, const TailView &V, const TailView &VC, double scale)
{
    for (int64_t i = threadIdx.x; i < L.nn; i += 1024) {
        if (L.rbm_dim == 3) rbm_prolong_node<6, 3>(i, L.node_agg, L.roff, L.nn, VC.x, scale, V.x);
        else rbm_prolong_node<3, 2>(i, L.node_agg, L.roff, L.nn, VC.x, scale, V.x);
    }
}
// The first level's right-hand side comes from global memory (the restriction above wrote it), its answer goes back there (the
// prolongation above reads it); nothing else of the tail is read by anybody.  Same operations in the same order: same bits.
// use_lds = 0: everything stays where it is (a tail whose vectors do not fit).
__global__ void __launch_bounds__(1024) k_amg_tail(const AmgTail T, const CgCtl *ctl, int use_lds, int64_t mat_slots)
{
    extern __shared__ double tail_lds[];
    if (ctl && ctl->flag != 0) return;
    const int nl = T.nlev;
    __shared__ double lam_of[kAmgTailLevels];    // (every level's eigenvalue bound: requested here, with everything else the tail reads first)
    if (threadIdx.x < static_cast<unsigned>(nl)) lam_of[threadIdx.x] = T.lev[threadIdx.x].lam[0];
    if (!use_lds) __syncthreads();
    if (use_lds) {
        if (mat_slots > 0) {                     // eight slots a thread and trip in flight (the loads do not wait for the LDS stores of the trip before)
            const AmgTailLevel &L = T.lev[0];
            const TailView V = tail_view(T, 0, use_lds, mat_slots, tail_lds);
            double *mv = const_cast<double *>(V.vals);
            int32_t *mc = const_cast<int32_t *>(V.cols);
            int64_t *so = const_cast<int64_t *>(V.slice_off);
            const double *__restrict__ gv = L.A.vals;
            const int32_t *__restrict__ gc = L.A.cols;
            for (int64_t q0 = threadIdx.x; q0 < mat_slots; q0 += 8 * 1024) {
                double v[8];
                int32_t c[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int64_t q = q0 + 1024 * j;
                    v[j] = q < mat_slots ? __builtin_nontemporal_load(gv + q) : 0.0;
                    c[j] = q < mat_slots ? __builtin_nontemporal_load(gc + q) : 0;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int64_t q = q0 + 1024 * j;
                    if (q < mat_slots) { mv[q] = v[j]; mc[q] = c[j]; }
                }
            }
            for (int64_t q = threadIdx.x; q <= L.A.n_slices; q += 1024) so[q] = L.A.slice_off[q];
        }
        for (int l = 0; l < nl; ++l) {
            const AmgTailLevel &L = T.lev[l];
            const TailView V = tail_view(T, l, use_lds, mat_slots, tail_lds);
            double *dv = const_cast<double *>(V.dinv);
            for (int64_t i = threadIdx.x; i < L.n; i += 1024) {
                dv[i] = L.dinv[i];
                if (l == 0) V.b[i] = L.b[i];
            }
        }
        __syncthreads();
    }
    for (int l = 0; l < nl; ++l) {
        const AmgTailLevel &L = T.lev[l];
        const TailView V = tail_view(T, l, use_lds, mat_slots, tail_lds);
        if (l == nl - 1) {
            if (T.dense_n > 0) {
                const int n = T.dense_n;
                for (int i = threadIdx.x; i < n; i += 1024) {
                    double a = 0.0;
                    for (int j = 0; j < n; ++j) a = __builtin_fma(T.dense_inv[j * n + i], V.b[j], a);
                    V.x[i] = a;
                }
                __syncthreads();
            } else {
                tail_smooth(L, V, V.b, true, T.coarsest_deg, T.ratio, lam_of[l]);
            }
            break;
        }
        const AmgTailLevel &C = T.lev[l + 1];
        const TailView VC = tail_view(T, l + 1, use_lds, mat_slots, tail_lds);
        tail_smooth(L, V, V.b, true, T.deg, T.ratio, lam_of[l]);
        tail_spmv(V, L.n, V.x, V.t);
        if (L.rbm_dim) tail_rbm_restrict(L, V, C, VC);
        else
        for (int64_t a = threadIdx.x; a < C.n; a += 1024) {
            double acc = 0.0;
            for (int q = L.mem_ptr[a]; q < L.mem_ptr[a + 1]; ++q) {
                const int i = L.mem_idx[q];
                acc += V.b[i] - V.t[i];
            }
            VC.b[a] = acc;
        }
        __syncthreads();
    }
    for (int l = nl - 2; l >= 0; --l) {
        const AmgTailLevel &L = T.lev[l];
        const TailView V = tail_view(T, l, use_lds, mat_slots, tail_lds);
        const TailView VC = tail_view(T, l + 1, use_lds, mat_slots, tail_lds);
        if (L.rbm_dim) tail_rbm_prolong(L, V, VC, T.scale);
        else
        for (int64_t i = threadIdx.x; i < L.n; i += 1024) V.x[i] = __builtin_fma(T.scale, VC.x[L.agg[i]], V.x[i]);
        __syncthreads();
        tail_smooth(L, V, V.b, false, T.deg, T.ratio, lam_of[l]);
    }
    if (use_lds) {
        const TailView V = tail_view(T, 0, use_lds, mat_slots, tail_lds);
        double *x_out = T.lev[0].x;
        for (int64_t i = threadIdx.x; i < T.lev[0].n; i += 1024) x_out[i] = V.x[i];
    }
}

// ---------------------------------------------------------------------------
// PCG with a stored z = M^-1 r (the V-cycle writes it): PETSc KSPCG semantics as in k_cg_update / k_cg_direction
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_pc_init(int64_t n, const double *__restrict__ b, double *__restrict__ x, double *__restrict__ r)
{
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * kBlock) {
        x[i] = 0.0;
        r[i] = b[i];
    }
}
__global__ void __launch_bounds__(kBlock) k_pc_copy(int64_t n, const double *__restrict__ z, double *__restrict__ p)
{
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * kBlock) p[i] = z[i];
}
// alpha = beta/(p,w); x += alpha p; r -= alpha w.  A breakdown (p,w) <= 0 is handed to the dots kernel through ctl->pad_.
// With z0 set the kernel also does step 0 of the V-cycle's pre-smoothing on the assembled matrix (zero guess:
// z0 = dd0 = D^-1 r / theta over the first n_pc rows, what k_amg_cheb_first would compute from the r written here).
__global__ void __launch_bounds__(kBlock) k_pc_update(CgCtl *ctl, int it, int64_t n, const double *part_pw, int nparts, const double *reduced_pw,
                                                       const double *__restrict__ p, const double *__restrict__ w, double *__restrict__ x,
                                                       double *__restrict__ r, int64_t n_pc, const double *__restrict__ dinv,
                                                       const double *__restrict__ lam, double ratio, double *__restrict__ z0, double *__restrict__ dd0)
{
    __shared__ double sm[4];
    if (ctl->flag != 0) return;
    const double pw = reduced_pw ? *reduced_pw : sum_partials(part_pw, nparts, sm);
    if (!(pw > 0.0)) {                      // KSP_DIVERGED_INDEFINITE_MAT
        if (blockIdx.x == 0 && threadIdx.x == 0) { ctl->its = it; ctl->pad_ = 1; }
        return;
    }
    const double alpha = ctl->beta[it & 1] / pw;
    const double c_first = z0 ? cheb_coef(lam[0], ratio, 0).c_first : 0.0;
    const bool defer_x = x == nullptr;           // (the x update rides on k_pc_post_dots_direction, which streams p anyway: alpha travels in the control block)
    if (defer_x && blockIdx.x == 0 && threadIdx.x == 0) ctl->alpha = alpha;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * kBlock) {
        if (!defer_x) x[i] = __builtin_fma(alpha, p[i], x[i]);
        const double ri = __builtin_fma(-alpha, w[i], r[i]);
        r[i] = ri;
        if (z0 && i < n_pc) {
            const double di = c_first * dinv[i] * ri;
            if (dd0) dd0[i] = di;
            z0[i] = di;
        }
    }
}
// last step of the V-cycle's post-smoothing on the assembled matrix when its degree is 1 (z += D^-1 (r - t) / theta, t = A z)
// together with the CG's (r,z), (z,z) over the owned rows: k_amg_cheb_first + k_pc_dots in one pass, same partial sums
__global__ void __launch_bounds__(kBlock) k_pc_post_dots(const CgCtl *ctl, int64_t n, int64_t n_owned, const double *__restrict__ r,
                                                          const double *__restrict__ t, const double *__restrict__ dinv,
                                                          const double *__restrict__ lam, double ratio, double *__restrict__ z, double *part_rz,
                                                          double *part_zz)
{
    __shared__ double sm[4];
    if (ctl && ctl->flag != 0) return;
    if (ctl && ctl->pad_ != 0) {
        if (threadIdx.x == 0) { part_rz[blockIdx.x] = 0.0; part_zz[blockIdx.x] = -1.0; }
        return;
    }
    const double c_first = cheb_coef(lam[0], ratio, 0).c_first;
    double rz = 0.0, zz = 0.0;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * kBlock) {
        if (i < n_owned) {
            const double ri = r[i];
            const double zi = z[i] + c_first * dinv[i] * (ri - t[i]);
            z[i] = zi;
            rz = __builtin_fma(ri, zi, rz);
            zz = __builtin_fma(zi, zi, zz);
        }
    }
    const double a = block_sum(rz, sm), c = block_sum(zz, sm);
    if (threadIdx.x == 0) { part_rz[blockIdx.x] = a; part_zz[blockIdx.x] = c; }
}
// k_pc_post_dots + k_cg_direction_b + the x update of k_pc_update in ONE launch (one rank; round 6).  The three kernels streamed
// r, t, dinv, z | z, p | p, x: z was written by the first only to be read back by the second, and p was read twice.  Here every
// thread keeps the z of its rows in registers across a grid barrier: phase 1 forms z and the block's (r,z), (z,z) partials;
// all blocks meet (an arrival counter in the control block, bounded spin: a timeout ends the solve with an error instead of
// hanging the device); phase 2 sums the partials -- every block the same sums in the same order --, judges convergence exactly as
// k_cg_direction_b does, and streams p once: x += alpha p (the step k_pc_update left to this launch), p = z + beta p.
// The grid is sized by the caller so that ALL blocks are resident at once (kCoopBlocksPerCu per CU, checked against the
// occupancy the runtime reports, with a block to spare); K = rows a thread owns <= kCoopMaxK.
constexpr int kCoopMaxK = 48;
constexpr int kCoopBlocksPerCu = 3;
constexpr unsigned long long kCoopSpinTicks = 300000000ull;          // 3 s at 100 MHz
__global__ void __launch_bounds__(kBlock, kCoopBlocksPerCu) k_pc_post_dots_direction(CgCtl *ctl, int it, int64_t n, const double *__restrict__ r, const double *__restrict__ t,
                                                                    const double *__restrict__ dinv, const double *__restrict__ lam, double ratio,
                                                                    const double *__restrict__ z_in, double *__restrict__ p, double *__restrict__ x,
                                                                    double *part_rz, double *part_zz, double *hist, int hist_cap, int maxits)
{
    __shared__ double sm[4];
    __shared__ int ok_s;
    if (ctl_finished_before(ctl, it)) return;
    const bool breakdown = ctl->pad_ != 0;                    // (flagged by k_pc_update: (p,Ap) <= 0; x was not advanced)
    const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock, i0 = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    const double c_first = cheb_coef(lam[0], ratio, 0).c_first;
    double zr[kCoopMaxK];
    double rz = 0.0, zz = 0.0;
    if (!breakdown) {
#pragma unroll
        for (int k = 0; k < kCoopMaxK; ++k) {
            const int64_t i = i0 + k * stride;
            zr[k] = 0.0;
            if (i < n) {
                const double ri = r[i];
                const double zi = z_in[i] + c_first * dinv[i] * (ri - t[i]);
                zr[k] = zi;
                rz = __builtin_fma(ri, zi, rz);
                zz = __builtin_fma(zi, zi, zz);
            }
        }
    }
    const double a = block_sum(rz, sm), c = block_sum(zz, sm);
    if (threadIdx.x == 0) {
        part_rz[blockIdx.x] = breakdown ? 0.0 : a;
        part_zz[blockIdx.x] = breakdown ? -1.0 : c;
        // ---- grid barrier: arrival it + 1 of gridDim.x blocks each (the control block is zeroed at the start of a solve)
        __threadfence();
        const unsigned want = static_cast<unsigned>(it + 1) * gridDim.x;
        __hip_atomic_fetch_add(&ctl->gbar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t0 = wall_clock64();
        int ok = 1;
        while (__hip_atomic_load(&ctl->gbar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
            __builtin_amdgcn_s_sleep(2);
            if (wall_clock64() - t0 > kCoopSpinTicks) { ok = 0; break; }
        }
        ok_s = ok;
    }
    __syncthreads();
    if (!ok_s) {                                              // not all blocks were resident: end the solve, do not hang
        if (threadIdx.x == 0) ctl_publish(ctl, -100, it + 1);
        return;
    }
    const int nparts = static_cast<int>(gridDim.x);
    const double srz = sum_partials(part_rz, nparts, sm), szz = sum_partials(part_zz, nparts, sm);
    const bool lead = blockIdx.x == 0 && threadIdx.x == 0;
    if (szz < 0.0) {
        if (lead) ctl_publish(ctl, -10, it + 1);
        return;
    }
    const double rn = sqrt(szz);
    const double beta_old = ctl->beta[it & 1], alpha = ctl->alpha;
    int flag = 0;
    if (rn <= ctl->ttol) flag = 2;
    else if (rn >= ctl->dtol * ctl->rn0) flag = -4;
    else if (srz < 0.0) flag = -8;
    else if (it + 1 >= maxits) flag = -3;
    // (every block has read beta, alpha, ttol ... before the lead overwrites anything a LATER kernel reads only: beta of the other
    // parity, rn, the history, the verdict -- nothing this launch still reads)
    if (lead) {
        ctl->beta[(it + 1) & 1] = srz;
        ctl->rn = rn;
        if (it + 1 < hist_cap) hist[it + 1] = rn;
        ctl_publish(ctl, flag, it + 1);
    }
    const double bb = srz / beta_old;
#pragma unroll
    for (int k = 0; k < kCoopMaxK; ++k) {
        const int64_t i = i0 + k * stride;
        if (i < n) {
            const double pi = p[i];
            x[i] = __builtin_fma(alpha, pi, x[i]);           // the iterate of THIS iteration (also when it is the last)
            if (flag == 0) p[i] = __builtin_fma(bb, pi, zr[k]);
        }
    }
}

// Single-reduction form of the loop (KSPCGUseSingleReduction; k_cg1_step with a STORED z = M^-1 r from the V-cycle): step `it`
// judges iterate `it` from (z,z), then p = z + b p, w = s + b w (s = A z), x += a p, r -= a w with
// b = (r,z)/(r,z)_old, (p,Ap) = (z,s) - b^2 (p,Ap)_old, a = (r,z)/(p,Ap): ONE all-reduce of [(z,s), (r,z), (z,z)] per iteration.
__global__ void __launch_bounds__(kBlock) k_pcg1_step(CgCtl *ctl, int it, int64_t n, const double *reduced /* [(z,s), (r,z), (z,z)] */,
                                                       const double *part_zs, int n_zs, const double *part_rz, const double *part_zz, int nparts,
                                                       const double *__restrict__ z, const double *__restrict__ sv, double *__restrict__ p,
                                                       double *__restrict__ w, double *__restrict__ x, double *__restrict__ r, double rtol,
                                                       double abstol, double dtol, double *hist, int hist_cap, int maxits)
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
    else if (it > 0 && rn <= ctl->ttol) flag = 2;
    else if (it > 0 && rn >= ctl->dtol * ctl->rn0) flag = -4;
    else if (rz < 0.0) flag = -8;
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
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * kBlock) {
        const double zi = z[i], si = sv[i];
        const double pi = it ? __builtin_fma(b, p[i], zi) : zi;
        const double wi = it ? __builtin_fma(b, w[i], si) : si;
        p[i] = pi;
        w[i] = wi;
        x[i] = __builtin_fma(a, pi, x[i]);
        r[i] = __builtin_fma(-a, wi, r[i]);
    }
}

__global__ void __launch_bounds__(kBlock) k_pc_dots(const CgCtl *ctl, int64_t n, int64_t n_owned, const double *__restrict__ r,
                                                     const double *__restrict__ z, double *part_rz, double *part_zz)
{
    __shared__ double sm[4];
    if (ctl && ctl->flag != 0) return;
    if (ctl && ctl->pad_ != 0) {            // breakdown flagged by k_pc_update: k_cg_direction_b turns zz < 0 into -10
        if (threadIdx.x == 0) { part_rz[blockIdx.x] = 0.0; part_zz[blockIdx.x] = -1.0; }
        return;
    }
    double rz = 0.0, zz = 0.0;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * kBlock) {
        if (i < n_owned) {
            const double zi = z[i];
            rz = __builtin_fma(r[i], zi, rz);
            zz = __builtin_fma(zi, zi, zz);
        }
    }
    const double a = block_sum(rz, sm), c = block_sum(zz, sm);
    if (threadIdx.x == 0) { part_rz[blockIdx.x] = a; part_zz[blockIdx.x] = c; }
}

}  // namespace pfem
