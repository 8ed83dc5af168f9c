// pfem_amg_types.hpp -- data of the aggregation-multigrid preconditioner (see pfem_amg.inc); included by pfem_device.hip
// after DevBuf and before the solver object, which owns one hierarchy.
#pragma once

// Neighbour plan of one level of a rank-coupled hierarchy (same layout as the solver's own plan for level 0: send list
// by neighbour, and per distinct shared dof the receive positions of the other holders' partials in rank order)
struct AmgPlan {
    std::vector<int> peers;
    std::vector<int64_t> peer_off{0};
    std::vector<int32_t> send_lidx;       // host copy: the next level's plan is derived from it
    int64_t n_send = 0, n_sh = 0;
    DevBuf<int32_t> d_send_lidx, d_sh_lidx, d_sh_ptr, d_sh_src;
    DevBuf<int32_t> d_row_sh;             // [n_loc] index of every local dof among the plan's shared dofs, -1: not shared (fused pack / unpack)
};

struct AmgLevel {
    int64_t n = 0, n_slices = 0, stored = 0, nnz = 0;
    // several ranks, coupled hierarchy: the level's matrix has n_loc >= n rows -- the n dofs this rank owns (the domain of
    // its aggregation) and after them the dofs of neighbours' aggregates that its own fine dofs belong to (ghosts); the
    // matrix holds this rank's share of every entry (sub-assembled like level 0).  One rank / block mode: n_loc == n.
    int64_t n_loc = 0;
    int64_t gid_off = 0;                  // global number of this rank's first owned dof of the level
    std::vector<int64_t> ghost_gid;       // ascending global numbers of the ghosts
    AmgPlan plan;                         // levels >= 1 (level 0 uses the solver's)
    DevBuf<int32_t> gid;                  // last level with a global dense inverse: global number of every local dof
    bool fine = false;                    // level 0: the solver's own matrix (and its SpMV forms)
    // matrix of a coarse level: wave-sliced CSR like the fine one, int32 columns
    DevBuf<int64_t> slice_off, rowptr;
    DevBuf<int32_t> rowlen, cols;
    DevBuf<double> vals;
    // dofs grouped in nodes (elasticity: 3 per node): null = every dof its own node
    int bs = 1;
    int64_t n_nodes = 0;
    DevBuf<int32_t> node_of, comp_of;
    DevBuf<int32_t> hint;                 // [n_nodes] place of every node along a space-filling curve (empty: the index), for the pairing
    bool lattice = false;                 // hint = lattice position (x | y << 10 | z << 20): the pairing goes axis by axis (k_amg_lat_*)
    int lat_hi[3] = {0, 0, 0};            // highest position along every axis on this level
    int lat_axis = 0;                     // axis of this level's first pass
    // bricks across ranks (amg_bricks_level, coupled): hint holds the GLOBAL positions of ALL local nodes (owned, then ghosts),
    // the same numbers on every rank, padded so that every plane where the nodes' owner changes (lat_cuts: first position of
    // every stretch of one owner, per axis, ascending, [0] = 0) is a multiple of the level's brick size
    bool lat_global = false;
    std::vector<int> lat_cuts[3];
    std::vector<char> lat_single[3];      // per stretch: the last odd position of the stretch went to its start (lat_pad alternates)
    int lat_shift[3] = {0, 0, 0};         // the halvings per axis the padding of this level was made for
    int lat_occ[6] = {0, 0, 0, 0, 0, 0};  // lowest / highest OCCUPIED position per axis over all ranks (a Dirichlet plane has a position and no dofs)
    int lat_box[6] = {0, 0, 0, 0, 0, 0};  // box of the local nodes' positions: lowest x, y, z, highest x, y, z
    // node bricks in one step (rigid-body levels, amg_node_bricks): the box of positions the level's nodes fill completely
    bool lat_full = false;
    int lat_tlo[3] = {0, 0, 0}, lat_thi[3] = {0, 0, 0};
    // ... across ranks (round 6): hint = GLOBAL positions of the owned nodes, the same numbers on every rank; lat_str = per axis
    // the occupied stretches of positions that one owner holds (inclusive ends, ascending; from all-reduced boxes: the same on every
    // rank); lat_tlo / lat_thi = the box THIS rank's owned nodes fill.  Bricks are cut at the stretches' ends (node_brick_axis).
    bool lat_nodes_global = false;
    std::vector<std::pair<int, int>> lat_str[3];
    // scalar bricks across ranks whose owned dofs do NOT fill boxes (a METIS-like partition; round 6): hint = the GLOBAL positions,
    // unpadded; an aggregate = the part of a brick that one rank owns (amg_split_bricks); the Galerkin maps come from the sorted
    // keys like those of any other aggregates.  lat_tlo / lat_thi = the box of this rank's OWNED positions.
    bool lat_split = false;
    std::vector<double> lat_coord;        // [3 x 1024] coordinate of every position (a brick sits at its lowest corner): the corners a level
                                          // needs when it leaves the brick path (xyz below) come from here instead of travelling down the levels
    DevBuf<double> xyz;                   // coupled hierarchy with a lattice: [3 x n_nodes] a corner of every node's aggregate (see k_amg_xyz_min)
    // how this level's aggregates were formed (pfem_solver_amg_aggregation): 0 none (last level), 1 bricks in one step, 2 node bricks
    // in one step, 3 bricks split between their owners, 4 pairing passes on the lattice, 5 matching on the strength graph,
    // 6 roots + neighbours (independent set)
    int agg_kind = 0;
    // transfer to the next level (piecewise-constant prolongation)
    int64_t nc = 0;
    DevBuf<int32_t> agg;                  // [n]  coarse dof of every dof
    DevBuf<int32_t> mem_ptr, mem_idx;     // [nc+1], [n]  members of every coarse dof, ascending
    // numeric Galerkin product into the next level
    int64_t nnz_c = 0;
    DevBuf<int64_t> src_ptr;              // [nnz_c+1]
    DevBuf<int32_t> src_slot;             // storage slots of this level's matrix, grouped by coarse entry
    DevBuf<int64_t> dst_slot;             // [nnz_c] storage slot in the next level's matrix
    // brick level (k_lat_codes_*): the product by coarse row instead (k_lat_galerkin) -- offset code of every stored entry
    // of this level's matrix and the occupied codes of every coarse row
    DevBuf<uint8_t> code_of;              // [stored]
    DevBuf<uint32_t> code_mask;           // [nc]
    // rigid-body-mode coarse space (pfem_amg_rbm.hpp): the level's dofs come `bs` to the node (block regular), the next level
    // has dim + (dim == 3 ? 3 : 1) per aggregate.  With rbm set, mem_ptr / mem_idx list the member NODES of every coarse node
    // and src_ptr / src_slot the fine node BLOCKS of every coarse node block; agg keeps the translation part of P.
    int dim = 0;                          // space dimension when the level may take the rigid-body transfer (0: not a candidate)
    bool rbm = false;                     // the transfer to the next level carries rotations
    DevBuf<double> cen;                   // [3 x n_nodes] node coordinates (level 0: mesh nodes; below: centroids of the aggregates)
    DevBuf<double> roff;                  // [3 x n_nodes] offset of every node from the centroid of its aggregate
    DevBuf<int32_t> node_agg;             // [n_nodes] aggregate (coarse node) of every node
    DevBuf<int64_t> gptr;                 // [n_nodes + 1] node graph = block pattern of the level's matrix ...
    DevBuf<int32_t> gcol, brow;           // ... column node and row node of every block
    int64_t nblk = 0;
    // smoother and work vectors (x and dd are SpMV inputs: on level 0 they carry the guard bands of the fast SpMV forms)
    DevBuf<double> dinv, t, r, b, x_store, dd_store, lam, part_max;
    // values as 16-bit codes into a dictionary of the distinct ones (pfem_valdict.hpp; scalar coarse levels of a one-rank hierarchy
    // whose Galerkin sums repeat -- bricks of a lattice): the fused SpMVs of the cycle stream 4 + 2 B a slot instead of 4 + 8
    DevBuf<uint16_t> vcodes;
    DevBuf<double> vdict;
    DevBuf<unsigned long long> vtable;
    int vd_n = 0;
    bool vd_ok = false, vd_have_dict = false, vd_refused = false;
    // round 6: the Galerkin product that forms this level's values also writes their codes (k_lat_galerkin through the hash table
    // of the dictionary, pfem_vdhash.hpp) and leaves the inverse diagonal + the rows' Gershgorin ratios behind
    DevBuf<VdHashEntry> vhash;
    bool vd_hash_ok = false;              // vhash belongs to vdict, and every slot of vcodes carries a code of it
    bool vd_direct = false;               // this solve's codes came from the Galerkin product: verdict in Amg::vd_states, not read yet
    bool bound_fresh = false;             // dinv and t (ratios) of this level were written by the product that formed its values
    double *x = nullptr, *dd = nullptr;
    DevBuf<double> x1;                    // W-cycle: the first visit's answer while the second is on its way
    double lam_host = 0.0;
};

struct Amg {
    std::vector<std::unique_ptr<AmgLevel>> lev;      // lev[0] = fine ... lev.back() = coarsest
    DevBuf<double> dense_inv;                        // inverse of the coarsest operator (n <= kAmgDense)
    bool dense = false;
    bool symbolic_ok = false;
    bool galerkin_fresh = false;                     // the symbolic phase has just formed every coarse operator from the CURRENT values (it needs them level
                                                     // by level): the numeric phase of the same solve does not form them again
    double symbolic_ms = 0.0, numeric_ms = 0.0;
    int cheb_degree = 2;
    int fine_degree = 1;                             // Chebyshev degree on level 0 (0: cheb_degree), the level where an SpMV is dearest:
                                                     // measured 200^3 52 -> 41 ms, beam 321 -> 279 ms against degree 2 everywhere
    bool eig_ratio_given = false;                    // likewise the smoothing interval: until set, lmax/16 .. lmax on scalar problems (with the lattice's
                                                     // hierarchy 3-7 % faster than /8 at 100^3, 160^3, 200^3), lmax/8 with 3 dofs per node (the beam: /16 costs 12 %)
    bool coarse_scale_given = false;                 // set through pfem_solver_set_amg_options / PFEM_AMG_COARSE_SCALE; else 1.5, 1.8 for 3-dof nodes
    double eig_ratio = 8.0, coarse_scale = 1.5;       // over-correction of the piecewise-constant coarse space (Braess 1995 takes 1.8).
                                                     // measured 1.5 -> 1.8: 200^3 equal, the beam 228 -> 189 ms, the 9^3 tet10 mesh WORSE
                                                     // (26 against 22 iterations): 1.5 is the robust middle for scalar problems; with
                                                     // 3 dofs per node 1.8 is the default (beam 205 -> 169 iterations, half-size beam 191 -> 155)
    // the V-cycle as a hipGraph (one rank): ~100 dependent launches, most of them on levels too small to fill the chip
    hipGraphExec_t graph = nullptr;
    std::vector<uint64_t> graph_key;
    bool graph_off = false;
    hipEvent_t ev_num0 = nullptr, ev_num1 = nullptr;   // around the numeric phase of a solve (created once)
    ~Amg()
    {
        if (graph) (void)hipGraphExecDestroy(graph);
        if (ev_num0) (void)hipEventDestroy(ev_num0);
        if (ev_num1) (void)hipEventDestroy(ev_num1);
    }
    int tail_from = -1;                              // first level of the single-launch tail of the cycle (-1: none)
    size_t tail_lds_allowed = 0;                     // dynamic LDS bytes k_amg_tail has been allowed beyond 64 KB (hipFuncSetAttribute, once)
    // -pc_mg_cycle_type: 1 = V (default), 2 = W -- the coarse problem of every level from 1 down to w_to is visited twice (second
    // visit on the residual of the first).  The levels of the single-launch tail (<= 1024 rows each) stay a V inside:
    // w_to = tail_from, whether or not the fused kernels are in use (amg_cycle_shape has the measurements).
    int cycle_gamma = 1, w_to = -1;
    bool cycle_given = false;
    bool fused = true;                               // fused SpMV epilogues on the coarse levels + the tail kernel (PFEM_AMG_FUSED=0: off)
    int coarsest_sweeps = 8;                         // Chebyshev degree on the last level when it is too large for the dense inverse
    // several ranks: one hierarchy ACROSS the ranks (aggregates stay inside a rank's owned dofs, the operators are the
    // global Galerkin products held sub-assembled, every SpMV of the cycle is followed by the level's neighbour exchange)
    // instead of one hierarchy per rank (block Jacobi).
    bool coupled = false;
    bool coupled_fused = true;                       // pack as the coarse SpMV's / the restriction's epilogue, unpack-sum inside the vector step (PFEM_AMG_COUPLED_FUSED=0: separate kernels)
    bool coupled_refused = false;                    // the ranks could not form it (decided once per pattern, by all of them together)
    int64_t n_last_global = 0;                       // rows of the last level over all ranks
    DevBuf<double> dense_glob, bx_glob, lam_all;     // coupled: assembled last-level operator, its right-hand side / solution, bounds of all ranks
    // coupled: from the level where the whole problem is small (kAmgReplicateRows rows over all ranks) every rank holds the
    // ASSEMBLED level and carries the rest of the hierarchy alone -- the same on all ranks, no exchange below that point, the
    // fused kernels of the one-rank cycle, aggregates that cross the ranks' borders.  lev.back() is that level in its
    // distributed form (its sub-assembled matrix feeds rep->lev[0] in every solve), rep->lev[0] the same level assembled.
    std::unique_ptr<Amg> rep;
    DevBuf<double> rep_buf;                          // all ranks' entries of the level, rank after rank (rows | cols at set-up, values in a solve)
    DevBuf<int32_t> rep_pack_slot;                   // this rank's entries: storage slot of each, row by row
    int64_t rep_total = 0, rep_off = 0, rep_mine = 0;
    bool rbm = false;                                // some level carries rigid-body modes: eigenvalue bounds on the symmetrically scaled operators
    DevBuf<double> lam2;                             // ... scratch of that second bound
    DevBuf<VdState> vd_states;                       // one verdict per level (amg_value_codes)
    int cycle_exchanges = 0, cycle_allreduces = 0;   // coupled: neighbour exchanges / all-reduces one V-cycle enqueues (counted by the last cycle)
};
