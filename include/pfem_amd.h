/*
 * pfem_amd.h -- C ABI of the MI355X-native implicit-FEM hot path that replaces, for
 * PFEMFort's drivers, the per-element stiffness routines, the PETSc Mat/Vec assembly
 * calls and Module_SolverPetsc's KSP solve.
 *
 * Every entry point is extern "C", takes plain pointers and sizes, returns an int
 * error code (PFEM_OK == 0) and never throws, exits or STOPs: the Fortran wrappers
 *  print and STOP on a nonzero code exactly where the reference
 * STOPs / CHKERRQs.  The caller owns all host arrays; the library owns all device
 * memory behind the opaque pfem_solver handle.  A handle is not thread-safe; use
 * one handle per process (= per GPU), as the reference uses one PetscSolver per rank.
 *
 * Citations are file:line of the reference (chennachaos/PFEMFort) interface that
 * each entry point replaces.
 *
 * Array conventions are those of the Fortran drivers (column-major == SoA):
 *   coords(nNode,ndim)         -> xyz  [d*nNode + n]
 *   elemNodeConn(nElem,npElem) -> conn [i*nElem + e]   0-based node ids
 *   ElemDofArray(nElem,nsize)  -> edof [i*nElem + e]   0-based GLOBAL dof ids, -1 = Dirichlet
 *   Klocal(nsize,nsize)        -> K[i + nsize*j]       column-major
 */
#ifndef PFEM_AMD_H
#define PFEM_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PFEM_VERSION 100

/* ---- error codes -------------------------------------------------------- */
#define PFEM_OK 0
#define PFEM_ERR_ARG 1        /* bad argument                                        */
#define PFEM_ERR_STATE 2      /* call out of order (currentStatus, solverpetsc.F:415,441) */
#define PFEM_ERR_NEG_JAC 3    /* STOP " Negative Jacobian ..." elementutilitiespoisson.F:157 */
#define PFEM_ERR_HIP 4        /* a HIP runtime call failed (pfem_last_error_string)  */
#define PFEM_ERR_NOGPU 5      /* no gfx950 device visible: the product has no CPU path */
#define PFEM_ERR_NOMEM 6
#define PFEM_ERR_DIVERGED 7   /* KSPConvergedReason < 0 (solverpetsc.F:481-483)      */
#define PFEM_ERR_PATTERN 8    /* ADD_VALUES into a slot outside the inserted pattern  */
#define PFEM_ERR_COMM 9       /* the communication backend (RCCL / host hooks) failed */

/* ---- element kinds ------------------------------------------------------ */
#define PFEM_POISSON_TRIA 1        /* StiffnessResidualPoissonLinearTria  elementutilitiespoisson.F:23   */
#define PFEM_POISSON_TET 2         /* StiffnessResidualPoissonLinearTetra elementutilitiespoisson.F:107  */
#define PFEM_ELAST_TET 3           /* StiffnessResidualElasticityLinearTetra elementutilitieselasticity3D.F:248 */
#define PFEM_POISSON_TRIA_INLINE 4 /* inline area*B*B^T of triapoissonserialimpl1.F:573-594             */
#define PFEM_ELAST_TRIA 5          /* StiffnessResidualElasticityLinearTria elementutilitieselasticity2D.F:23 (next row 8f.1) */

/* ---- solver status (solverpetsc.F:64-68) -------------------------------- */
#define PFEM_SOLVER_EMPTY 1
#define PFEM_PATTERN_OK 2
#define PFEM_INIT_OK 3
#define PFEM_ASSEMBLY_OK 4
#define PFEM_FACTORISE_OK 5

/* ---- InsertMode (PETSc INSERT_VALUES / ADD_VALUES) ---------------------- */
#define PFEM_INSERT_VALUES 1
#define PFEM_ADD_VALUES 2

typedef struct pfem_solver pfem_solver;

/* ========================================================================= */
/* 0. library / device                                                        */
/* ========================================================================= */
int pfem_version(void);
const char *pfem_strerror(int code);
const char *pfem_last_error_string(void);
/* number of visible HIP devices (0 without a GPU; never an error) */
int pfem_device_count(int *n);
int pfem_device_info(int device, char *name, int name_len, int *compute_units,
                     int64_t *hbm_bytes, int *clock_khz);
/* free / total device memory right now (hipMemGetInfo): what a mesh of a given size leaves of the 288 GB             */
int pfem_device_memory(int device, int64_t *free_bytes, int64_t *total_bytes);

/* ========================================================================= */
/* 1. per-element routines, host, one element per call.                       */
/*    This is the call surface of MODULE ElementUtilitiesPoisson /            */
/*    ElementUtilitiesElasticity3D: the unchanged drivers call them once per  */
/*    element and read Klocal back on the host for the Dirichlet lifting      */
/*    (tetrapoissonparallelimpl1.F:841-870), so by construction they are host */
/*    functions.  They share one source (csrc/pfem_elem.hpp) with the device  */
/*    kernels used by pfem_assemble().                                        */
/* ========================================================================= */
/* elementutilitiespoisson.F:23-101 */
int pfem_poisson_tria_ke(const double xNode[3], const double yNode[3],
                         const double *elemData /*kx,ky*/, const double *timeData /*(2)=af*/,
                         const double valC[3], double K[9], double F[3]);
/* elementutilitiespoisson.F:107-193 */
int pfem_poisson_tet_ke(const double xNode[4], const double yNode[4], const double zNode[4],
                        const double *elemData /*kx,ky,kz*/, const double *timeData,
                        const double valC[4], double K[16], double F[4]);
/* elementutilitieselasticity3D.F:248-393 (intended semantics, DESIGN.md "deviations") */
int pfem_elast_tet_ke(const double xNode[4], const double yNode[4], const double zNode[4],
                      const double *elemData /*E,nu,thick,bx,by,bz*/, const double *timeData,
                      const double valC[12], double K[144], double F[12]);

/* elementutilitieselasticity2D.F:23-153: plane stress, D(3,3) = b1(1-nu) as in the reference */
int pfem_elast_tria_ke(const double xNode[3], const double yNode[3],
                       const double *elemData /*E,nu,thick,bx,by*/, const double *timeData,
                       const double valC[6], double K[36], double F[6]);

/* ========================================================================= */
/* 2. driver bookkeeping, host, integer-exact (tetrapoissonparallelimpl1.F)   */
/* ========================================================================= */
/* genTetra.cpp:152-216,247-323,348-525 -- structured box, 6 tets per hex.
 * kz0/kz1 select the hex layers [kz0,kz1) to emit elements for (0,nEz = all);
 * nodes are always the full grid.  bc_mode 0: u=x^2+y^2+z^2 on all six faces
 * (float coordinates, %.8f text round trip); bc_mode 1: clamp the plane y=y0
 * (all ndof dofs, value 0).  Pass NULL output arrays to query *nDBC only.      */
int pfem_gen_box_tets(double x0, double x1, int nEx, double y0, double y1, int nEy,
                      double z0, double z1, int nEz, int kz0, int kz1, int bc_mode,
                      int ndof, double *xyz, int32_t *conn, int64_t *nDBC,
                      int32_t *bc_node, int32_t *bc_dof, double *bc_val);

/* :316-367 + :393-679.  0-based everywhere.  nParts==1 -> identity maps.
 * node_start/node_end/row_start/row_end have nParts entries, ends exclusive.   */
int pfem_dof_numbering(int64_t nNode, int ndof, int64_t nDBC, const int32_t *dbc_node,
                       const int32_t *dbc_dof, const double *dbc_val, int nParts,
                       const int32_t *node_proc_id, int32_t *node_map_get_old,
                       int32_t *node_map_get_new, int32_t *NodeDofArrayNew,
                       double *solnApplied, int64_t *node_start, int64_t *node_end,
                       int64_t *row_start, int64_t *row_end, int64_t *size_global);
/* The renumbering gathers of the driver: elemNodeConn(e,a) = node_map_get_new(elemNodeConn(e,a)) (:659-664)
 * and the coordinate gather through node_map_get_old (:832-838).  conn arrays are SoA [npElem][nElem],
 * xyz arrays SoA [ndim][nNode]; all ids 0-based.  Threaded (the outputs are first touched in parallel). */
int pfem_renumber_mesh(int64_t nNode, int ndim, int64_t nElem, int npElem, const int32_t *conn_old,
                       const double *xyz_old, const int32_t *node_map_get_new,
                       const int32_t *node_map_get_old, int32_t *conn_new, double *xyz_new);
/* :698-713 */
int pfem_elem_dof_array(int64_t nElem, int npElem, int ndof, const int32_t *conn_new,
                        const int32_t *NodeDofArrayNew, int32_t *edof);
/* :722-734 */
int pfem_assy_for_soln(int64_t nNode, int ndof, const int32_t *NodeDofArrayNew,
                       int32_t *assyForSoln);
/* Stand-in for METIS_PartMeshNodal (:464; METIS is a third-party dependency that is
 * not part of the reference tree): deterministic partition of a pfem_gen_box_tets
 * mesh into nParts slabs of hex layers along z; a node belongs to the lowest part
 * among the elements that touch it.  Any other partitioner's
 * (elem_proc_id,node_proc_id) can be fed to pfem_dof_numbering instead.        */
int pfem_partition_box_slabs(int nEx, int nEy, int nEz, int nParts,
                             int32_t *elem_proc_id, int32_t *node_proc_id);
/* the same along any axis (0 x, 1 y, 2 z; -1: the axis with the most hex layers, ties to z) */
int pfem_partition_box_slabs_axis(int nEx, int nEy, int nEz, int axis, int nParts,
                                  int32_t *elem_proc_id, int32_t *node_proc_id);
/* Recursive coordinate bisection of the element centroids (median splits along the longest axis, any nParts): a
 * geometric stand-in for METIS_PartMeshNodal on meshes WITH coordinates; a node goes to the lowest part among the
 * elements touching it.  conn SoA npElem x nElem, 0-based; xyz SoA ndim x nNode.  (A real METIS partition is taken
 * through its files: PFEM_METIS_PREFIX / read_metis_partition.)                                                     */
int pfem_partition_rcb(int64_t nNode, int ndim, const double *xyz, int64_t nElem, int npElem, const int32_t *conn,
                       int nParts, int32_t *elem_proc_id, int32_t *node_proc_id);

/* Mesh ingest, the step before the path: the three/four `.dat` files are whitespace-
 * separated ASCII tables, one record per line (tetrapoissonparallelimpl1.F:216-355).  `shape` counts
 * the non-empty records and the tokens of the first one; `parse` fills out[c*rows + r] (column-major,
 * like the Fortran arrays), OpenMP over records.  Extra tokens on a line are ignored, as a
 * list-directed READ does.                                                                     */
int pfem_text_table_shape(const char *buf, int64_t len, int64_t *rows, int *cols);
int pfem_text_table_parse(const char *buf, int64_t len, int64_t rows, int cols, double *out);

/* Output step after the path: legacy ASCII VTK exactly as MODULE WriterVTK writes it
 * (writervtk.F:33-201: F12.6 reals, I10 ids, CELL_TYPES 5/10, `procid` cell scalars, scalar or vector
 * `solution` point data).  conn is 0-based SoA, soln is [node*ndof+d] by node id.                  */
int pfem_write_vtk(const char *path, int ndim, int64_t nElem, int64_t nNode, int npElem, int ndof,
                   const double *coords, const int32_t *conn, const int32_t *elem_procid,
                   const double *soln);
/* temp.dat of the drivers: one record per free dof, " ii ind value" (tetrapoissonparallelimpl1.F:935-942) or, with
 * ii = ind = NULL, the value alone (tetraelasticityparallelimpl1.F:1031-1046); values as %.16E.  Formatted on all host
 * threads like pfem_write_vtk.                                                                                      */
int pfem_write_temp_dat(const char *path, int64_t n, const int64_t *ii, const int64_t *ind, const double *val);

/* ========================================================================= */
/* 3. the solver object == TYPE PetscSolver (solverpetsc.F:72-105)            */
/* ========================================================================= */
/* initialise(size_local,size_global,diag_nnz,offdiag_nnz) solverpetsc.F:116-214.
 * row_start = first global row owned by this rank (PETSc derives it from the
 * size_local of the lower ranks; pass 0 on one rank).  diag_nnz/offdiag_nnz are
 * accepted for call compatibility and ignored: the pattern is computed exactly.
 * device < 0 -> current HIP device.  Fails with PFEM_ERR_NOGPU without a GPU.   */
int pfem_solver_create(pfem_solver **s, int64_t size_local, int64_t size_global,
                       int64_t row_start, const int *diag_nnz, const int *offdiag_nnz,
                       int device);
/* free  solverpetsc.F:254-278 */
int pfem_solver_destroy(pfem_solver *s);
/* launch everything on the caller's hipStream_t (e.g. torch's current stream) instead of the
 * solver's own non-blocking stream; a null handle means the legacy default stream */
int pfem_solver_set_stream(pfem_solver *s, void *hip_stream);
/* KSPSetFromOptions / petsc_options.dat stand-in (tetrapoissonparallelimpl1.F:168,
 * solverpetsc.F:198): PETSc defaults are rtol 1e-5, abstol 1e-50, dtol 1e5, maxits 1e4 */
int pfem_solver_set_tolerances(pfem_solver *s, double rtol, double abstol, double dtol, int maxits);
int pfem_solver_status(pfem_solver *s, int *currentStatus);
/* setZero  solverpetsc.F:222-246: finalises the inserted pattern, zeroes values + rhs */
int pfem_solver_set_zero(pfem_solver *s);
/* printInfo solverpetsc.F:286-320 (writes nRow, nnz, storage to stdout) */
int pfem_solver_print_info(pfem_solver *s);

/* MatSetValues(mtx,m,idxm,n,idxn,v,mode) as called at tetrapoissonparallelimpl1.F:798,851:
 * global 0-based indices, negative rows/cols ignored, v read ROW-major (PETSc).
 * INSERT before set_zero records the pattern; ADD after it accumulates on the host
 * staging copy, uploaded by solve (compat path for the unchanged drivers).       */
int pfem_mat_set_values(pfem_solver *s, int m, const int *idxm, int n, const int *idxn,
                        const double *v, int mode);
/* VecSetValues(rhsVec,n,idx,v,mode) :880; negative indices ignored (solverpetsc.F:142) */
int pfem_vec_set_values(pfem_solver *s, int n, const int *idx, const double *v, int mode);
/* assembleMatrix / assembleVector / assembleMatrixAndVector solverpetsc.F:328-401 */
int pfem_solver_assemble_matrix_and_vector(pfem_solver *s, int n, const int *rows,
                                           const int *cols, const double *K /*col-major*/,
                                           const double *F);

/* factorise solverpetsc.F:409-423 (status check only, as in the reference) */
int pfem_solver_factorise(pfem_solver *s);
/* solve solverpetsc.F:431-490: final assembly, zero initial guess, Jacobi-PCG on the GPU.
 * reason follows KSPConvergedReason (2 rtol, 3 atol, -3 its, -4 dtol, -10 indefinite matrix, -8 indefinite PC).
 * Returns PFEM_OK also when reason < 0 ("Divergence." is printed by the caller).   */
int pfem_solver_solve(pfem_solver *s, int *its, int *reason, double *rnorm);
/* factoriseAndSolve solverpetsc.F:498-509 */
int pfem_solver_factorise_and_solve(pfem_solver *s, int *its, int *reason, double *rnorm);
/* VecGetArray on solnVec: the size_local owned entries, device -> host */
int pfem_solver_get_solution(pfem_solver *s, double *x_owned);
/* residual-norm history ||M^-1 r_k||, k = 0..its (up to n entries) */
int pfem_solver_get_history(pfem_solver *s, double *hist, int n, int *n_written);

/* ========================================================================= */
/* 4. batched device path: the element loop of the drivers as ONE call        */
/*    (tetrapoissonparallelimpl1.F:786-884, tetraelasticityparallelimpl1.F:   */
/*    906-965, triapoissonserialimpl1.F:559-650)                              */
/* ========================================================================= */
/* Upload this rank's elements (those with elem_proc_id == rank): connectivity in the
 * NEW numbering, coordinates of all nodes in NEW order, ElemDofArray rows (GLOBAL
 * dof ids; ids outside [row_start,row_start+size_local) become ghost columns/rows
 * of the sub-assembled local matrix), solnApplied by NEW node*ndof+d.            */
int pfem_mesh_upload(pfem_solver *s, int kind, int64_t nElem, const int32_t *conn,
                     int64_t nNode, const double *xyz, const int32_t *edof,
                     const double *solnApplied);
/* Pattern-initialisation loop :786-802 on the device (symbolic phase). */
/* Synthetic configurations without a host mesh (BASELINE configs 2-5): genTetra.cpp's structured box -- node order,
 * "%.8f" coordinates, the 6-tet split of :263-322, Dirichlet data of :505-525 (bc_mode 0: u = x^2+y^2+z^2 on all six
 * faces; 1: the plane y = y0 clamped) -- and the driver's bookkeeping for it (free-dof numbering
 * tetrapoissonparallelimpl1.F:357-367, ElemDofArray :698-713; for nparts > 1 the z-slab partition of
 * pfem_partition_box_slabs, whose renumbering :541-612 is the identity) evaluated ON THE DEVICE for slab `part`.
 * pfem_box_slab_sizes (host, closed forms) gives the sizes to create the solver with; a rank holds the node planes of
 * its own hex layers only.  Replaces pfem_gen_box_tets + pfem_dof_numbering + pfem_renumber_mesh +
 * pfem_elem_dof_array + pfem_mesh_upload for these meshes (bit-identical arrays: tests/test_gpu_parity.py).      */
int pfem_box_slab_sizes(int nEx, int nEy, int nEz, int bc_mode, int ndof, int nparts, int part,
                        int64_t *size_global, int64_t *row_start, int64_t *size_local,
                        int64_t *nNode_local, int64_t *nElem_local);
int pfem_mesh_generate_box(pfem_solver *s, int kind, double x0, double x1, int nEx, double y0, double y1, int nEy,
                           double z0, double z1, int nEz, int bc_mode, int nparts, int part);
/* The same for slabs of hex layers along ANY axis (0 x, 1 y, 2 z; -1: the axis with the most hex layers, ties to z) --
 * what a graph partitioner (METIS_PartMeshNodal, tetraelasticityparallelimpl1.F:522-529) does with a slender body:
 * BASELINE config 4's 50x300x50 beam is cut across its length (37-38 layers per rank on 8 ranks, faces of 51x51 nodes).
 * The reference's renumbering (:541-612: ranks concatenated, ascending old id inside a rank) is no longer the identity
 * for x- or y-slabs; the device evaluates it in closed form, bit-identical to pfem_partition_box_slabs_axis +
 * pfem_dof_numbering + pfem_renumber_mesh + pfem_elem_dof_array on the host (tests/test_gpu_parity.py).
 * axis_used / layer0 / layer1 (may be NULL): the axis taken and the slab's hex layers [layer0, layer1).           */
int pfem_box_slab_sizes_axis(int nEx, int nEy, int nEz, int bc_mode, int ndof, int axis, int nparts, int part,
                             int64_t *size_global, int64_t *row_start, int64_t *size_local,
                             int64_t *nNode_local, int64_t *nElem_local, int *axis_used, int *layer0, int *layer1);
int pfem_mesh_generate_box_axis(pfem_solver *s, int kind, double x0, double x1, int nEx, double y0, double y1, int nEy,
                                double z0, double z1, int nEz, int bc_mode, int axis, int nparts, int part);
/* the mesh as the device holds it; edof in LOCAL numbering (owned rows first, ghosts after); any pointer may be NULL */
int pfem_mesh_download(pfem_solver *s, int32_t *conn, double *xyz, int32_t *edof_local, double *solnApplied);
int pfem_pattern_build(pfem_solver *s);
/* setZero + element loop :817-884 on the device: Ke/Fe, Dirichlet lifting,
 * atomic scatter into the device matrix and rhs.                                */
int pfem_assemble(pfem_solver *s, const double *elemData, const double *timeData);
/* Preconditioner of the CG (PCSetType, solverpetsc.F:206).  JACOBI (default) is the diagonal scaling
 * BASELINE's north_star names.  NODE_BLOCK_JACOBI (PETSc: -pc_type pbjacobi) inverts the
 * diagonal block of every row group of the SpMV (the 1..3 dof rows of a node); it takes effect when the
 * pattern has such groups (3-dof problems) and, on several ranks, when all ranks hold the same groups for
 * the dofs they share (voted at the start of a solve; blocks of shared nodes are summed over the ranks),
 * otherwise JACOBI stays in effect -- pfem_solver_get_preconditioner reports which one the last solve
 * used / the next one will try. */
#define PFEM_PC_JACOBI 0
#define PFEM_PC_NODE_BLOCK_JACOBI 1
/* GAMG (PETSc: -pc_type gamg, reachable from the reference through KSPSetFromOptions / petsc_options.dat,
 * solverpetsc.F:191-206): aggregation algebraic multigrid, one V(1,1) cycle per CG iteration -- aggregates of up to 8 nodes
 * (on a lattice with strong couplings along its axes: bricks of node positions formed in one step; else three passes of
 * pairing along the lattice's axes, or pairwise matching on the strength graph), Galerkin coarse operators re-summed in
 * every solve, Chebyshev smoothing on D^-1 A with a Gershgorin bound (degree 1 on the assembled matrix, 2 below), dense
 * inverse at the coarsest level.  Coarse space: one vector per dof component (piecewise constant) for scalar problems; the
 * rigid-body modes of every aggregate for displacement problems (pfem_solver_amg_transfer).  The reference's own
 * PCBJACOBI/ILU(0) needs 110 / 1 122 iterations on BASELINE configs 3 / 4 and point Jacobi 370 / 5 207; this needs 12 / 18.
 * Aggregates are formed from the values of the first solve after a pattern build and reused while the pattern lives
 * (-pc_gamg_reuse_interpolation true).  Several ranks: ONE hierarchy across the ranks (what PCGAMG does under MPI) --
 * aggregates stay inside a rank's owned dofs, coarse operators are the global Galerkin products held sub-assembled like the
 * matrix, every level has its own neighbour plan, levels small enough are replicated; with PFEM_AMG_COUPLED=0 (or beyond
 * 52 ranks) block Jacobi over the ranks with one hierarchy per rank on its owned diagonal block as assembled from its own
 * elements (PETSc: -pc_type bjacobi -sub_pc_type gamg).                                                                   */
#define PFEM_PC_GAMG 2
int pfem_solver_set_preconditioner(pfem_solver *s, int pc);
int pfem_solver_get_preconditioner(pfem_solver *s, int *pc_in_effect);
/* -pc_gamg knobs: Chebyshev degree on the coarse levels (1..6, default 2) and on the assembled matrix itself (0 = the same;
 * default 1: there an SpMV is dearest), lmax/lmin of the smoothing interval (until set: 16, and 8 for 3-dof nodes WITHOUT rigid-body modes), scaling of the coarse-grid
 * correction (the over-correction a piecewise-constant coarse space wants; until set: 1.5, and 1.8 for 3-dof nodes without rigid-body modes)   */
int pfem_solver_set_amg_options(pfem_solver *s, int cheb_degree, int fine_degree, double eig_ratio, double coarse_scale);
/* (eig_ratio <= 0 / coarse_scale <= 0: that knob stays automatic -- picked per kind of problem at the next symbolic phase) */
/* -pc_mg_cycle_type v|w (PETSc's PCMGSetCycleType behind PCGAMG): 1 = V(1,1), 2 = W(1,1) -- every coarse problem above the
 * single-launch tail of the cycle is visited twice, the second time on the residual of the first --, 0 = the default again
 * (V, as in PETSc).  W roughly halves the iterations on aggregates matched on the strength graph and costs more than it saves
 * on this hardware (DESIGN.md has the numbers).  One hierarchy across several ranks always runs the V-cycle.              */
int pfem_solver_set_amg_cycle(pfem_solver *s, int cycle);
/* KSPCGUseSingleReduction / -ksp_cg_single_reduction (PETSc option of the KSPCG the reference creates, solverpetsc.F:187;
 * off by default there and here): the Chronopoulos-Gear form of the same iteration -- s = A z instead of w = A p,
 * (p,Ap) by recurrence -- so that (z,r), (z,s), (z,z) are reduced together: ONE all-reduce per iteration on several
 * ranks instead of two, at the price of two more vectors of traffic per iteration and one more SpMV per solve.  Point
 * Jacobi, and -pc_type gamg with ONE hierarchy across several ranks (two all-reduces per iteration instead of three: the
 * cycle keeps its own); node-block Jacobi and gamg on one rank keep the two-reduction loop.  on = 1 / 0, or -1: the environment variable
 * PFEM_CG_SINGLE_REDUCTION decides (default off).  Same stopping rule and reasons; iterates agree with the default
 * loop to rounding.                                                                                                    */
int pfem_solver_set_cg_single_reduction(pfem_solver *s, int on);
/* Specified nodal forces after the element loop (VecSetValue(rhsVec,row,fact,ADD_VALUES),
 * tetraelasticityparallelimpl1.F:971-982) for the batched path: GLOBAL free-dof ids (i.e.
 * NodeDofArrayNew(n,d)-1; the reference's own row formula ignores constrained dofs);
 * ids this rank does not own and negative ids are skipped.                                      */
int pfem_rhs_add_values(pfem_solver *s, int64_t n, const int64_t *global_dof, const double *values);

/* ---- inspection / export (device -> host) -------------------------------- */
/* global dof id of every local row (owned first, then ghosts ascending) */
int pfem_get_local_to_global(pfem_solver *s, int64_t *gid);
/* CSR of the LOCAL (sub-assembled) matrix, local column ids, columns ascending */
int pfem_get_csr(pfem_solver *s, int64_t *rowptr, int32_t *cols, double *vals);
int pfem_get_rhs(pfem_solver *s, double *rhs_local);

/* ========================================================================= */
/* 5. multi-GPU: one process per GPU, sub-assembled interface rows            */
/*    (replaces MatAssembly stash + VecScatter + VecDot MPI_Allreduce inside   */
/*    KSPSolve, solverpetsc.F:447-476)                                        */
/* ========================================================================= */
/* Every rank assembles only its own elements (elem_proc_id == rank, tetrapoissonparallelimpl1.F:829) into a local
 * matrix over owned + ghost rows; rows of dofs that another rank also touches hold partial sums.  Per CG iteration:
 *   1. w = A_loc p on the slices that hold shared rows, then  pack -> NEIGHBOUR exchange  on a second stream while
 *      the interior slices run; every rank adds the partials of a shared dof in ascending rank order (same bits on
 *      all ranks, so the replicated ghost entries of the vectors never drift);
 *   2. two scalar all-reduces: (p, A p) and [(r,z), (z,z)].
 * Who shares which dofs with whom is the NEIGHBOUR PLAN (pure integer host logic):                                */

/* Neighbour plan of `rank`: dof g is shared with rank q when both have it in their local numbering (owned block or
 * ghost list).  Inputs: every rank's owned block [row_start[r], row_end[r]) and ghost list (ascending global ids,
 * concatenated; ghost_off[r]..ghost_off[r+1]).  Two-call: with peers == NULL only *n_peers and *n_total are set.
 * Outputs: peers[] ascending, peer_off[n_peers+1], shared_gid[] = for each peer the ascending global ids shared with
 * it (both sides of a pair compute the same list).  Host only, no GPU needed.                                      */
int pfem_neighbour_plan(int nranks, int rank, const int64_t *row_start, const int64_t *row_end,
                        const int64_t *ghost_off, const int64_t *ghost_gid, int *n_peers, int64_t *n_total,
                        int *peers, int64_t *peer_off, int64_t *shared_gid);
/* install the plan (after pfem_mesh_upload, or after the first setZero of the compat path: the local numbering
 * must exist; and after the communication backend, which tells the solver its rank).  n_peers == 0 is legal (a rank that shares nothing still takes part in the scalar all-reduces).   */
int pfem_solver_set_neighbours(pfem_solver *s, int n_peers, const int *peers, const int64_t *peer_off,
                               const int64_t *shared_gid);
/* Communication backend 1 -- RCCL over xGMI, bound inside the library (librccl is loaded at run time; nothing else
 * needs it): rank 0 creates the unique ids (256 bytes), the host program broadcasts them by whatever means it has
 * (torch.distributed store, MPI_Bcast, a file) and every rank calls pfem_solver_set_comm_rccl.  The exchange is one
 * grouped ncclSend/ncclRecv per neighbour on the solver's communication stream (under the interior SpMV); the
 * reductions are ncclAllReduce on a second communicator, in order on the compute stream.                          */
#define PFEM_RCCL_ID_BYTES 256     /* two ncclUniqueIds: one communicator for the exchange, one for the all-reduces */
int pfem_rccl_unique_id(void *id_out);
int pfem_solver_set_comm_rccl(pfem_solver *s, int rank, int nranks, const void *id);
/* Communication backend 2 -- host hooks, for hosts whose transport works on HOST memory (MPI: the reference's own
 * transport, pfemfort_amd/fortran/pfem_mpi.cpp; gloo in the tests where several ranks share one GPU).  The library
 * stages the packed buffers through pinned host memory and calls:
 *   allreduce(ctx, buf, count)         in-place SUM of `count` doubles; EVERY rank must receive the same bits
 *   exchange(ctx, n_peers, peers, off, send, recv)   send[off[k]..off[k+1]) goes to peers[k], the same range of
 *                                      recv is filled with what peers[k] sent (symmetric counts)
 * Return 0 on success.                                                                                             */
typedef int (*pfem_host_allreduce_fn)(void *ctx, double *buf, int64_t count);
typedef int (*pfem_host_exchange_fn)(void *ctx, int n_peers, const int *peers, const int64_t *off,
                                     const double *send, double *recv);
int pfem_solver_set_comm_host(pfem_solver *s, int rank, int nranks, pfem_host_allreduce_fn allreduce,
                              pfem_host_exchange_fn exchange, void *ctx);
/* Peer memory: the ranks map each other's receive boxes (hipIpcGetMemHandle / hipIpcOpenMemHandle) and their kernels write a
 * neighbour's segment straight into its box, one release / acquire flag per (rank, neighbour) pair; small all-reduces the same
 * way (every rank sums the boxes in rank order: identical bits).  Replaces the VecScatter / MPI_Allreduce inside KSPSolve
 * (solverpetsc.F:476) without a collective library and without the host in the data path.  Works between processes that SHARE
 * one device -- where RCCL refuses a second rank -- and is what the ranks-on-one-GPU tests run the device path with; between
 * devices it needs peer-mapped memory the runtime keeps coherent (not available to the builder: RCCL stays the default for
 * one rank per GPU).  The host hooks carry the bring-up (memory handles through `allreduce`), the teardown barrier and
 * all-reduces of more than 65536 doubles (symbolic phases); `exchange` is unused and may be NULL.  At most 16 ranks;
 * a neighbour's segment may hold PFEM_PEER_CAP_DOUBLES doubles (default 2^20).  Waits are bounded (10 s): a lost rank
 * surfaces as PFEM_ERR_COMM after the solve instead of a hung device.                                                    */
int pfem_solver_set_comm_peer(pfem_solver *s, int rank, int nranks, pfem_host_allreduce_fn allreduce,
                              pfem_host_exchange_fn exchange, void *ctx);
/* Collective teardown step of the communication backend, for all ranks while they can still reach each other (the host mirror's
 * free() calls it): the peer-memory transport waits here until no rank may still write into another's region.  Destroying a
 * solver without it is safe but not synchronised (the destructor is deliberately NOT collective).  Idempotent; PFEM_OK without a
 * backend.  Reference: MPI_Finalize-time destruction of the VecScatter inside KSPDestroy (solverpetsc.F:254-320). */
int pfem_solver_comm_shutdown(pfem_solver *s);
/* host-only helper (no GPU needed): ascending unique global dof ids in edof[0..count) that
 * lie outside the owned block [row_start,row_start+n_owned); two-call (NULL -> count).  */
int pfem_find_ghosts(int64_t count, const int32_t *edof, int64_t row_start, int64_t n_owned,
                     int64_t *n_ghost, int64_t *ghost_gid);
/* ghost dof ids (ascending) of the uploaded mesh: two-call (ghost_gid NULL -> count) */
int pfem_get_ghosts(pfem_solver *s, int64_t *n_ghost, int64_t *ghost_gid);

#ifdef __cplusplus
}
#endif
#endif /* PFEM_AMD_H */
