/* pfem_amd_diag.h -- introspection, measurement and lab knobs of libpfem_amd.so.
 *
 * NOT part of the drop-in boundary: nothing here replaces an interface of the reference (chennachaos/PFEMFort).  include/pfem_amd.h
 * is what a maintainer binds (INTEGRATION.md section 1 maps each of its entry points to the reference interface it replaces); this
 * header is what the build's own tests, bench.py and lab probes use to look inside the solver: which SpMV form and assembly
 * formulation a pattern got, the multigrid hierarchy of the last solve (levels, aggregates, transfer, layout across ranks, one
 * instrumented cycle), device-evaluated element matrices for the parity tests, event timings, and the transport's self-test and
 * latency probes.  Same conventions: extern "C", plain pointers and sizes, int return codes.                                      */
#ifndef PFEM_AMD_DIAG_H
#define PFEM_AMD_DIAG_H

#include "pfem_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Numeric-assembly formulation used by pfem_assemble:
 *   GATHER  (default) one thread per node walks its incident elements in ascending element
 *           order and owns its matrix rows: no atomics, K and F bit-identical to the serial
 *           reference loop, run-to-run deterministic;
 *   SCATTER one thread per element, hardware f64 atomicAdd into the matrix (sum order varies). */
#define PFEM_ASSEMBLY_GATHER 0
#define PFEM_ASSEMBLY_SCATTER 1
int pfem_solver_set_assembly_mode(pfem_solver *s, int mode);
/* what pfem_assemble does with the current pattern: gather form in effect (1/0); nodes whose rows are too long for the
 * gather records (> 255 entries: "hubs" -- their rows alone are assembled by a scatter pass with atomics, every other
 * row keeps one writer); threads and LDS bytes per block of the row-accumulating gather kernels                      */
int pfem_solver_assembly_info(pfem_solver *s, int *gather_form, int *hub_nodes, int *block_threads, int64_t *lds_bytes);
/* Matrix encoding streamed by the SpMV.  AUTO uses 16-bit gaps between the ascending columns of a
 * row (4 + 2 B per entry instead of 4 B) whenever every gap of the pattern fits, and on top of that
 * serves consecutive rows with identical column sets (the dof rows of a node) from one lane with a
 * shared column stream when the pattern has, on average, at least 2.75 such rows per group of 3;
 * or (scalar problems) serves 4 consecutive rows from one lane with the union of their columns
 * relative to the row -- both only when the system is large enough to keep the chip full with
 * 3-4x fewer waves (>= 327 680 groups); GROUPED takes the group forms at any size, GAPS16 forces
 * the plain row form with 16-bit gaps, INT32 plain int32 columns.  Every row sums the same
 * products in the same order: y is bit-identical in all of them. */
#define PFEM_SPMV_AUTO 0
#define PFEM_SPMV_INT32 1
#define PFEM_SPMV_GAPS16 2
#define PFEM_SPMV_GROUPED 3   /* the row-group forms whenever the pattern has them, however small the system */
int pfem_solver_set_spmv_format(pfem_solver *s, int format);
/* 16 if the SpMV currently streams 16-bit column gaps, 32 for int32 columns */
int pfem_solver_get_spmv_format(pfem_solver *s, int *bits_per_column);
/* rows served by one lane of the current SpMV: 3 in the row-grouped form, else 1 */
int pfem_solver_get_spmv_row_group(pfem_solver *s, int *rows_per_lane);
/* Relative row groups whose pattern has gaps beyond 65535 (planes of more than 65 535 nodes): number of entries of the
 * table of distinct large gaps when the 16-bit DICTIONARY form is in use (codes >= 0x8000 index the table; at most
 * 256 distinct gaps of 32768 and more), 0 otherwise (literal 16-bit gaps, 32-bit gaps, or another form).            */
int pfem_solver_get_spmv_gap_table(pfem_solver *s, int *entries);
/* Value dictionary of the SpMV's group forms (pfem_valdict.hpp): when the assembled matrix holds at most 4096 distinct values
 * (structured meshes: element matrices repeat) the SpMV streams 16-bit codes into a dictionary held in LDS instead of the
 * doubles -- the same doubles, the same products, the same bits, 2 B a slot instead of 8.  *entries = the dictionary's size
 * after the last solve / product, 0 when the fp64 copy is streamed (PFEM_SPMV_VALDICT=0 turns the form off).               */
int pfem_solver_get_spmv_value_dictionary(pfem_solver *s, int *entries);
/* ... and per level of the last gamg hierarchy (scalar coarse levels of >= 2^20 slots on one rank take the form too: the Galerkin
 * sums over the bricks of a lattice repeat like the element matrices do): entries[l] = the level's dictionary size, 0 = fp64 values */
int pfem_solver_amg_value_dictionaries(pfem_solver *s, int max_levels, int *n_levels, int *entries);
/* 1 when the row form streams 16-bit gaps WITH ESCAPES (k_spmv16e: the code 0xffff sends a column to the matrix's int32 column
 * array -- numberings whose far neighbours are too many and too irregular for the table: partition-renumbered and
 * curve-ordered meshes; taken when at most a quarter of the entries escape), else 0.  Same bits as every other form.     */
int pfem_solver_get_spmv_gap_escapes(pfem_solver *s, int *in_use);
/* bytes one launch of the selected SpMV form moves at best: its own storage (values, gap words / columns, offsets)
 * + x + y, each touched once.  (The judged figure 12 nnz + 20 N of SURVEY 8d is the plain int32-CSR equivalent.)   */
int pfem_solver_spmv_bytes(pfem_solver *s, int64_t *format_bytes);

/* GAMG hierarchy of the last solve: number of levels, rows / nonzeros / eigenvalue bound of each (arrays of max_levels),
 * time of the symbolic phase (once per pattern) and of the numeric phase of the last solve (inside its timer), the knobs */
int pfem_solver_amg_info(pfem_solver *s, int max_levels, int *n_levels, int64_t *rows, int64_t *nnz, double *lambda_max,
                         double *symbolic_ms, double *numeric_ms, int *cheb_degree, int *fine_degree, double *eig_ratio, double *coarse_scale);
/* coarse dof of every dof of `level` (0 = the assembled matrix); what the oracle's restatement of the cycle is given */
int pfem_solver_amg_aggregates(pfem_solver *s, int level, int32_t *agg);
/* Displacement problems (as many dofs per node as space dimensions: the tetra / tria elasticity kinds), mesh on the
 * device: the coarse space carries the RIGID-BODY MODES of every aggregate (PETSc: MatSetNearNullSpace / PCSetCoordinates
 * ahead of PCGAMG; tetraelasticityparallelimpl1.F:894-902, 993) -- dim translations and 3 (plane: 1) rotations about the
 * aggregate's centroid, so a coarse node has 6 (3) dofs; the beam of BASELINE config 4 needs 18 iterations instead of 169.
 * What the transfer from `level` to the next one looks like (one rank, and the hierarchy across several): *rbm = 1 when it
 * carries rotations, dofs per node on this level and the next, the space dimension, and (optional, [3 x n_nodes] as x | y | z) the coordinates of this level's
 * nodes (level 0: the mesh nodes in dof order; below: the centroids of the aggregates).  With *rbm = 1
 * pfem_solver_amg_aggregates reports the TRANSLATION part: dof c of node i belongs to coarse dof coarse_bs * aggregate(i) + c. */
int pfem_solver_amg_transfer(pfem_solver *s, int level, int *rbm, int *fine_bs, int *coarse_bs, int *dim, int64_t *n_nodes, double *node_xyz);
/* several ranks: is the hierarchy of the last solve one across the ranks (1) or one per rank (0); how many of its levels
 * are distributed over the ranks (the levels after them -- at most PFEM_AMG_REPLICATE_ROWS rows over all ranks, default
 * 150000 -- are assembled on every rank, which carries the rest of the cycle alone); per level (arrays of max_levels) the
 * global number of this rank's first dof and its local rows (owned + ghosts; replicated levels: 0 and all rows).  With a
 * coupled hierarchy pfem_solver_amg_aggregates hands out GLOBAL coarse numbers and pfem_solver_amg_info's rows are the
 * owned ones on the distributed levels, all rows on the replicated ones.                                                */
int pfem_solver_amg_layout(pfem_solver *s, int max_levels, int *coupled, int *distributed_levels, int64_t *first_dof, int64_t *local_rows);
/* the gather form's incidence lists as translated copies of patterns (lists equal up to a shift of the node numbers, as a mesh
 * numbered along lines has them): number of patterns in use (0: every node reads its own records) and the longest list      */
int pfem_solver_incidence_patterns(pfem_solver *s, int *count, int *longest);
/* how the aggregates of every level of the last gamg hierarchy were formed (kind[l], l < *n_levels; the last level: 0):
 * 1 bricks of the lattice in one step, 2 node bricks in one step (rigid-body transfer), 3 bricks split between their owners
 * (several ranks whose dofs do not fill boxes), 4 pairing passes along the axes of the lattice, 5 matching on the strength
 * graph, 6 roots + neighbours (an independent set of the strength graph).                                                 */
int pfem_solver_amg_aggregation(pfem_solver *s, int max_levels, int *n_levels, int *kind);
/* several ranks: neighbour exchanges and all-reduces ONE V-cycle of the last solve enqueued (next to the CG's own exchange
 * and two all-reduces per iteration); 0 / 0 on one rank                                                                */
int pfem_solver_amg_comm_counts(pfem_solver *s, int *exchanges_per_cycle, int *allreduces_per_cycle);
/* One instrumented V-cycle of the hierarchy across the ranks (collective, after a -pc_type gamg solve of a multi-rank run):
 * per level l < *n_levels the neighbour exchanges the cycle enqueues there, their summed time (event pairs on the stream that
 * carries them) and doubles sent; the cycle's all-reduces (the replicated level's right-hand side / the dense bottom) likewise;
 * the whole cycle's time.  PFEM_ERR_STATE without such a hierarchy.  Diagnostic (bench.py's N > 1 line). */
int pfem_solver_amg_cycle_profile(pfem_solver *s, int max_levels, int *n_levels, int *exchanges, double *exchange_ms,
                                  int64_t *exchange_doubles, int *allreduces, double *allreduce_ms, int64_t *allreduce_doubles,
                                  double *cycle_ms);
/* how many levels of the last hierarchy were coarsened by pairing on the mesh's lattice (the others: by matching on the
 * strength graph); 0 when the mesh has no lattice or came without coordinates                                          */
int pfem_solver_amg_pairing(pfem_solver *s, int *lattice_levels);
/* the cycle the last gamg solve ran: 1 = V, 2 = W, and the last level whose problem got two visits (0 with V)          */
int pfem_solver_amg_cycle(pfem_solver *s, int *cycle, int *last_level_visited_twice);

/* Per-element Ke/Fe of the uploaded mesh as computed by the DEVICE kernel (parity
 * inspection): K_out[e*nsize*nsize + i + nsize*j], F_out[e*nsize + i].           */
int pfem_eval_elems(pfem_solver *s, const double *elemData, const double *timeData,
                    double *K_out, double *F_out);

/* local matrix dimensions: n_local = owned + ghost rows, nnz of the pattern      */
int pfem_matrix_info(pfem_solver *s, int64_t *n_owned, int64_t *n_local, int64_t *nnz,
                     int64_t *stored_entries);

/* y = A_local x (both length n_local, host arrays): one launch of the CG SpMV kernel */
int pfem_spmv(pfem_solver *s, const double *x, double *y);
/* time `reps` back-to-back SpMV launches with HIP events on the solver's stream */
int pfem_bench_spmv(pfem_solver *s, int reps, double *ms_per_launch);

/* Timings of the last calls, measured with HIP events on the solver's stream [ms]. */
typedef struct pfem_timings {
    double pattern_ms;      /* pfem_pattern_build                                  */
    double assemble_ms;     /* pfem_assemble (reference timer :826 -> :893)        */
    double solve_ms;        /* pfem_solver_solve (reference timer :898 -> :902)    */
    double spmv_ms_total;   /* sum of the SpMV launches inside the last solve      */
    int64_t spmv_launches;  /* number of SpMV launches inside the last solve       */
    double upload_ms;       /* pfem_mesh_upload (PCIe, host wall clock)            */
    double event_overhead_ms; /* what a start/stop event pair reports for an EMPTY kernel (marker-to-
                               * dispatch gap of the measurement itself, calibrated at solve start);
                               * kernel time per SpMV = spmv_ms_total/spmv_launches - event_overhead_ms */
    double iface_ms_total;  /* multi-rank, sampled with the SpMV: pack + neighbour exchange, time on the comm stream ... */
    double scalar_ms_total; /* ... and the two scalar all-reduces of the iteration, time on the comm stream           */
    int64_t comm_samples;   /* number of iterations both were sampled in                                             */
    double exposed_ms_total;/* of those: time the compute stream spent WAITING for the comm stream (not hidden by the
                             * interior SpMV), per sampled iteration                                                 */
    int64_t graph_iterations; /* iterations of the last solve that were replayed from a hipGraph                       */
    double host_enqueue_ms;   /* host time spent enqueueing the iterations of the last solve (without the per-chunk     */
    int64_t host_enqueued_iterations; /* wait for the control block) and the number of iterations enqueued;            */
    double host_comm_ms;      /* of it: time inside the communication backend's calls (RCCL launch cost)              */
} pfem_timings;
int pfem_get_timings(pfem_solver *s, pfem_timings *t);
/* record an event pair around every SpMV launch of the next solves (bench.py) */
int pfem_solver_profile_spmv(pfem_solver *s, int enable);   /* 0 off, 1 every launch, k > 1 every k-th launch */

/* Transport self-test (collective, no mesh needed): stamped buffers of `count` doubles to every other rank (to itself
 * when there is one rank) through the backend's exchange, then an all-reduce of a known vector; *bad = wrong entries. */
int pfem_solver_comm_selftest(pfem_solver *s, int64_t count, int64_t *bad);
/* Transport timing (collective, no mesh needed): `reps` back-to-back exchanges of `count` doubles with every other rank (with
 * itself when there is one rank), then `reps` all-reduces of 4 doubles, each series between two events on the solver's stream
 * (for the host backend the time includes its staging); milliseconds per call.                                           */
int pfem_solver_comm_bench(pfem_solver *s, int64_t count, int reps, double *ms_per_exchange, double *ms_per_allreduce);
/* The same with the sizes of a run in hand: exchanges of `count` doubles (62 KB = 7 803: a face of config 4; 1.29 MB = 160 801:
 * a face of config 5) with every other rank, or (slab_neighbours != 0) with rank - 1 and rank + 1 only, what a slab partition
 * exchanges; all-reduces of `allreduce_count` doubles (3: the CG's scalars; 125 000: the right-hand side of a replicated
 * multigrid level).  bench.py prints these for N > 1 so that the one 8-GPU run shows what the links cost. */
int pfem_solver_comm_bench_sizes(pfem_solver *s, int64_t count, int64_t allreduce_count, int reps, int slab_neighbours,
                                 double *ms_per_exchange, double *ms_per_allreduce);
/* what the last solve exchanged per iteration: number of neighbours, doubles sent to all of them together, and
 * how many of the SpMV's slices hold shared rows (they run first) out of how many                                  */
int pfem_solver_comm_info(pfem_solver *s, int *n_peers, int64_t *doubles_per_exchange, int64_t *boundary_slices,
                          int64_t *total_slices);
/* What carries the multi-rank solve, as the transport reports it (replaces what `-log_view` / MPI_Comm_size would tell a
 * PETSc user, solverpetsc.F:447-476): backend name ("rccl", "host", "peer-ipc", "none"), ncclCommCount / ncclCommCuDevice /
 * ncclGetVersion of the bound communicators (-1 for host hooks), the device the solver runs on, and the form of the
 * multi-rank SpMV all ranks agreed on for the current plan (0 in order, 1 overlapped, -1 before the first solve).     */
int pfem_solver_comm_describe(pfem_solver *s, char *backend, int backend_len, int *backend_ranks, int *backend_device,
                              int *backend_version, int *solver_device, int *overlapped_form);

#ifdef __cplusplus
}
#endif
#endif /* PFEM_AMD_DIAG_H */
