/* oracle/pfem_oracle_mpi.c -- TEST / MEASUREMENT INFRASTRUCTURE, not product code.
 *
 * The reference's parallel run, `mpirun -np P tetrapoissonparallelimpl1` (tetrapoissonparallelimpl1.F:826-902: the
 * element loop between its timers, then KSPSolve), restated for the CPU-baseline leg of bench.py with one MPI rank per
 * core and without PETSc (absent here): every rank owns a slab of node planes of the [-1,1]^3 box (the partition the
 * reference's renumbering produces for z-slabs: the identity), evaluates the elements that touch its nodes with the
 * oracle's element routine (pfem_oracle.c: orc_poisson_tet_ke, elementutilitiespoisson.F:107-193), adds them into ITS rows
 * of a CSR matrix (MatSetValues ADD_VALUES with the Dirichlet lifting of :859-870; the neighbours' shares of an interface
 * row are computed here too instead of travelling through PETSc's stash), and runs KSPCG + point Jacobi with PETSc's
 * default convergence test on ||M^-1 r|| (orc_pcg_jacobi's loop, solverpetsc.F:187-476) -- one exchange of a node plane
 * with each neighbour per SpMV (MPI_Sendrecv), two MPI_Allreduce per iteration.
 *
 *   mpiexec -n P pfem_oracle_mpi <cells per side> [rtol=1e-5] [maxits=10000] [repeats=2]
 *
 * Rank 0 prints ONE JSON line: the times of the last repeat (maximum over the ranks, between barriers), the iteration
 * count, and max|u - (x^2+y^2+z^2)| at the nodes (the answer the whole-matrix oracle gives is what tests compare with).
 */
#include <math.h>
#include <mpi.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int orc_gen_box_tets(double x0, double x1, int nEx, double y0, double y1, int nEy, double z0, double z1, int nEz, int bc_mode, int ndof,
                     double *xyz, int32_t *conn, int64_t *nDBC, int32_t *bc_node, int32_t *bc_dof, double *bc_val);
int orc_poisson_tet_ke(const double *xN, const double *yN, const double *zN, const double *elemData, const double *timeData,
                       const double *valC, double *K, double *F);
int orc_csr_pattern(int64_t nElem, int nsize, const int32_t *edof, int64_t N, int64_t *rowptr, int32_t *cols);

static double round8(double v)
{
    char buf[64];
    snprintf(buf, sizeof buf, "%.8f", v);
    return strtod(buf, NULL);
}

#define DIE(msg) do { fprintf(stderr, "pfem_oracle_mpi rank %d: %s\n", rank, msg); MPI_Abort(MPI_COMM_WORLD, 1); } while (0)

int main(int argc, char **argv)
{
    int rank = 0, P = 1;
    MPI_Init(&argc, &argv);
    MPI_Comm_rank(MPI_COMM_WORLD, &rank);
    MPI_Comm_size(MPI_COMM_WORLD, &P);
    if (argc < 2) DIE("usage: pfem_oracle_mpi <cells> [rtol] [maxits] [repeats]");
    const int n = atoi(argv[1]);
    const double rtol = argc > 2 ? atof(argv[2]) : 1e-5;
    const int maxits = argc > 3 ? atoi(argv[3]) : 10000;
    const int repeats = argc > 4 ? atoi(argv[4]) : 2;
    const int m = n - 1;                                  /* free nodes per line, free node planes */
    if (n < 2 || P > m) DIE("needs cells >= 2 and at most cells-1 ranks (one free node plane per rank)");
    /* free planes k = 1..n-1 in contiguous blocks: rank owns [k0, k1) */
    const int k0 = 1 + (int)((int64_t)m * rank / P), k1 = 1 + (int)((int64_t)m * (rank + 1) / P);
    const int planes = k1 - k0;
    const int64_t per_plane = (int64_t)m * m, n_own = per_plane * planes;
    const int has_lo = k0 > 1, has_hi = k1 < n;           /* free ghost plane below / above */
    const int64_t n_loc = n_own + per_plane * (has_lo + has_hi);
    /* local mesh: hex layers k0-1 .. k1-1, node planes k0-1 .. k1 */
    const int nEz = planes + 1, nN1 = n + 1;
    const int64_t nNode = (int64_t)nN1 * nN1 * (nEz + 1), nElem = 6LL * n * n * nEz;
    double *xyz = (double *)malloc(sizeof(double) * 3 * (size_t)nNode);
    int32_t *conn = (int32_t *)malloc(sizeof(int32_t) * 4 * (size_t)nElem);
    int32_t *dof = (int32_t *)malloc(sizeof(int32_t) * (size_t)nNode);        /* local dof of a node, -1: Dirichlet */
    double *ubc = (double *)calloc((size_t)nNode, sizeof(double));
    int32_t *edof = (int32_t *)malloc(sizeof(int32_t) * 4 * (size_t)nElem);
    if (!xyz || !conn || !dof || !ubc || !edof) DIE("out of memory");
    int64_t nDBC = 0;
    const double dz = 2.0 / n;
    if (orc_gen_box_tets(-1.0, 1.0, n, -1.0, 1.0, n, -1.0 + dz * (k0 - 1), -1.0 + dz * k1, nEz, 0, 1, xyz, conn, &nDBC, NULL, NULL, NULL)) DIE("mesh");
    {   /* z of the node planes exactly as the whole mesh has them (genTetra.cpp accumulates z += dz from z0 and prints %.8f),
           and the Dirichlet values u = x^2+y^2+z^2 on float-rounded coordinates (genTetra.cpp:510-525) */
        double *zs = (double *)malloc(sizeof(double) * (size_t)nN1), *cs = (double *)malloc(sizeof(double) * (size_t)nN1), v = -1.0;
        int i, j, k;
        for (k = 0; k < nN1; ++k) { zs[k] = v; v += dz; }
        v = -1.0;
        for (i = 0; i < nN1; ++i) { cs[i] = v; v += 2.0 / n; }
        int64_t ind = 0;
        for (k = k0 - 1; k <= k1; ++k)
            for (j = 0; j < nN1; ++j)
                for (i = 0; i < nN1; ++i, ++ind) {
                    xyz[2 * nNode + ind] = round8(zs[k]);
                    const int on = i == 0 || i == n || j == 0 || j == n || k == 0 || k == n;
                    if (on) {
                        const double cx = (double)(float)cs[i], cy = (double)(float)cs[j], cz = (double)(float)zs[k];
                        ubc[ind] = round8(cx * cx + cy * cy + cz * cz);
                        dof[ind] = -1;
                    } else {
                        /* owned planes first, then the ghost plane below, then the one above */
                        const int64_t in_plane = (int64_t)(j - 1) * m + (i - 1);
                        int64_t base;
                        if (k >= k0 && k < k1) base = per_plane * (k - k0);
                        else if (k < k0) base = n_own;
                        else base = n_own + per_plane * has_lo;
                        dof[ind] = (int32_t)(base + in_plane);
                    }
                }
        free(zs); free(cs);
    }
    for (int64_t e = 0; e < nElem; ++e)
        for (int a = 0; a < 4; ++a) edof[(int64_t)a * nElem + e] = dof[conn[(int64_t)a * nElem + e]];
    /* pattern (before the timers, :786-802); ghost rows come out too and are never used */
    int64_t *rowptr = (int64_t *)malloc(sizeof(int64_t) * ((size_t)n_loc + 1));
    if (!rowptr || orc_csr_pattern(nElem, 4, edof, n_loc, rowptr, NULL)) DIE("pattern");
    int32_t *cols = (int32_t *)malloc(sizeof(int32_t) * (size_t)(rowptr[n_loc] ? rowptr[n_loc] : 1));
    if (!cols || orc_csr_pattern(nElem, 4, edof, n_loc, rowptr, cols)) DIE("pattern");
    const int64_t nnz_own = rowptr[n_own];
    double *vals = (double *)malloc(sizeof(double) * (size_t)(nnz_own ? nnz_own : 1));
    double *vec = (double *)malloc(sizeof(double) * (size_t)(5 * n_own + 2 * n_loc + 1));
    if (!vals || !vec) DIE("out of memory");
    double *rhs = vec, *x = rhs + n_own, *r = x + n_own, *z = r + n_own, *dinv = z + n_own, *p = dinv + n_own, *w = p + n_loc;
    const double elemData[3] = {1.0, 1.0, 1.0}, timeData[5] = {0.0, 1.0, 0.0, 0.0, 0.0}, valC[4] = {0.0, 0.0, 0.0, 0.0};
    double t_asm = 0.0, t_sol = 0.0;
    int its = 0, reason = 0;
    double rn = 0.0;
    for (int rep = 0; rep < repeats; ++rep) {
        MPI_Barrier(MPI_COMM_WORLD);
        const double t0 = MPI_Wtime();
        /* ---- element loop (:826-893): this rank's rows only */
        memset(vals, 0, sizeof(double) * (size_t)nnz_own);
        memset(rhs, 0, sizeof(double) * (size_t)n_own);
        for (int64_t e = 0; e < nElem; ++e) {
            int32_t idx[4], nd[4];
            int any = 0;
            for (int a = 0; a < 4; ++a) { nd[a] = conn[(int64_t)a * nElem + e]; idx[a] = edof[(int64_t)a * nElem + e]; any |= idx[a] >= 0 && idx[a] < n_own; }
            if (!any) continue;
            double xN[4], yN[4], zN[4], K[16], F[4];
            for (int a = 0; a < 4; ++a) { xN[a] = xyz[nd[a]]; yN[a] = xyz[nNode + nd[a]]; zN[a] = xyz[2 * nNode + nd[a]]; }
            if (orc_poisson_tet_ke(xN, yN, zN, elemData, timeData, valC, K, F)) DIE("negative Jacobian");
            for (int ii = 0; ii < 4; ++ii) {
                if (idx[ii] < 0 || idx[ii] >= n_own) continue;
                for (int jj = 0; jj < 4; ++jj) {
                    if (idx[jj] < 0) continue;
                    int64_t lo = rowptr[idx[ii]], hi = rowptr[idx[ii] + 1];
                    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (cols[mid] < idx[jj]) lo = mid + 1; else hi = mid; }
                    vals[lo] += K[jj + 4 * ii];
                }
            }
            for (int ii = 0; ii < 4; ++ii)
                if (idx[ii] == -1)
                    for (int jj = 0; jj < 4; ++jj)
                        if (idx[jj] != -1) F[jj] = F[jj] - K[jj + 4 * ii] * ubc[nd[ii]];
            for (int ii = 0; ii < 4; ++ii)
                if (idx[ii] >= 0 && idx[ii] < n_own) rhs[idx[ii]] += F[ii];
        }
        MPI_Barrier(MPI_COMM_WORLD);
        const double t1 = MPI_Wtime();
        /* ---- KSPCG + PCJACOBI (orc_pcg_jacobi's loop, distributed) */
        for (int64_t i = 0; i < n_own; ++i) {
            double d = 0.0;
            for (int64_t k = rowptr[i]; k < rowptr[i + 1]; ++k) if (cols[k] == i) d = vals[k];
            dinv[i] = 1.0 / d;
            x[i] = 0.0; r[i] = rhs[i]; z[i] = r[i] * dinv[i]; p[i] = z[i];
        }
        double loc[2], glo[2];
        loc[0] = loc[1] = 0.0;
        for (int64_t i = 0; i < n_own; ++i) { loc[0] += r[i] * z[i]; loc[1] += z[i] * z[i]; }
        MPI_Allreduce(loc, glo, 2, MPI_DOUBLE, MPI_SUM, MPI_COMM_WORLD);
        double beta = glo[0];
        const double rn0 = sqrt(glo[1]), ttol = fmax(rtol * rn0, 1e-50);
        rn = rn0; its = 0; reason = rn0 <= 1e-50 ? 3 : 0;
        while (!reason) {
            if (its >= maxits) { reason = -3; break; }
            ++its;
            /* the neighbours' planes of p: my lowest owned plane goes down, my highest up */
            if (P > 1) {
                const int dn = rank > 0 ? rank - 1 : MPI_PROC_NULL, up = rank + 1 < P ? rank + 1 : MPI_PROC_NULL;
                MPI_Sendrecv(p, (int)per_plane, MPI_DOUBLE, dn, 1, p + n_own + per_plane * has_lo, (int)per_plane, MPI_DOUBLE, up, 1, MPI_COMM_WORLD, MPI_STATUS_IGNORE);
                MPI_Sendrecv(p + n_own - per_plane, (int)per_plane, MPI_DOUBLE, up, 2, p + n_own, (int)per_plane, MPI_DOUBLE, dn, 2, MPI_COMM_WORLD, MPI_STATUS_IGNORE);
            }
            double pw = 0.0;
            for (int64_t i = 0; i < n_own; ++i) {
                double s = 0.0;
                for (int64_t k = rowptr[i]; k < rowptr[i + 1]; ++k) s += vals[k] * p[cols[k]];
                w[i] = s;
                pw += p[i] * s;
            }
            MPI_Allreduce(MPI_IN_PLACE, &pw, 1, MPI_DOUBLE, MPI_SUM, MPI_COMM_WORLD);
            if (!(pw > 0.0)) { reason = -10; break; }
            const double alpha = beta / pw;
            loc[0] = loc[1] = 0.0;
            for (int64_t i = 0; i < n_own; ++i) {
                x[i] += alpha * p[i];
                r[i] -= alpha * w[i];
                z[i] = r[i] * dinv[i];
                loc[0] += r[i] * z[i];
                loc[1] += z[i] * z[i];
            }
            MPI_Allreduce(loc, glo, 2, MPI_DOUBLE, MPI_SUM, MPI_COMM_WORLD);
            const double betan = glo[0];
            rn = sqrt(glo[1]);
            if (rn <= ttol) { reason = rn <= 1e-50 ? 3 : 2; break; }
            if (rn >= 1e5 * rn0) { reason = -4; break; }
            if (betan < 0.0) { reason = -8; break; }
            const double bb = betan / beta;
            for (int64_t i = 0; i < n_own; ++i) p[i] = z[i] + bb * p[i];
            beta = betan;
        }
        MPI_Barrier(MPI_COMM_WORLD);
        const double t2 = MPI_Wtime();
        double tl[2] = {t1 - t0, t2 - t1}, tg[2];
        MPI_Reduce(tl, tg, 2, MPI_DOUBLE, MPI_MAX, 0, MPI_COMM_WORLD);
        t_asm = tg[0]; t_sol = tg[1];
    }
    /* max |u - (x^2+y^2+z^2)| over the owned nodes */
    double err = 0.0;
    {
        int64_t ind = 0;
        for (int k = k0 - 1; k <= k1; ++k)
            for (int j = 0; j < nN1; ++j)
                for (int i = 0; i < nN1; ++i, ++ind)
                    if (dof[ind] >= 0 && dof[ind] < n_own) {
                        const double X = xyz[ind], Y = xyz[nNode + ind], Z = xyz[2 * nNode + ind];
                        const double d = fabs(x[dof[ind]] - (X * X + Y * Y + Z * Z));
                        if (d > err) err = d;
                    }
    }
    double errg = 0.0;
    long long nnz_l = (long long)nnz_own, nnz_g = 0;
    MPI_Reduce(&err, &errg, 1, MPI_DOUBLE, MPI_MAX, 0, MPI_COMM_WORLD);
    MPI_Reduce(&nnz_l, &nnz_g, 1, MPI_LONG_LONG, MPI_SUM, 0, MPI_COMM_WORLD);
    if (rank == 0)
        printf("{\"ranks\": %d, \"cells\": %d, \"free_dofs\": %lld, \"nnz\": %lld, \"rtol\": %g, \"iterations\": %d, \"converged_reason\": %d, "
               "\"rnorm\": %.17g, \"assembly_s\": %.6f, \"solve_s\": %.6f, \"total_s\": %.6f, \"max_nodal_error\": %.6e}\n",
               P, n, (long long)per_plane * m, nnz_g, rtol, its, reason, rn, t_asm, t_sol, t_asm + t_sol, errg);
    MPI_Finalize();
    return 0;
}
