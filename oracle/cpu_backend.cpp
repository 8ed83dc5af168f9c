// cpu_backend.cpp -- TEST INFRASTRUCTURE ONLY (build container): a CPU stand-in for the few C-ABI entry points of
// include/pfem_amd.h that the Fortran side of the boundary (pfemfort_amd/fortran) calls, so that the reference's
// UNCHANGED driver programs can be run where no GPU exists and their own bookkeeping can be captured as committed
// fixtures (tests/golden/drivers/, made by tests/golden/make_driver_fixtures.py).  Never shipped, never linked by
// the product, never sent to the GPU box.
//
// What it does: stages MatSetValues / VecSetValues exactly like PETSc would see them (global indices, negative
// ones ignored, value block read row-major), RECORDS every call (the driver's ElemDofArray rows, the lifted
// element vectors, the row blocks) and, at solve time, assembles the global system from all ranks' entries and
// runs the oracle's Jacobi-PCG (oracle/pfem_oracle.c: orc_pcg_jacobi).  With -DPFEM_WITH_MPI the entries of all
// ranks are exchanged with MPI_Allgatherv (what MatAssemblyEnd's stash does, solverpetsc.F:447-450).
//
// Trace files (cwd): pfem_trace.<rank>.txt (header) + .insert.i32 / .add.i32 / .vecidx.i32 / .vecval.f64 /
// .soln.f64, raw little-endian arrays.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#ifdef PFEM_WITH_MPI
#include <mpi.h>
#endif

#include "../include/pfem_amd.h"

extern "C" int orc_pcg_jacobi(int64_t N, const int64_t *rowptr, const int32_t *cols, const double *vals, const double *b,
                              double *x, double rtol, double abstol, double dtol, int maxits, int *its_out, int *reason_out,
                              double *rnorm_out, double *history, int hist_len);

struct pfem_solver {
    int64_t size_local = 0, size_global = 0, row_start = 0;
    int status = PFEM_SOLVER_EMPTY;
    double rtol = 1e-5, abstol = 1e-50, dtol = 1e5;
    int maxits = 10000;
    int rank = 0, nranks = 1;
    bool pattern_final = false;
    std::vector<int32_t> rows, cols;         // staged entries of this rank (global ids), in call order
    std::vector<double> vals;
    std::vector<int32_t> ridx;               // staged rhs entries
    std::vector<double> rval;
    std::vector<double> x;                   // global solution (every rank holds all of it)
    int its = 0, reason = 0;
    double rnorm = 0;
    // trace
    std::vector<int32_t> t_insert, t_add, t_vecidx;   // m, idxm[0..m) per call
    std::vector<double> t_vecval;
    int64_t n_insert = 0, n_add = 0, n_vec = 0;
#ifdef PFEM_WITH_MPI
    MPI_Comm comm = MPI_COMM_WORLD;
#endif
};

namespace {
std::string g_err;

void dump(const pfem_solver *s)
{
    char name[256];
    auto put = [&](const char *sfx, const void *p, size_t bytes) {
        std::snprintf(name, sizeof name, "pfem_trace.%d.%s", s->rank, sfx);
        FILE *f = std::fopen(name, "wb");
        if (!f) return;
        if (bytes) std::fwrite(p, 1, bytes, f);
        std::fclose(f);
    };
    put("insert.i32", s->t_insert.data(), s->t_insert.size() * 4);
    put("add.i32", s->t_add.data(), s->t_add.size() * 4);
    put("vecidx.i32", s->t_vecidx.data(), s->t_vecidx.size() * 4);
    put("vecval.f64", s->t_vecval.data(), s->t_vecval.size() * 8);
    put("soln.f64", s->x.data(), s->x.size() * 8);
    std::snprintf(name, sizeof name, "pfem_trace.%d.txt", s->rank);
    if (FILE *f = std::fopen(name, "w")) {
        std::fprintf(f, "rank %d\nnranks %d\nsize_local %lld\nsize_global %lld\nrow_start %lld\nn_insert %lld\nn_add %lld\nn_vec %lld\nits %d\nreason %d\nrtol %.17g\n",
                     s->rank, s->nranks, (long long)s->size_local, (long long)s->size_global, (long long)s->row_start,
                     (long long)s->n_insert, (long long)s->n_add, (long long)s->n_vec, s->its, s->reason, s->rtol);
        std::fclose(f);
    }
}
}  // namespace

extern "C" {

int pfem_version(void) { return PFEM_VERSION; }
const char *pfem_strerror(int code) { return code == PFEM_OK ? "ok" : "cpu trace backend error"; }
const char *pfem_last_error_string(void) { return g_err.c_str(); }

int pfem_solver_create(pfem_solver **out, int64_t size_local, int64_t size_global, int64_t row_start, const int *, const int *, int)
{
    if (!out) return PFEM_ERR_ARG;
    pfem_solver *s = new pfem_solver();
    s->size_local = size_local;
    s->size_global = size_global;
    s->row_start = row_start;
#ifdef PFEM_WITH_MPI
    MPI_Comm_rank(s->comm, &s->rank);
    MPI_Comm_size(s->comm, &s->nranks);
#endif
    *out = s;
    return PFEM_OK;
}

int pfem_solver_destroy(pfem_solver *s)
{
    if (s) { dump(s); delete s; }
    return PFEM_OK;
}

int pfem_solver_set_tolerances(pfem_solver *s, double rtol, double abstol, double dtol, int maxits)
{
    s->rtol = rtol; s->abstol = abstol; s->dtol = dtol; s->maxits = maxits;
    return PFEM_OK;
}
int pfem_solver_set_preconditioner(pfem_solver *, int) { return PFEM_OK; }
int pfem_solver_status(pfem_solver *s, int *st) { *st = s->status; return PFEM_OK; }
int pfem_solver_print_info(pfem_solver *) { return PFEM_OK; }

int pfem_matrix_info(pfem_solver *s, int64_t *n_owned, int64_t *n_local, int64_t *nnz, int64_t *stored)
{
    if (n_owned) *n_owned = s->size_local;
    if (n_local) *n_local = s->size_local;
    if (nnz) *nnz = 0;
    if (stored) *stored = 0;
    return PFEM_OK;
}

int pfem_mat_set_values(pfem_solver *s, int m, const int *idxm, int n, const int *idxn, const double *v, int mode)
{
    const bool insert_pass = s->status == PFEM_SOLVER_EMPTY;
    std::vector<int32_t> &t = insert_pass ? s->t_insert : s->t_add;
    t.push_back(m);
    t.insert(t.end(), idxm, idxm + m);
    (insert_pass ? s->n_insert : s->n_add)++;
    if (insert_pass) return PFEM_OK;               // the pattern: nothing to accumulate
    for (int i = 0; i < m; ++i) {
        if (idxm[i] < 0) continue;
        for (int j = 0; j < n; ++j) {
            if (idxn[j] < 0) continue;
            s->rows.push_back(idxm[i]);
            s->cols.push_back(idxn[j]);
            s->vals.push_back(v[(size_t)i * n + j]);        // PETSc reads v row-major
        }
    }
    (void)mode;
    return PFEM_OK;
}

int pfem_vec_set_values(pfem_solver *s, int n, const int *idx, const double *v, int)
{
    s->t_vecidx.push_back(n);
    s->t_vecidx.insert(s->t_vecidx.end(), idx, idx + n);
    s->t_vecval.insert(s->t_vecval.end(), v, v + n);
    s->n_vec++;
    for (int i = 0; i < n; ++i)
        if (idx[i] >= 0) { s->ridx.push_back(idx[i]); s->rval.push_back(v[i]); }
    return PFEM_OK;
}

int pfem_solver_assemble_matrix_and_vector(pfem_solver *s, int n, const int *rows, const int *cols, const double *K, const double *F)
{
    if (K) {
        std::vector<double> rm((size_t)n * n);
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) rm[(size_t)i * n + j] = K[i + (size_t)n * j];
        pfem_mat_set_values(s, n, rows, n, cols, rm.data(), PFEM_ADD_VALUES);
    }
    if (F) pfem_vec_set_values(s, n, rows, F, PFEM_ADD_VALUES);
    return PFEM_OK;
}

int pfem_solver_set_zero(pfem_solver *s)
{
    s->rows.clear(); s->cols.clear(); s->vals.clear(); s->ridx.clear(); s->rval.clear();
    s->status = PFEM_INIT_OK;
    return PFEM_OK;
}

int pfem_solver_factorise(pfem_solver *s) { s->status = PFEM_FACTORISE_OK; return PFEM_OK; }

int pfem_solver_solve(pfem_solver *s, int *its, int *reason, double *rnorm)
{
    std::vector<int32_t> R = s->rows, Cc = s->cols, RI = s->ridx;
    std::vector<double> V = s->vals, RV = s->rval;
#ifdef PFEM_WITH_MPI
    if (s->nranks > 1) {          // the stash exchange: every rank ends up with all entries, rank by rank
        auto gatherv = [&](auto &vec, MPI_Datatype ty) {
            int n = (int)vec.size();
            std::vector<int> cnt(s->nranks), dsp(s->nranks);
            MPI_Allgather(&n, 1, MPI_INT, cnt.data(), 1, MPI_INT, s->comm);
            int tot = 0;
            for (int r = 0; r < s->nranks; ++r) { dsp[r] = tot; tot += cnt[r]; }
            std::remove_reference_t<decltype(vec)> all((size_t)tot);
            MPI_Allgatherv(vec.data(), n, ty, all.data(), cnt.data(), dsp.data(), ty, s->comm);
            vec.swap(all);
        };
        gatherv(R, MPI_INT32_T); gatherv(Cc, MPI_INT32_T); gatherv(V, MPI_DOUBLE);
        gatherv(RI, MPI_INT32_T); gatherv(RV, MPI_DOUBLE);
    }
#endif
    const int64_t N = s->size_global;
    std::vector<std::map<int32_t, double>> rowmap((size_t)N);
    for (size_t k = 0; k < R.size(); ++k) rowmap[R[k]][Cc[k]] += V[k];
    std::vector<int64_t> rowptr((size_t)N + 1, 0);
    std::vector<int32_t> cols;
    std::vector<double> vals;
    for (int64_t i = 0; i < N; ++i) {
        for (auto &kv : rowmap[i]) { cols.push_back(kv.first); vals.push_back(kv.second); }
        rowptr[i + 1] = (int64_t)cols.size();
    }
    std::vector<double> b((size_t)N, 0.0);
    for (size_t k = 0; k < RI.size(); ++k) b[RI[k]] += RV[k];
    s->x.assign((size_t)N, 0.0);
    const int rc = orc_pcg_jacobi(N, rowptr.data(), cols.data(), vals.data(), b.data(), s->x.data(), s->rtol, s->abstol, s->dtol,
                                  s->maxits, &s->its, &s->reason, &s->rnorm, nullptr, 0);
    if (rc != 0) { g_err = "orc_pcg_jacobi failed"; return PFEM_ERR_NOMEM; }
    if (its) *its = s->its;
    if (reason) *reason = s->reason;
    if (rnorm) *rnorm = s->rnorm;
    return PFEM_OK;
}

int pfem_solver_factorise_and_solve(pfem_solver *s, int *its, int *reason, double *rnorm)
{
    s->status = PFEM_FACTORISE_OK;
    return pfem_solver_solve(s, its, reason, rnorm);
}

int pfem_solver_get_solution(pfem_solver *s, double *x_owned)
{
    if (s->x.empty() && s->size_global > 0) return PFEM_ERR_STATE;
    std::copy(s->x.begin() + s->row_start, s->x.begin() + s->row_start + s->size_local, x_owned);
    return PFEM_OK;
}

// the device element loop has no stand-in here: the reference drivers never call it
int pfem_mesh_upload(pfem_solver *, int, int64_t, const int32_t *, int64_t, const double *, const int32_t *, const double *)
{
    g_err = "the CPU trace backend has no batched path";
    return PFEM_ERR_STATE;
}
int pfem_pattern_build(pfem_solver *) { g_err = "the CPU trace backend has no batched path"; return PFEM_ERR_STATE; }
int pfem_assemble(pfem_solver *, const double *, const double *) { g_err = "the CPU trace backend has no batched path"; return PFEM_ERR_STATE; }

// MPI flavour of the Fortran shim (pfemfort_amd/fortran/pfem_mpi.cpp provides these for the product)
int pfem_mpi_attach(pfem_solver *, int, int64_t, int64_t) { return PFEM_OK; }
int pfem_mpi_pick_device(int, int *device) { *device = 0; return PFEM_OK; }
int pfem_mpi_gather_solution(pfem_solver *s, int, int64_t, double *out)
{
    std::copy(s->x.begin(), s->x.end(), out);
    return PFEM_OK;
}

}  // extern "C"
