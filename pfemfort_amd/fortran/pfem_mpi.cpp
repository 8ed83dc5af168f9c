// pfem_mpi.cpp -- MPI binding of libpfem_amd's multi-rank hooks, for hosts that are MPI programs
// (the reference's Fortran drivers under mpiexec: one MPI rank per GPU, or several ranks sharing one).
//
// Does for an MPI host what pfemfort_amd/distributed.py does for a torch.distributed host:
//   * gathers every rank's ghost dof list and derives the interface plan (which dofs are shared, and
//     their slot in the packed global interface vector) -- same rule as distributed.interface_plan;
//   * installs an all-reduce hook (include/pfem_amd.h: pfem_allreduce_fn).  The hook stages the
//     device buffer through the host and uses MPI_Reduce + MPI_Bcast rather than MPI_Allreduce: the
//     solver keeps replicated ghost rows and takes its convergence decisions on every rank from the
//     reduced scalars, so all ranks must receive THE SAME BITS, which MPI_Allreduce does not promise.
// Replaces, for the reference, PETSc's MatAssemblyEnd stash exchange, the VecScatter inside MatMult
// and the MPI_Allreduce inside VecDot (solverpetsc.F:447-476), and VecScatterCreateToAll (:922-932).
#include <mpi.h>

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdint>
#include <vector>

#include "../../include/pfem_amd.h"

namespace {

struct Ctx {
    MPI_Comm comm;
    std::vector<double> host;
};

int allreduce_hook(void *vctx, void *buf, int64_t count, void *stream)
{
    Ctx *c = static_cast<Ctx *>(vctx);
    if (count <= 0) return 0;
    if (count > INT32_MAX) return 1;
    c->host.resize(static_cast<size_t>(count) * 2);
    double *in = c->host.data(), *out = in + count;
    if (hipStreamSynchronize(static_cast<hipStream_t>(stream)) != hipSuccess) return 1;
    if (hipMemcpy(in, buf, sizeof(double) * count, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    if (MPI_Reduce(in, out, static_cast<int>(count), MPI_DOUBLE, MPI_SUM, 0, c->comm) != MPI_SUCCESS) return 1;
    if (MPI_Bcast(out, static_cast<int>(count), MPI_DOUBLE, 0, c->comm) != MPI_SUCCESS) return 1;
    if (hipMemcpy(buf, out, sizeof(double) * count, hipMemcpyHostToDevice) != hipSuccess) return 1;
    return 0;
}

}  // namespace

// Wire a solver whose local numbering exists (compat path: after the first setZero; batched path:
// after pfem_mesh_upload) to the communicator.  `fcomm` is the Fortran handle (MPI_Comm_c2f).
extern "C" int pfem_mpi_attach(pfem_solver *s, int fcomm, int64_t row_start, int64_t size_local)
{
    if (!s) return PFEM_ERR_ARG;
    MPI_Comm comm = MPI_Comm_f2c(static_cast<MPI_Fint>(fcomm));
    int rank = 0, world = 1;
    MPI_Comm_rank(comm, &rank);
    MPI_Comm_size(comm, &world);
    if (world == 1) return PFEM_OK;

    int64_t ng = 0;
    int rc = pfem_get_ghosts(s, &ng, nullptr);
    if (rc != PFEM_OK) return rc;
    std::vector<int64_t> mine(static_cast<size_t>(ng) + 1);
    if (ng) { rc = pfem_get_ghosts(s, &ng, mine.data()); if (rc != PFEM_OK) return rc; }
    mine.resize(static_cast<size_t>(ng));

    // every rank's ghost list and owned row range
    std::vector<int> counts(world), displs(world);
    const int my_n = static_cast<int>(ng);
    MPI_Allgather(&my_n, 1, MPI_INT, counts.data(), 1, MPI_INT, comm);
    int64_t total = 0;
    for (int r = 0; r < world; ++r) { displs[r] = static_cast<int>(total); total += counts[r]; }
    if (total > INT32_MAX) return PFEM_ERR_ARG;
    std::vector<int64_t> all(static_cast<size_t>(total) + 1);
    MPI_Allgatherv(mine.data(), my_n, MPI_INT64_T, all.data(), counts.data(), displs.data(), MPI_INT64_T, comm);
    all.resize(static_cast<size_t>(total));

    // interface = union of all ghost lists, numbered ascending (identical on every rank)
    std::vector<int64_t> iface(all);
    std::sort(iface.begin(), iface.end());
    iface.erase(std::unique(iface.begin(), iface.end()), iface.end());
    // this rank touches: the interface dofs it owns, and its own ghosts
    std::vector<int64_t> gid;
    std::vector<int32_t> slot;
    const int64_t lo = row_start, hi = row_start + size_local;
    for (size_t k = 0; k < iface.size(); ++k) {
        const int64_t g = iface[k];
        const bool owned = g >= lo && g < hi;
        if (owned || std::binary_search(mine.begin(), mine.end(), g)) {
            gid.push_back(g);
            slot.push_back(static_cast<int32_t>(k));
        }
    }
    rc = pfem_solver_set_interface(s, static_cast<int64_t>(gid.size()), gid.data(), slot.data(),
                                   static_cast<int64_t>(iface.size()));
    if (rc != PFEM_OK) return rc;
    Ctx *ctx = new Ctx{comm, {}};      // lives as long as the process: the solver keeps the pointer
    return pfem_solver_set_comm(s, rank, world, allreduce_hook, ctx);
}

// VecScatterCreateToAll + VecGetArray: every rank receives the whole solution (size_global doubles)
extern "C" int pfem_mpi_gather_solution(pfem_solver *s, int fcomm, int64_t size_local, double *out)
{
    if (!s || !out || size_local < 0 || size_local > INT32_MAX) return PFEM_ERR_ARG;
    MPI_Comm comm = MPI_Comm_f2c(static_cast<MPI_Fint>(fcomm));
    int world = 1;
    MPI_Comm_size(comm, &world);
    std::vector<double> own(static_cast<size_t>(size_local) + 1);
    const int rc = pfem_solver_get_solution(s, own.data());
    if (rc != PFEM_OK) return rc;
    std::vector<int> counts(world), displs(world);
    const int my_n = static_cast<int>(size_local);
    MPI_Allgather(&my_n, 1, MPI_INT, counts.data(), 1, MPI_INT, comm);
    int64_t total = 0;
    for (int r = 0; r < world; ++r) { displs[r] = static_cast<int>(total); total += counts[r]; }
    MPI_Allgatherv(own.data(), my_n, MPI_DOUBLE, out, counts.data(), displs.data(), MPI_DOUBLE, comm);
    return PFEM_OK;
}

// Several MPI ranks on a node: rank r drives GPU (r mod device count), the usual one-process-per-GPU
// placement; on a 1-GPU box all ranks share device 0.
extern "C" int pfem_mpi_pick_device(int fcomm, int *device)
{
    if (!device) return PFEM_ERR_ARG;
    MPI_Comm comm = MPI_Comm_f2c(static_cast<MPI_Fint>(fcomm));
    MPI_Comm node;
    MPI_Comm_split_type(comm, MPI_COMM_TYPE_SHARED, 0, MPI_INFO_NULL, &node);
    int local = 0, count = 0;
    MPI_Comm_rank(node, &local);
    MPI_Comm_free(&node);
    const int rc = pfem_device_count(&count);
    if (rc != PFEM_OK) return rc;
    *device = count > 0 ? local % count : 0;
    return PFEM_OK;
}
