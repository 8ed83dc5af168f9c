// pfem_mpi.cpp -- MPI binding of libpfem_amd's multi-rank interface, for hosts that are MPI programs
// (the reference's Fortran drivers under mpiexec: one MPI rank per GPU, or several ranks sharing one).
//
// Does for an MPI host what pfemfort_amd/distributed.py does for a torch.distributed host:
//   * gathers every rank's ghost dof list and row block and derives the neighbour plan with the library's own
//     pfem_neighbour_plan (who shares which dofs with whom);
//   * installs the HOST communication backend (include/pfem_amd.h: pfem_solver_set_comm_host): the library
//     stages its packed buffers through pinned memory and calls back here --
//       exchange : MPI_Isend / MPI_Irecv with every neighbour, MPI_Waitall;
//       allreduce: MPI_Reduce + MPI_Bcast rather than MPI_Allreduce -- the solver keeps replicated ghost rows and
//                  takes its convergence decisions on every rank from the reduced scalars, so all ranks must receive
//                  THE SAME BITS, which MPI_Allreduce does not promise.
//     (Set PFEM_MPI_RCCL=1 to use RCCL for the transport instead, the unique id travelling by MPI_Bcast: one rank
//     per GPU only.)
// Replaces, for the reference, PETSc's MatAssemblyEnd stash exchange, the VecScatter inside MatMult and the
// MPI_Allreduce inside VecDot (solverpetsc.F:447-476), and VecScatterCreateToAll (:922-932).
#include <mpi.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <vector>

#include "../../include/pfem_amd.h"

namespace {

struct Ctx {
    MPI_Comm comm;
    std::vector<MPI_Request> req;
};

int allreduce_hook(void *vctx, double *buf, int64_t count)
{
    Ctx *c = static_cast<Ctx *>(vctx);
    if (count <= 0) return 0;
    if (count > INT32_MAX) return 1;
    int rank = 0;
    MPI_Comm_rank(c->comm, &rank);
    std::vector<double> out(static_cast<size_t>(count));
    if (MPI_Reduce(buf, out.data(), static_cast<int>(count), MPI_DOUBLE, MPI_SUM, 0, c->comm) != MPI_SUCCESS) return 1;
    if (rank == 0) std::copy(out.begin(), out.end(), buf);
    if (MPI_Bcast(buf, static_cast<int>(count), MPI_DOUBLE, 0, c->comm) != MPI_SUCCESS) return 1;
    return 0;
}

int exchange_hook(void *vctx, int n_peers, const int *peers, const int64_t *off, const double *send, double *recv)
{
    Ctx *c = static_cast<Ctx *>(vctx);
    c->req.assign(static_cast<size_t>(2 * n_peers), MPI_REQUEST_NULL);
    for (int k = 0; k < n_peers; ++k) {
        const int64_t cnt = off[k + 1] - off[k];
        if (cnt > INT32_MAX) return 1;
        if (MPI_Irecv(recv + off[k], static_cast<int>(cnt), MPI_DOUBLE, peers[k], 77, c->comm, &c->req[2 * k]) != MPI_SUCCESS) return 1;
        if (MPI_Isend(send + off[k], static_cast<int>(cnt), MPI_DOUBLE, peers[k], 77, c->comm, &c->req[2 * k + 1]) != MPI_SUCCESS) return 1;
    }
    return MPI_Waitall(2 * n_peers, c->req.data(), MPI_STATUSES_IGNORE) == MPI_SUCCESS ? 0 : 1;
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

    // every rank's ghost list and owned row block
    std::vector<int> counts(world), displs(world);
    const int my_n = static_cast<int>(ng);
    MPI_Allgather(&my_n, 1, MPI_INT, counts.data(), 1, MPI_INT, comm);
    std::vector<int64_t> ghost_off(static_cast<size_t>(world) + 1, 0);
    for (int r = 0; r < world; ++r) { displs[r] = static_cast<int>(ghost_off[r]); ghost_off[r + 1] = ghost_off[r] + counts[r]; }
    if (ghost_off[world] > INT32_MAX) return PFEM_ERR_ARG;
    std::vector<int64_t> all(static_cast<size_t>(ghost_off[world]) + 1);
    MPI_Allgatherv(mine.data(), my_n, MPI_INT64_T, all.data(), counts.data(), displs.data(), MPI_INT64_T, comm);
    std::vector<int64_t> rs(world), re(world);
    const int64_t my_rs = row_start, my_re = row_start + size_local;
    MPI_Allgather(&my_rs, 1, MPI_INT64_T, rs.data(), 1, MPI_INT64_T, comm);
    MPI_Allgather(&my_re, 1, MPI_INT64_T, re.data(), 1, MPI_INT64_T, comm);

    int n_peers = 0;
    int64_t total = 0;
    rc = pfem_neighbour_plan(world, rank, rs.data(), re.data(), ghost_off.data(), all.data(), &n_peers, &total, nullptr, nullptr, nullptr);
    if (rc != PFEM_OK) return rc;
    std::vector<int> peers(static_cast<size_t>(n_peers) + 1);
    std::vector<int64_t> peer_off(static_cast<size_t>(n_peers) + 1, 0), gid(static_cast<size_t>(total) + 1);
    if (n_peers) {
        rc = pfem_neighbour_plan(world, rank, rs.data(), re.data(), ghost_off.data(), all.data(), &n_peers, &total, peers.data(),
                                 peer_off.data(), gid.data());
        if (rc != PFEM_OK) return rc;
    }
    const char *use_rccl = std::getenv("PFEM_MPI_RCCL");
    if (use_rccl && use_rccl[0] == '1') {
        char id[PFEM_RCCL_ID_BYTES] = {0};
        if (rank == 0) { rc = pfem_rccl_unique_id(id); }
        MPI_Bcast(&rc, 1, MPI_INT, 0, comm);
        if (rc != PFEM_OK) return rc;
        MPI_Bcast(id, PFEM_RCCL_ID_BYTES, MPI_BYTE, 0, comm);
        rc = pfem_solver_set_comm_rccl(s, rank, world, id);
    } else {
        Ctx *ctx = new Ctx{comm, {}};      // lives as long as the process: the solver keeps the pointer
        rc = pfem_solver_set_comm_host(s, rank, world, allreduce_hook, exchange_hook, ctx);
    }
    if (rc != PFEM_OK) return rc;
    return pfem_solver_set_neighbours(s, n_peers, peers.data(), peer_off.data(), gid.data());
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
