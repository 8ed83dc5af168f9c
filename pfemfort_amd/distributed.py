"""Multi-GPU host logic: one process per GPU, sub-assembled interface rows.

Replaces what PETSc does inside ``MatAssemblyEnd`` (off-rank stash), ``MatMult`` (VecScatter
halo) and ``VecDot`` (MPI_Allreduce) for the reference (solverpetsc.F:447-476):

* every rank assembles ONLY its own elements (``elem_proc_id == rank``,
  tetrapoissonparallelimpl1.F:829) into a local matrix over owned + ghost rows;
* rows of interface dofs are therefore partial sums; each SpMV is followed by ONE
  all-reduce of a packed interface vector (``n_iface_global`` doubles + the (p,Ap) scalar), and
  the two remaining CG scalars ride in a second 2-double all-reduce;
* the collective is supplied to the C library as a hook.  Here it is bound to
  ``torch.distributed`` (backend "nccl" == RCCL over xGMI on the GPU box; "gloo" in the CPU
  tests), operating in place on a device tensor that the library uses as exchange buffer.

The interface plan (which dofs are shared, and their slot in the packed vector) is pure integer
host logic and is what the world_size-2 gloo tests pin.
"""
from __future__ import annotations

import numpy as np


def interface_plan(ghost_lists, row_ranges, rank):
    """``ghost_lists[r]`` = ascending global dof ids rank r touches but does not own;
    ``row_ranges[r] = (row_start, row_end)``.  Returns (shared_gid, shared_slot, n_iface_global)
    for ``rank``: the interface is the union of all ghost lists, numbered ascending."""
    iface = np.unique(np.concatenate([np.asarray(g, dtype=np.int64) for g in ghost_lists] + [np.empty(0, np.int64)]))
    lo, hi = row_ranges[rank]
    mine = np.asarray(ghost_lists[rank], dtype=np.int64)
    touched = np.zeros(len(iface), bool)
    touched |= (iface >= lo) & (iface < hi)                       # interface dofs this rank owns
    touched[np.searchsorted(iface, mine)] = True                  # and its own ghosts
    slots = np.nonzero(touched)[0].astype(np.int32)
    return iface[slots], slots, int(len(iface))


def gather_ghost_lists(ghosts, dist):
    """all_gather of variable-length int64 arrays through ``torch.distributed``."""
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, np.asarray(ghosts, dtype=np.int64))
    return out


class TorchAllReduce:
    """The all-reduce hook bound to torch.distributed on a torch-owned exchange tensor."""

    def __init__(self, dist, xbuf):
        self.dist = dist
        self.xbuf = xbuf                         # 1-D float64 tensor on the solver's device
        self.base = xbuf.data_ptr()
        self.calls = 0
        self.error = None

    def __call__(self, ctx, buf, count, stream):
        try:
            off = (int(buf) - self.base) // 8
            self.dist.all_reduce(self.xbuf[off:off + int(count)])    # SUM, in place, current stream
            self.calls += 1
            return 0
        except Exception as e:   # never let an exception cross the C boundary
            self.error = e
            return 1


class HostStagedAllReduce:
    """The all-reduce hook for process groups that work on HOST memory (gloo): stream-sync, device -> host,
    reduce to rank 0 + broadcast (every rank receives the same bits, whatever the group's all-reduce
    algorithm), host -> device.  What pfemfort_amd/fortran/pfem_mpi.cpp does with MPI; used where several
    ranks share one GPU (RCCL refuses two ranks on a device)."""

    def __init__(self, dist, torch, xbuf, stream):
        self.dist, self.torch = dist, torch
        self.xbuf = xbuf
        self.base = xbuf.data_ptr()
        self.stream = stream                     # torch.cuda.Stream the solver was given
        self.calls = 0
        self.error = None
        self.log = None                          # optional list of (call#, count) for call-sequence checks

    def __call__(self, ctx, buf, count, stream):
        try:
            off = (int(buf) - self.base) // 8
            view = self.xbuf[off:off + int(count)]
            with self.torch.cuda.stream(self.stream):
                self.stream.synchronize()
                host = view.cpu()
                self.dist.reduce(host, 0)
                self.dist.broadcast(host, 0)
                view.copy_(host)
                self.stream.synchronize()
            self.calls += 1
            if self.log is not None:
                self.log.append((self.calls, int(count)))
            return 0
        except Exception as e:   # never let an exception cross the C boundary
            self.error = e
            return 1


def attach(solver, dist, torch, device, staged=False):
    """Wire a solver that already holds its mesh to the process group: exchanges the ghost
    lists, installs the interface plan, the exchange buffer and the all-reduce hook.
    ``staged``: the group reduces host memory (gloo) -- the hook stages the exchange buffer through the host."""
    rank, world = dist.get_rank(), dist.get_world_size()
    ghosts = solver.ghosts()
    lists = gather_ghost_lists(ghosts, dist)
    ranges = [None] * world
    dist.all_gather_object(ranges, (solver.row_start, solver.row_start + solver.size_local))
    gid, slot, n_iface = interface_plan(lists, ranges, rank)
    xbuf = torch.zeros(n_iface + 4, dtype=torch.float64, device=device)
    if staged:
        stream = torch.cuda.Stream(device)
        hook = HostStagedAllReduce(dist, torch, xbuf, stream)
        solver.setStream(stream.cuda_stream)
        solver._keep.append(stream)
    else:
        hook = TorchAllReduce(dist, xbuf)
        if device.type == "cuda":
            solver.setStream(torch.cuda.current_stream(device).cuda_stream)
    solver.setExchangeBuffer(xbuf.data_ptr(), n_iface + 4)
    solver.setInterface(gid, slot, n_iface)
    solver.setComm(rank, world, hook)
    solver._keep.extend([xbuf, hook])
    return hook, n_iface
