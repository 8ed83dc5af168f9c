"""Multi-GPU host logic: one process per GPU, sub-assembled interface rows.

Replaces what PETSc does inside ``MatAssemblyEnd`` (off-rank stash), ``MatMult`` (VecScatter
halo) and ``VecDot`` (MPI_Allreduce) for the reference (solverpetsc.F:447-476):

* every rank assembles ONLY its own elements (``elem_proc_id == rank``,
  tetrapoissonparallelimpl1.F:829) into a local matrix over owned + ghost rows;
* rows of dofs that another rank also holds are therefore partial sums; every SpMV is followed by a
  NEIGHBOUR exchange of those partials (each rank adds them in ascending rank order), overlapped with
  the interior part of the SpMV, and the CG scalars ride in two small all-reduces;
* transport: RCCL over xGMI, bound inside the C++ library (``pfem_solver_set_comm_rccl``) -- this
  module only carries the 256 bytes of the two unique ids from rank 0 to the others; or host hooks over a
  ``torch.distributed`` group that works on host memory (gloo: the tests where ranks share a GPU).

The neighbour plan is pure integer host logic (``pfem_neighbour_plan``) and is what the world_size-2/3
gloo tests pin on CPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import host as H
from .solver import rccl_unique_id


def gather_ghost_lists(ghosts, dist, group=None):
    """all_gather of variable-length int64 arrays through ``torch.distributed``."""
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, np.asarray(ghosts, dtype=np.int64), group=group)
    return out


def plan_for(solver, dist, group=None):
    """Exchange ghost lists and row blocks, derive this rank's neighbour plan."""
    rank, world = dist.get_rank(), dist.get_world_size()
    lists = gather_ghost_lists(solver.ghosts(), dist, group)
    ranges = [None] * world
    dist.all_gather_object(ranges, (solver.row_start, solver.row_start + solver.size_local), group=group)
    return H.neighbour_plan(rank, ranges, lists)


class HostHooks:
    """Host-memory hooks over a ``torch.distributed`` group (gloo).  The all-reduce is reduce-to-rank-0 +
    broadcast, so that every rank receives the same bits whatever algorithm the group would pick."""

    def __init__(self, dist, torch, group=None):
        self.dist, self.torch, self.group = dist, torch, group
        self.calls = 0               # hook calls (all-reduces + exchanges)
        self.error = None
        self.log = None              # optional list of ("a"|"x", count) for call-sequence checks

    def allreduce(self, ctx, buf, count):
        try:
            a = np.ctypeslib.as_array(buf, shape=(int(count),))
            t = self.torch.from_numpy(a)
            self.dist.reduce(t, 0, group=self.group)
            self.dist.broadcast(t, 0, group=self.group)
            self.calls += 1
            if self.log is not None:
                self.log.append(("a", int(count)))
            return 0
        except Exception as e:   # never let an exception cross the C boundary
            self.error = e
            return 1

    def exchange(self, ctx, n_peers, peers, off, send, recv):
        try:
            n = int(off[n_peers])
            s = self.torch.from_numpy(np.ctypeslib.as_array(send, shape=(n,)))
            r = self.torch.from_numpy(np.ctypeslib.as_array(recv, shape=(n,)))
            reqs = []
            for k in range(n_peers):
                lo, hi, q = int(off[k]), int(off[k + 1]), int(peers[k])
                reqs.append(self.dist.isend(s[lo:hi], q, group=self.group))
                reqs.append(self.dist.irecv(r[lo:hi], q, group=self.group))
            for w in reqs:
                w.wait()
            self.calls += 1
            if self.log is not None:
                self.log.append(("x", n))
            return 0
        except Exception as e:
            self.error = e
            return 1


def broadcast_rccl_id(dist, group=None, make_id=rccl_unique_id):
    """Rank 0 creates the two ncclUniqueIds and every rank receives them -- or every rank raises the SAME error.
    The creation can fail on rank 0 alone (librccl not loadable, ncclGetUniqueId refused): rank 0 must not leave the
    collective the others are waiting in, so a (status, bytes) pair is ALWAYS broadcast and all ranks raise or return
    together, before any further collective (the MPI binding, pfem_mpi.cpp, broadcasts its rc the same way)."""
    from ._lib import ERR_COMM, PfemError
    msg = [None, None]
    if dist.get_rank() == 0:
        try:
            msg = [None, make_id()]
        except Exception as e:      # noqa: BLE001 -- whatever it was, the other ranks must hear of it
            msg = [f"{type(e).__name__}: {e}", None]
    dist.broadcast_object_list(msg, src=0, group=group)
    if msg[0] is not None:
        raise PfemError(ERR_COMM, "pfem_rccl_unique_id on rank 0", msg[0])
    return msg[1]


def attach(solver, dist, torch=None, staged=False, group=None, make_id=rccl_unique_id, peer=False):
    """Wire a solver that already holds its mesh (or, compat path, its pattern) to the process group: neighbour
    plan + communication backend.  ``staged=False``: RCCL inside the library (one rank per GPU).
    ``staged=True``: host hooks over the group (gloo; several ranks may share a GPU); ``group``: a gloo subgroup when
    the default group is not one.  ``peer=True``: peer memory mapped through hipIpc* (ranks that share a GPU, device-side data
    path; the hooks over the group carry the bring-up).  Returns the HostHooks object (staged / peer) or None."""
    rank, world = dist.get_rank(), dist.get_world_size()
    peers, off, gid = plan_for(solver, dist, group)
    hooks = None
    if peer:          # peer memory (hipIpc*): the data path stays on the device; the hooks carry the bring-up
        hooks = HostHooks(dist, torch, group)
        solver.setCommPeer(rank, world, hooks.allreduce, hooks.exchange)
        solver._keep.append(hooks)
    elif staged:
        hooks = HostHooks(dist, torch, group)
        solver.setCommHost(rank, world, hooks.allreduce, hooks.exchange)
        solver._keep.append(hooks)
    else:
        solver.setCommRccl(rank, world, broadcast_rccl_id(dist, group, make_id))
    solver.setNeighbours(peers, off, gid)
    return hooks
