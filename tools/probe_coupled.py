"""Timing probe of the rank-coupled gamg cycle on ONE GPU (no second GPU is available to the builder): as in
tools/probe_overlap.py the rank names ITSELF as its only neighbour (PFEM_DEBUG_SELF_PEER) and shares the dofs of the two outer
free z-planes of its box with itself, so every exchange of the cycle is a grouped ncclSend/ncclRecv (to self) and every
all-reduce an ncclAllReduce, on the streams a real multi-GPU run uses.  The sums are wrong by construction (own partials are
added twice: the operator is no longer symmetric and CG may stop early), so only times are reported: per iteration against
the one-rank gamg loop on the same box, which has the fused coarse-level kernels and no exchange at all.

    python tools/probe_coupled.py [cells=200] [iterations=30] [cells_z=cells] > out.json
    (cells=400 cells_z=50: one rank's share of BASELINE config 5 -- 7.8 M rows, two faces of 399^2 dofs)
"""
import json
import os
import socket
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import faulthandler
    faulthandler.enable()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    its = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    nz = int(sys.argv[3]) if len(sys.argv) > 3 else n
    import torch.distributed as dist
    import pfemfort_amd as pf
    from pfemfort_amd import host as H
    from pfemfort_amd.solver import rccl_unique_id
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=0, world_size=1)
    sz = H.box_slab_sizes(n, n, nz)
    N = sz["size_global"]
    s = pf.PetscSolver().initialise(N, N)
    s.setTolerances(rtol=1e-30, maxits=its)            # a fixed number of iterations
    s.setPreconditioner("gamg")
    s.generateBoxMesh(pf.POISSON_TET, -1.0, 1.0, n, -1.0, 1.0, n, -1.0, -1.0 + 2.0 * nz / n, nz)
    s.buildPattern()
    s.assemble(H.POISSON_ELEMDATA, H.TIMEDATA)

    def run(tag):
        s.factoriseAndSolve()
        i, r, _ = s.factoriseAndSolve()
        t = s.timings()
        ai, lay = s.amgInfo(), s.amgLayout()
        return {"iterations": i, "reason": r, "ms_per_iteration": (t["solve_ms"] - ai["numeric_ms"]) / max(i, 1), "numeric_setup_ms": ai["numeric_ms"],
                "symbolic_setup_ms": ai["symbolic_ms"], "levels": ai["levels"], "distributed_levels": lay["distributed_levels"], "coupled": lay["coupled"],
                "rows_per_level": ai["rows"], "host_enqueue_ms_per_iteration": t["host_enqueue_ms"] / max(t["host_enqueued_iterations"], 1),
                "host_ms_inside_rccl_calls_per_iteration": t["host_comm_ms"] / max(t["host_enqueued_iterations"], 1)}
    single = run("single")
    m = n - 1
    gid = np.concatenate([np.arange(m * m), np.arange(N - m * m, N)]).astype(np.int64)
    os.environ["PFEM_DEBUG_SELF_PEER"] = "1"
    os.environ["PFEM_FORCE_MULTI"] = "1"
    s.setCommRccl(0, 1, rccl_unique_id())
    s.setNeighbours(np.array([0], np.int32), np.array([0, len(gid)], np.int64), gid)
    assert s.commSelftest(4096) == 0
    out = {"what": "rank-coupled gamg cycle with the rank as its own neighbour over RCCL (timing only; sums wrong by construction)",
           "cells": [n, n, nz], "free_dofs": N, "one_rank_loop": single}
    lay_keys = ("exchanges_per_cycle", "allreduces_per_cycle")
    for tag, env in (("coupled_replicated_bottom", None), ("coupled_replicated_from_150000_rows", "150000"),
                     ("coupled_replicated_from_1000000_rows", "1000000"), ("coupled_all_levels_distributed", "0")):
        if env is None:
            os.environ.pop("PFEM_AMG_REPLICATE_ROWS", None)
        else:
            os.environ["PFEM_AMG_REPLICATE_ROWS"] = env
        s.setNeighbours(np.array([0], np.int32), np.array([0, len(gid)], np.int64), gid)      # (drops the hierarchy: rebuilt with the new setting)
        out[tag] = run(tag)
        out[tag].update({k: s.amgLayout()[k] for k in lay_keys})
    print(json.dumps(out))
    s.free()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
