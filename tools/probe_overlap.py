"""Timing probe of the multi-rank iteration pipeline on ONE GPU (no second GPU is available to the builder): the
rank names ITSELF as its only neighbour (PFEM_DEBUG_SELF_PEER) and shares the dofs of the two outer free z-planes of
the 200^3 box with itself, so every iteration runs boundary slices -> pack -> grouped ncclSend/ncclRecv (to self) on
the communication stream under the interior slices -> ncclAllReduce -> rank-ordered unpack -> update -> ncclAllReduce
-> direction, exactly as with real neighbours.  The sums are wrong by construction (the own partial is added twice), so
only times are reported: per-iteration time against the single-rank loop, time of the exchange and of the all-reduces
on the communication stream, and how long the compute stream waited for them.

    python tools/probe_overlap.py [cells=200] [iterations=200] [cells_z=cells] > out.json
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
    its = int(sys.argv[2]) if len(sys.argv) > 2 else 200
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
    s.generateBoxMesh(pf.POISSON_TET, -1.0, 1.0, n, -1.0, 1.0, n, -1.0, -1.0 + 2.0 * nz / n, nz)
    s.buildPattern()
    s.assemble(H.POISSON_ELEMDATA, H.TIMEDATA)
    s.profileSpmv(32)
    s.factoriseAndSolve()
    i1, r1, _ = s.factoriseAndSolve()
    t1 = s.timings()
    single = {"iterations": i1, "reason": r1, "ms_per_iteration": t1["solve_ms"] / max(i1, 1),
              "host_enqueue_ms_per_iteration": t1["host_enqueue_ms"] / max(t1["host_enqueued_iterations"], 1),
              "spmv_ms": t1["spmv_ms_total"] / max(t1["spmv_launches"], 1) - t1["event_overhead_ms"]}
    # the plan: one neighbour (this rank), sharing the first and the last free z-plane
    m = n - 1
    gid = np.concatenate([np.arange(m * m), np.arange(N - m * m, N)]).astype(np.int64)
    os.environ["PFEM_DEBUG_SELF_PEER"] = "1"
    os.environ["PFEM_FORCE_MULTI"] = "1"
    s.setCommRccl(0, 1, rccl_unique_id())
    s.setNeighbours(np.array([0], np.int32), np.array([0, len(gid)], np.int64), gid)
    assert s.commSelftest(4096) == 0
    s.factoriseAndSolve()
    i2, r2, _ = s.factoriseAndSolve()
    t2 = s.timings()
    c = max(t2["comm_samples"], 1)
    info = s.commInfo()
    out = {"what": "multi-rank iteration pipeline with the rank as its own neighbour (timing only; sums wrong by construction)",
           "cells": [n, n, nz], "free_dofs": N, "spmv_rows_per_lane_and_gap_table": [s.spmvRowGroup(), s.spmvGapTable()], "single_rank_loop": single,
           "multi_rank_loop": {"iterations": i2, "reason": r2, "ms_per_iteration": t2["solve_ms"] / max(i2, 1),
                               "spmv_boundary_plus_interior_ms": t2["spmv_ms_total"] / max(t2["spmv_launches"], 1) - 2 * t2["event_overhead_ms"],
                               "exchange_ms_on_comm_stream": t2["iface_ms_total"] / c,
                               "two_allreduces_ms_on_comm_stream": t2["scalar_ms_total"] / c,
                               "compute_stream_waited_ms": t2["exposed_ms_total"] / c, "samples": t2["comm_samples"],
                               "bytes_per_exchange": 8 * info["doubles_per_exchange"],
                               "boundary_slices": info["boundary_slices"], "slices": info["total_slices"],
                               "iterations_replayed_from_graph": t2["graph_iterations"],
                               "host_enqueue_ms_per_iteration": t2["host_enqueue_ms"] / max(t2["host_enqueued_iterations"], 1),
                               "host_ms_inside_rccl_calls_per_iteration": t2["host_comm_ms"] / max(t2["host_enqueued_iterations"], 1)}}
    print(json.dumps(out))
    s.free()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
