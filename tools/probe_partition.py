"""gamg across ranks whose dofs do NOT fill boxes of the lattice (what a METIS partition looks like), at size, with the ranks sharing
the one GPU over gloo host hooks (counts, phase times and hierarchy shapes -- not link times): the same cube cut into z-slabs
("slabs": every rank a box -> bricks with padded borders) and into slabs whose border planes step up halfway along x ("stairs":
no boxes -> bricks split between their owners, amg_split_bricks; PFEM_AMG_SPLIT_BRICKS_OFF=1: the pairing passes of round 5).

    python tools/probe_partition.py [cells=100] [world=3] > out.json        (spawns the ranks itself)
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def part_stairs(mesh, world):
    nz = mesh.box[2]
    x0, x1, z0, z1 = mesh.xyz[0].min(), mesh.xyz[0].max(), mesh.xyz[2].min(), mesh.xyz[2].max()

    def part_of(xyz):
        step = (xyz[0] > 0.5 * (x0 + x1) + 1e-9).astype(np.float64)
        layer = np.floor((xyz[2] - z0) / (z1 - z0) * nz - 1e-9) - step
        return np.clip(np.floor(layer * world / nz), 0, world - 1).astype(np.int32)
    return part_of(mesh.xyz[:, mesh.conn].mean(axis=1)), part_of(mesh.xyz)


def worker(rank, world, port, n, how, env, out_dir):
    import faulthandler
    faulthandler.dump_traceback_later(900, exit=True)
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), **env)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pfemfort_amd as pf
    from pfemfort_amd import distributed as PD
    from pfemfort_amd import host as H
    mesh = H.gen_box_tets(-1, 1, n, -1, 1, n, -1, 1, n)
    epid, npid = part_stairs(mesh, world) if how == "stairs" else H.partition_box_slabs(*mesh.box, world)
    dm = H.dof_numbering(mesh.nNode, 1, mesh.bc_node, mesh.bc_dof, mesh.bc_val, world, npid)
    conn_new, xyz_new = H.renumber_mesh(mesh, dm)
    mine = np.nonzero(epid == rank)[0]
    conn_loc = np.ascontiguousarray(conn_new[:, mine])
    edof = H.elem_dof_array(conn_loc, dm.NodeDofArrayNew)
    rs, re = int(dm.row_start[rank]), int(dm.row_end[rank])
    s = pf.PetscSolver().initialise(re - rs, dm.size_global, row_start=rs, device=0)
    s.setTolerances(rtol=1e-5)
    s.setPreconditioner("gamg")
    s.uploadMesh(pf.POISSON_TET, conn_loc, xyz_new, edof, dm.solnApplied)
    PD.attach(s, dist, torch, staged=True)
    s.buildPattern()
    s.assemble(H.POISSON_ELEMDATA, H.TIMEDATA)
    t0 = time.perf_counter()
    its, reason, _ = s.factoriseAndSolve()
    first = time.perf_counter() - t0
    s.assemble(H.POISSON_ELEMDATA, H.TIMEDATA)
    t0 = time.perf_counter()
    its2, reason2, _ = s.factoriseAndSolve()
    warm = time.perf_counter() - t0
    ai, lay = s.amgInfo(), s.amgLayout()
    json.dump({"rank": rank, "owned_rows": re - rs, "iterations": its2, "reason": reason2, "first_solve_ms_gloo_hooks": first * 1e3, "warm_solve_ms_gloo_hooks": warm * 1e3,
               "symbolic_ms": ai["symbolic_ms"], "numeric_ms": ai["numeric_ms"], "rows_per_level_owned": ai["rows"], "aggregation": s.amgAggregation(),
               "distributed_levels": lay["distributed_levels"], "spmv_form": s.spmvRowGroup(), "spmv_gap_escapes": bool(s.spmvGapEscapes()),
               "value_dictionary": s.spmvValueDictionary()}, open(os.path.join(out_dir, f"{how}_{rank}.json"), "w"))
    s.free()
    dist.destroy_process_group()


def main():
    import socket
    import tempfile
    import torch.multiprocessing as mp
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    out = {"what": __doc__.split("\n\n")[0], "cells": n, "world": world, "head": os.popen(f"git -C {ROOT} rev-parse --short HEAD 2>/dev/null").read().strip() or os.environ.get("PFEM_HEAD", "")}
    for tag, how, env in (("slabs", "slabs", {}), ("stairs_split_bricks", "stairs", {}), ("stairs_pairing_passes_round5", "stairs", {"PFEM_AMG_SPLIT_BRICKS_OFF": "1"})):
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        d = tempfile.mkdtemp()
        mp.spawn(worker, args=(world, port, n, how, env, d), nprocs=world, join=True)
        ranks = [json.load(open(os.path.join(d, f"{how}_{r}.json"))) for r in range(world)]
        out[tag] = {"iterations": ranks[0]["iterations"], "reason": ranks[0]["reason"], "aggregation": ranks[0]["aggregation"],
                    "symbolic_ms_per_rank": [round(r["symbolic_ms"], 2) for r in ranks], "numeric_ms_per_rank": [round(r["numeric_ms"], 2) for r in ranks],
                    "rows_per_level_owned_by_rank": [r["rows_per_level_owned"] for r in ranks], "distributed_levels": ranks[0]["distributed_levels"],
                    "first_solve_ms_gloo_hooks": round(max(r["first_solve_ms_gloo_hooks"] for r in ranks), 1),
                    "warm_solve_ms_gloo_hooks": round(max(r["warm_solve_ms_gloo_hooks"] for r in ranks), 1),
                    "spmv_rows_per_lane": [r["spmv_form"] for r in ranks], "spmv_gap_escapes": [r["spmv_gap_escapes"] for r in ranks],
                    "value_dictionary_entries": [r["value_dictionary"] for r in ranks]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
