"""Root-cause probe for the round-1 hang: two ranks on ONE GPU, the all-reduce hook bound to ProcessGroupGloo
operating directly on DEVICE tensors on the legacy null stream (what tests/test_distributed.py did in round 1).

  python tools/probe_gloo_device_hang.py [loops] [null|own]

Each loop spawns 2 fresh ranks that solve the 12x10x14 Poisson slab problem with PFEM_CG_CHUNK=1 and log every hook
call (call#, count, seconds spent in all_reduce) to gpurun_out/gloo_probe/loop<k>_rank<r>.log; a rank that sits in one
call for 40 s dumps its stacks (all threads) and exits.  The summary says how many loops hung, in which call, and
whether the two ranks had issued the same call sequence up to there.
"stream" = null: solver on the legacy default stream (round 1); own: solver on its own torch stream.
"""
import faulthandler
import os
import socket
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "gpurun_out", "gloo_probe")


def _worker(rank, world, port, loop, which):
    os.environ["PFEM_CG_CHUNK"] = "1"
    logf = open(os.path.join(OUT, f"loop{loop}_rank{rank}.log"), "w")
    faulthandler.dump_traceback_later(40, exit=True, file=logf)
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pfemfort_amd as pf
    from pfemfort_amd import distributed as PD
    from pfemfort_amd import host as H
    mesh = H.gen_box_tets(-1, 1, 12, -1, 1, 10, -1, 1, 14)
    epid, npid = H.partition_box_slabs(*mesh.box, world)
    dm = H.dof_numbering(mesh.nNode, 1, mesh.bc_node, mesh.bc_dof, mesh.bc_val, world, npid)
    conn_new, xyz_new = H.renumber_mesh(mesh, dm)
    mine = np.nonzero(epid == rank)[0]
    conn_loc = np.ascontiguousarray(conn_new[:, mine])
    edof = H.elem_dof_array(conn_loc, dm.NodeDofArrayNew)
    rs, re = int(dm.row_start[rank]), int(dm.row_end[rank])
    s = pf.PetscSolver().initialise(re - rs, dm.size_global, row_start=rs, device=0)
    s.setTolerances(rtol=1e-10)
    s.uploadMesh(pf.POISSON_TET, conn_loc, xyz_new, edof, dm.solnApplied)
    hook, n_iface = PD.attach(s, dist, torch, torch.device("cuda", 0))        # TorchAllReduce on device tensors
    if which == "own":
        st = torch.cuda.Stream()
        torch.cuda.set_stream(st)
        s.setStream(st.cuda_stream)
    inner = hook.__call__

    def logged(ctx, buf, count, stream):
        t0 = time.time()
        logf.write(f"call {hook.calls + 1} count {int(count)} enter\n"); logf.flush()
        faulthandler.dump_traceback_later(40, exit=True, file=logf)       # re-armed: 40 s inside ONE call
        rc = inner(ctx, buf, count, stream)
        logf.write(f"call {hook.calls} count {int(count)} done {time.time() - t0:.4f}\n"); logf.flush()
        return rc

    s.setComm(rank, world, logged)          # replaces the callback attach() installed (both stay referenced)
    s.buildPattern()
    s.assemble(H.POISSON_ELEMDATA, H.TIMEDATA)
    its, reason, _ = s.factoriseAndSolve()
    logf.write(f"solved its {its} reason {reason}\n"); logf.flush()
    faulthandler.cancel_dump_traceback_later()
    s.free()
    dist.destroy_process_group()


def main():
    loops = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    which = sys.argv[2] if len(sys.argv) > 2 else "null"
    os.makedirs(OUT, exist_ok=True)
    import torch.multiprocessing as mp
    hung = []
    for k in range(loops):
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        try:
            mp.spawn(_worker, args=(2, port, k, which), nprocs=2, join=True)
        except Exception as e:
            hung.append(k)
            print(f"loop {k}: FAILED {type(e).__name__}", flush=True)
    rep = [f"stream={which} loops={loops} failed={len(hung)} {hung}"]
    for k in hung:
        seqs = []
        for r in range(2):
            lines = open(os.path.join(OUT, f"loop{k}_rank{r}.log")).read().splitlines()
            calls = [ln for ln in lines if ln.startswith("call")]
            seqs.append([ln.split()[3] for ln in calls if ln.endswith("enter")])
            rep.append(f"loop {k} rank {r}: last = {calls[-1] if calls else None}; entered {len(seqs[-1])} calls")
        m = min(len(seqs[0]), len(seqs[1]))
        rep.append(f"loop {k}: call sequences identical over the common prefix: {seqs[0][:m] == seqs[1][:m]}; lengths {len(seqs[0])}/{len(seqs[1])}")
    open(os.path.join(OUT, "summary.txt"), "a").write("\n".join(rep) + "\n")
    print("\n".join(rep))


if __name__ == "__main__":
    main()
