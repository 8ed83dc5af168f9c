"""Per-call latency of the transports between TWO ranks that share cuda:0: the peer-memory transport (device-side: mapped
receive boxes + flags, pfem_solver_set_comm_peer) against the host-staged gloo hooks, for the face sizes of SURVEY 8(e)
(62 KB = config 4's 51x51x3 doubles, 1.29 MB = config 5's 401^2 doubles) and a 4-double all-reduce.
    python tools/probe_peer.py > out.json"""
import json
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    import pfemfort_amd as pf
    from pfemfort_amd import distributed as PD
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = {}
    for name in ("peer", "host"):
        s = pf.PetscSolver().initialise(8, 8 * world, row_start=8 * rank, device=0)
        hooks = PD.HostHooks(dist, torch)
        (s.setCommPeer if name == "peer" else s.setCommHost)(rank, world, hooks.allreduce, hooks.exchange)
        assert s.commSelftest(1000) == 0
        for label, count in (("face_62KB", 51 * 51 * 3), ("face_1p29MB", 401 * 401)):
            x, a = s.commBench(count, 200 if name == "peer" else 20)
            res[f"{name}_{label}_us_per_exchange"] = 1e3 * x
            res[f"{name}_allreduce_4_doubles_us"] = 1e3 * a
        res[f"{name}_backend"] = s.commDescribe()["backend"]
        dist.barrier()
        s.free()
    if rank == 0:
        with open(out, "w") as f:
            json.dump(res, f)
    dist.destroy_process_group()


def main():
    import torch.multiprocessing as mp
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    out = "/tmp/probe_peer.json"
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    mp.spawn(worker, args=(world, port, out), nprocs=world, join=True)
    d = json.load(open(out))
    d["what"] = f"{world} ranks sharing one MI355X: peer-memory transport (hipIpc-mapped boxes, flags) against host-staged gloo hooks"
    print(json.dumps(d))


if __name__ == "__main__":
    main()
