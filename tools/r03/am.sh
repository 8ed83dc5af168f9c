#!/bin/bash
# OpenMP port under the container's CPU quota: 64 threads (all it may start) against as many as the quota grants
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python3 - <<'PY'
import sys, json
sys.argv = ["bench.py"]
import bench
for q in (None, 16, 32):
    bench.cpu_quota = (lambda q=q: q)
    real_mpi = bench.cpu_baseline_mpi
    bench.cpu_baseline_mpi = lambda n, rtol: None
    o = bench.cpu_baseline(200, 1e-5, extra_sample=False)
    bench.cpu_baseline_mpi = real_mpi
    print("quota", q, "threads", o["cores"], "asm", round(o["assembly_s"], 2), "solve", round(o["solve_s"], 2), "triad", round(o["host_stream_triad_gbps"]))
bench.cpu_quota = lambda: 16
for ranks in (12, 16, 20, 24):
    bench.physical_cores = lambda r=ranks: r
    bench.cpu_quota = lambda r=ranks: r
    m = bench.cpu_baseline_mpi(200, 1e-5)
    print("mpi ranks", ranks, m["assembly_s"], m["solve_s"], m["binding"])
PY
