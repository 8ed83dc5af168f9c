#!/usr/bin/env python3
"""Copy what tools/r06/final.sh left in gpurun_out/ into profiles/r06/ (every file named after what it holds; the commit it was
taken at is in final_head.txt and inside every probe's JSON) and refresh the two replay files bench.py reads when rocprofv3 is not
on the box (profiles/single_gpu_reference.json, profiles/spmv_pmc_traffic.json).  Run here, after the lease."""
import json
import os
import re
import shutil

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles", "r06")
os.makedirs(P, exist_ok=True)
head = open(os.path.join(G, "final_head.txt")).read().strip()
cp = {"final_bench_n1.json": "bench_n1.json", "final_bench_beam.json": "bench_beam.json",
      "final_bench_n1_fp64_values.json": "bench_n1_fp64_values_PFEM_SPMV_VALDICT_0.json",
      "final_bench_cfg5_single_gpu.json": "bench_cfg5_single_gpu.json", "final_bench_cfg2_100cube.json": "bench_cfg2_100cube.json",
      "final_bench_cfg5_8ranks_sharing_one_gpu_gloo.json": "bench_cfg5_8ranks_sharing_one_gpu_gloo.json",
      "final_bench_beam_8ranks_sharing_one_gpu_gloo.json": "bench_beam_8ranks_sharing_one_gpu_gloo.json",
      "final_bench_2ranks_same_device_transports_ab.json": "bench_2ranks_same_device_transports_ab.json",
      "final_bench_cfg3_jitter.json": "bench_cfg3_jitter.json", "final_bench_cfg4_jitter.json": "bench_cfg4_jitter.json",
      "final_bench_numbering_rcb8.json": "bench_numbering_rcb8.json",
      "final_rocprofv3_kernel_stats.txt": "rocprofv3_kernel_stats.txt", "final_rocprofv3_kernel_stats_beam.txt": "rocprofv3_kernel_stats_beam.txt",
      "final_rocprofv3_kernel_stats_jacobi_loop.txt": "rocprofv3_kernel_stats_jacobi_loop.txt",
      "final_rocprofv3_pmc_FETCH_SIZE.txt": "rocprofv3_pmc_FETCH_SIZE.txt", "final_rocprofv3_pmc_WRITE_SIZE.txt": "rocprofv3_pmc_WRITE_SIZE.txt",
      "final_rocprofv3_pmc_beam_FETCH_SIZE.txt": "rocprofv3_pmc_beam_FETCH_SIZE.txt", "final_rocprofv3_pmc_beam_WRITE_SIZE.txt": "rocprofv3_pmc_beam_WRITE_SIZE.txt",
      "final_kernel_timeline_gamg_loop.txt": "kernel_timeline_gamg_loop.txt", "final_kernel_phases_of_a_step.txt": "kernel_phases_of_a_step.txt",
      "final_gather_and_spmv_sq_counters.txt": "gather_and_spmv_sq_counters.txt",
      "final_coupled_cycle_rccl_self_peer.json": "coupled_cycle_rccl_self_peer.json"}
for a, b in cp.items():
    src = os.path.join(G, a)
    if not os.path.exists(src):
        print("missing", a)
        continue
    if b.endswith(".txt"):
        open(os.path.join(P, b), "w").write(f"(taken at commit {head}, tools/r06/final.sh)\n" + open(src).read())
    else:
        shutil.copy(src, os.path.join(P, b))
n1 = json.load(open(os.path.join(P, "bench_n1.json")))
c5 = json.load(open(os.path.join(P, "bench_cfg5_single_gpu.json")))
ref_path = os.path.join(R, "profiles", "single_gpu_reference.json")
ref = json.load(open(ref_path))


def upd(key, d, src, jac=False):
    e = ref[key]
    if jac:
        j = d["jacobi_step"]
        e.update(ms_per_iteration=j["ms_per_step"] / j["iterations"], ms_per_step=j["ms_per_step"], iterations=j["iterations"])
    else:
        e.update(ms_per_iteration=d["ms_per_iteration"], ms_per_step=d["ms_per_step"], iterations=d["iterations"])
    e["free_dofs"] = d["config"]["free_dofs"]
    e["source"] = src


upd("cfg3_200cube_gamg", n1, f"profiles/r06/bench_n1.json (python bench.py --steps 20 --warmup 5, builder lease, round 6, commit {head})")
upd("cfg3_200cube_jacobi", n1, "profiles/r06/bench_n1.json: jacobi_step (same run)", True)
upd("cfg5_400cube_gamg", c5, f"profiles/r06/bench_cfg5_single_gpu.json (python bench.py --cells 400 --steps 2 --warmup 1, builder lease, round 6, commit {head})")
upd("cfg5_400cube_jacobi", c5, "profiles/r06/bench_cfg5_single_gpu.json: jacobi_step (same run)", True)
json.dump(ref, open(ref_path, "w"), indent=2)


def pmc(fname, kernel_rx):
    for ln in open(os.path.join(P, fname)):
        if re.search(kernel_rx, ln):
            return float(ln.split()[-1])
    return None


def trace(fname, kernel_rx):
    for ln in open(os.path.join(P, fname)):
        if re.search(kernel_rx, ln):
            return float(ln.split()[-3])
    return None


tp = os.path.join(R, "profiles", "spmv_pmc_traffic.json")
doc = json.load(open(tp))
for e in doc["entries"]:
    if e.get("value_dictionary") and e.get("workload") != "beam":
        e["FETCH_SIZE_KB"] = pmc("rocprofv3_pmc_FETCH_SIZE.txt", r"k_spmvr_vd<true")
        e["WRITE_SIZE_KB"] = pmc("rocprofv3_pmc_WRITE_SIZE.txt", r"k_spmvr_vd<true")
        e["kernel_trace_avg_us"]["multigrid_loop"] = trace("rocprofv3_kernel_stats.txt", r"k_spmvr_vd<true")
        e["kernel_trace_avg_us"]["jacobi_loop"] = trace("rocprofv3_kernel_stats_jacobi_loop.txt", r"k_spmvr_vd<true")
        e["source"] = f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, tools/r06/final.sh at commit {head}: profiles/r06/rocprofv3_pmc_*.txt"
    if e.get("value_dictionary") and e.get("workload") == "beam":
        e["FETCH_SIZE_KB"] = pmc("rocprofv3_pmc_beam_FETCH_SIZE.txt", r"k_spmvg_vd<true")
        e["WRITE_SIZE_KB"] = pmc("rocprofv3_pmc_beam_WRITE_SIZE.txt", r"k_spmvg_vd<true")
        e["kernel_trace_avg_us"]["multigrid_loop"] = trace("rocprofv3_kernel_stats_beam.txt", r"k_spmvg_vd<true")
        e["source"] = f"rocprofv3 --pmc, separate passes, tools/r06/final.sh at commit {head}: profiles/r06/rocprofv3_pmc_beam_*.txt"
json.dump(doc, open(tp, "w"), indent=1)
for f in ("bench_n1", "bench_n1_fp64_values_PFEM_SPMV_VALDICT_0", "bench_beam", "bench_cfg5_single_gpu", "bench_cfg2_100cube"):
    d = json.load(open(os.path.join(P, f + ".json")))
    r = d["roofline"]
    print(f, round(d["ms_per_step"], 2), "ms warm,", round(d["first_step_ms_including_once_per_pattern_setup"], 1), "cold, jacobi",
          round((d.get("jacobi_step") or {}).get("ms_per_step") or 0, 1), "spmv us", round(r["avg_launch_ms"] * 1e3, 1), "frac", round(r["frac"], 3),
          "algorithmic", round(r["algorithmic_frac"], 3), "traffic", r["traffic"], (r["traffic_source"] or "")[:40])
