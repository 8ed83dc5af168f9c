#!/usr/bin/env python3
"""Copy what tools/r05/final.sh left in gpurun_out/ into profiles/r05/ and refresh the two replay files bench.py reads
(profiles/single_gpu_reference.json, profiles/spmv_pmc_traffic.json) from it.  Run here, after the lease."""
import json, os, re, shutil
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles", "r05")
cp = {"final_bench_n1.json": "bench_n1.json", "final_bench_beam.json": "bench_beam.json",
      "final_bench_n1_fp64_values.json": "bench_n1_fp64_values_PFEM_SPMV_VALDICT_0.json",
      "final_bench_beam_fp64_values.json": "bench_beam_fp64_values_PFEM_SPMV_VALDICT_0.json",
      "final_bench_cfg5_single_gpu.json": "bench_cfg5_single_gpu.json", "final_bench_cfg2_100cube.json": "bench_cfg2_100cube.json",
      "final_bench_cfg5_8ranks_sharing_one_gpu_gloo.json": "bench_cfg5_8ranks_sharing_one_gpu_gloo.json",
      "final_rocprofv3_kernel_stats.txt": "rocprofv3_kernel_stats.txt", "final_rocprofv3_kernel_stats_beam.txt": "rocprofv3_kernel_stats_beam.txt",
      "final_rocprofv3_kernel_stats_jacobi_loop.txt": "rocprofv3_kernel_stats_jacobi_loop.txt",
      "final_rocprofv3_pmc_FETCH_SIZE.txt": "rocprofv3_pmc_FETCH_SIZE.txt", "final_rocprofv3_pmc_WRITE_SIZE.txt": "rocprofv3_pmc_WRITE_SIZE.txt",
      "final_rocprofv3_pmc_beam_FETCH_SIZE.txt": "rocprofv3_pmc_beam_FETCH_SIZE.txt", "final_rocprofv3_pmc_beam_WRITE_SIZE.txt": "rocprofv3_pmc_beam_WRITE_SIZE.txt",
      "final_kernel_timeline_gamg_loop.txt": "kernel_timeline_gamg_loop.txt"}
for a, b in cp.items():
    shutil.copy(os.path.join(G, a), os.path.join(P, b))
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


upd("cfg3_200cube_gamg", n1, "profiles/r05/bench_n1.json (python bench.py --steps 20 --warmup 5, builder lease, round 5, value dictionaries in the SpMVs)")
upd("cfg3_200cube_jacobi", n1, "profiles/r05/bench_n1.json: jacobi_step (same run)", True)
upd("cfg5_400cube_gamg", c5, "profiles/r05/bench_cfg5_single_gpu.json (python bench.py --cells 400 --steps 2 --warmup 1, builder lease, round 5, value dictionaries in the SpMVs)")
upd("cfg5_400cube_jacobi", c5, "profiles/r05/bench_cfg5_single_gpu.json: jacobi_step (same run)", True)
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
    if e.get("value_dictionary") and e.get("workload") == "beam":
        e["FETCH_SIZE_KB"] = pmc("rocprofv3_pmc_beam_FETCH_SIZE.txt", r"k_spmvg_vd<true")
        e["WRITE_SIZE_KB"] = pmc("rocprofv3_pmc_beam_WRITE_SIZE.txt", r"k_spmvg_vd<true")
        e["kernel_trace_avg_us"]["multigrid_loop"] = trace("rocprofv3_kernel_stats_beam.txt", r"k_spmvg_vd<true")
json.dump(doc, open(tp, "w"), indent=1)
for f in ("bench_n1", "bench_n1_fp64_values_PFEM_SPMV_VALDICT_0", "bench_beam", "bench_cfg5_single_gpu", "bench_cfg2_100cube"):
    d = json.load(open(os.path.join(P, f + ".json")))
    print(f, round(d["ms_per_step"], 2), "ms warm,", round(d["first_step_ms_including_once_per_pattern_setup"], 1), "cold, jacobi",
          round((d.get("jacobi_step") or {}).get("ms_per_step") or 0, 1), "spmv us", round(d["roofline"]["avg_launch_ms"] * 1e3, 1), "frac", round(d["roofline"]["frac"], 3),
          "hbm_frac", round(d["roofline"]["hbm_frac"], 3))
