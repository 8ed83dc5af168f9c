#!/bin/bash
# One GPU visit for the numbers: bench (default = headline config), rocprofv3 kernel stats + PMC (separate passes) for the
# headline config and the beam, summaries -> gpurun_out/ (copy what is to be judged into profiles/)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
TAG=${1:-r02}
pmc() {   # $1 = name, rest = bench flags
  local name=$1; shift
  rm -rf /tmp/prof_fetch /tmp/prof_write
  timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "k_spmv|k_cg_|k_assemble|k_gather" -f csv -d /tmp/prof_fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity-step "$@" > $OUT/prof_fetch_${name}_$TAG.log 2>&1
  python tools/summarize_prof.py pmc /tmp/prof_fetch FETCH_SIZE > $OUT/rocprof_pmc_fetch_${name}_$TAG.txt 2>&1
  timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "k_spmv|k_cg_|k_assemble|k_gather" -f csv -d /tmp/prof_write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity-step "$@" > $OUT/prof_write_${name}_$TAG.log 2>&1
  python tools/summarize_prof.py pmc /tmp/prof_write WRITE_SIZE > $OUT/rocprof_pmc_write_${name}_$TAG.txt 2>&1
}
( timeout 900 python bench.py 2>$OUT/bench_$TAG.err | grep '^{' | tail -1 ) > $OUT/bench_$TAG.json
( timeout 600 python bench.py --workload beam --no-cpu-baseline 2>$OUT/bench_beam_$TAG.err | grep '^{' | tail -1 ) > $OUT/bench_beam_$TAG.json
rm -rf /tmp/prof_stats
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity-step > $OUT/prof_stats_$TAG.log 2>&1
python tools/summarize_prof.py stats /tmp/prof_stats > $OUT/rocprof_kernel_stats_$TAG.txt 2>&1
pmc poisson
pmc beam --workload beam
python3 - <<PY > $OUT/spmv_pmc_traffic.json
import json
def val(path):
    for line in open(path):
        if "k_spmv" in line and "<true" in line:
            return float(line.split()[-1])
ent = []
for name, bench in (("poisson", "$OUT/bench_$TAG.json"), ("beam", "$OUT/bench_beam_$TAG.json")):
    nnz = json.load(open(bench))["roofline"]["nnz"]
    ent.append({"workload": name, "kernel": "the CG SpMV the solver selected for it", "nnz": nnz,
                "FETCH_SIZE_KB": val(f"$OUT/rocprof_pmc_fetch_{name}_$TAG.txt"), "WRITE_SIZE_KB": val(f"$OUT/rocprof_pmc_write_{name}_$TAG.txt"),
                "source": "rocprofv3 --pmc, separate passes, tools/gpu_r02_measure.sh $TAG"})
print(json.dumps({"entries": ent}))
PY
cat $OUT/bench_$TAG.json; head -14 $OUT/rocprof_kernel_stats_$TAG.txt; cat $OUT/rocprof_pmc_fetch_*_$TAG.txt $OUT/rocprof_pmc_write_*_$TAG.txt; cat $OUT/bench_beam_$TAG.json; cat $OUT/spmv_pmc_traffic.json
