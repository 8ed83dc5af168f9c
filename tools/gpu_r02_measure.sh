#!/bin/bash
# One GPU visit for the numbers: bench (default = headline config), rocprofv3 kernel stats + PMC (separate passes),
# beam bench, summaries -> gpurun_out/ (copy what is to be judged into profiles/)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
TAG=${1:-r02}
( timeout 900 python bench.py 2>$OUT/bench_$TAG.err | tail -1 ) > $OUT/bench_$TAG.json
rm -rf /tmp/prof_stats /tmp/prof_fetch /tmp/prof_write
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity-step > $OUT/prof_stats_$TAG.log 2>&1
python tools/summarize_prof.py stats /tmp/prof_stats > $OUT/rocprof_kernel_stats_$TAG.txt 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "k_spmv|k_cg_|k_assemble|k_gather" -f csv -d /tmp/prof_fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity-step > $OUT/prof_fetch_$TAG.log 2>&1
python tools/summarize_prof.py pmc /tmp/prof_fetch FETCH_SIZE > $OUT/rocprof_pmc_fetch_$TAG.txt 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "k_spmv|k_cg_|k_assemble|k_gather" -f csv -d /tmp/prof_write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity-step > $OUT/prof_write_$TAG.log 2>&1
python tools/summarize_prof.py pmc /tmp/prof_write WRITE_SIZE > $OUT/rocprof_pmc_write_$TAG.txt 2>&1
python3 - <<PY > $OUT/spmv_pmc_traffic.json
import re, json
def val(path):
    for line in open(path):
        if "k_spmv" in line and "<true>" in line:
            return float(line.split()[-1])
nnz = json.load(open("$OUT/bench_$TAG.json"))["roofline"]["nnz"]
print(json.dumps({"kernel": "the CG SpMV the solver selected (pfem::k_spmvr<true> / k_spmvg<true> / k_spmv16<true> / k_spmv<true>)", "nnz": nnz, "FETCH_SIZE_KB": val("$OUT/rocprof_pmc_fetch_$TAG.txt"),
                  "WRITE_SIZE_KB": val("$OUT/rocprof_pmc_write_$TAG.txt"), "source": "rocprofv3 --pmc, separate passes, tools/gpu_r02_measure.sh $TAG"}))
PY
( timeout 600 python bench.py --workload beam --no-cpu-baseline 2>$OUT/bench_beam_$TAG.err | tail -1 ) > $OUT/bench_beam_$TAG.json
( timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --backend gloo --same-device --cells 100 --steps 2 --warmup 1 2>$OUT/bench_2ranks_samedev_$TAG.err | grep '^{' | tail -1 ) > $OUT/bench_2ranks_samedev_$TAG.json
cat $OUT/bench_$TAG.json; head -14 $OUT/rocprof_kernel_stats_$TAG.txt; cat $OUT/rocprof_pmc_fetch_$TAG.txt $OUT/rocprof_pmc_write_$TAG.txt; cat $OUT/bench_beam_$TAG.json; cat $OUT/bench_2ranks_samedev_$TAG.json
