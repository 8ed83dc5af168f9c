#!/bin/bash
# round 3, visit A: the new tests (any-axis device generator, multi-rank devgen, bench self-launch) + a bench line
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "any_axis or device_generated" 2>&1 | tail -15 ) > $OUT/r03a_parity.log 2>&1
( timeout 1200 python -m pytest tests/test_distributed.py -m gpu -x -q -k "devgen or yslabs or xslabs or rccl" 2>&1 | tail -15 ) > $OUT/r03a_dist.log 2>&1
( timeout 2400 python -m pytest tests/test_bench_contract.py -m gpu -x -q --durations=10 2>&1 | tail -25 ) > $OUT/r03a_bench_tests.log 2>&1
( timeout 600 python bench.py 2>$OUT/r03a_bench.err | tail -1 ) > $OUT/r03a_bench.json
tail -5 $OUT/r03a_parity.log $OUT/r03a_dist.log; tail -25 $OUT/r03a_bench_tests.log; cut -c1-600 $OUT/r03a_bench.json
