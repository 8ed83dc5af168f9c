#!/bin/bash
# the tests added this round (one pass), then the whole GPU tier as the driver runs it
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
timeout 3000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_bench_contract.py -x -q -m gpu -k "lattice_bricks or beam_config4 or default_solver or eight_ranks_is_cut" --durations=8 > $OUT/new_tests.log 2>&1
tail -25 $OUT/new_tests.log
