#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_bench_contract.py -m gpu -x -q -k "gamg or one_json_line or two_ranks" 2>&1 | tail -4 )
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-jacobi-step --no-parity-step 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['preconditioner']['levels_paired_on_the_lattice'], d['preconditioner']['levels'], d['config']['solver'][:160])"
