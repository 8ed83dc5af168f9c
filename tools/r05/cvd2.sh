#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_full_size.py -m gpu -x -q -k "value_codes" 2>&1 | tail -15
timeout 1200 python -m pytest tests/test_bench_contract.py -m gpu -x -q 2>&1 | tail -3
