#!/bin/bash
# coupled gamg through the RCCL backend (world size 1) + bench contract tests that run several ranks
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_distributed.py -m gpu -x -q -k "rccl or gamg" 2>&1 | tail -5 )
( timeout 2400 python -m pytest tests/test_bench_contract.py -m gpu -x -q --durations=8 2>&1 | tail -16 )
