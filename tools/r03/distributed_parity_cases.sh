#!/bin/bash
# rank-coupled gamg hierarchy: the distributed parity cases (with their iteration counts) + the one-rank cases
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_distributed.py -m gpu -q -s -k "gamg or rccl" 2>&1 | grep -E "^gamg|passed|failed|Error|error|assert" | tail -40 ) > gpurun_out/r03aa_dist.log 2>&1
tail -30 gpurun_out/r03aa_dist.log
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "gamg or renumbering" 2>&1 | tail -5 )
