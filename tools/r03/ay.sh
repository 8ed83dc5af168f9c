#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_distributed.py -m gpu -q -s -k "gamg or rccl" 2>&1 | grep -E "^gamg|passed|failed|FAILED|amg_levels|rows_glob|^E  " | tail -40 )
