#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_distributed.py -m gpu -q -s -k "gamg or rccl" 2>&1 | grep -E "^gamg levels elast x3 yslabs gamg:|passed|failed|FAILED|^E " | tail -20 )
