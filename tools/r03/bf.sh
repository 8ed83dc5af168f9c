#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_distributed.py tests/test_gpu_full_size.py tests/test_golden_drivers.py -m gpu -q -k "gamg or rccl" 2>&1 | tail -5 )
