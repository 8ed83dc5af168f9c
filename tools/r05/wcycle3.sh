#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "gamg" 2>&1 | tail -15
timeout 600 python -m pytest tests/test_fortran_boundary.py -m gpu -x -q 2>&1 | tail -3
