#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_fortran_boundary.py -m gpu -x -q -k "gamg" 2>&1 | tail -25 )
