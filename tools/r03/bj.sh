#!/bin/bash
cd "$GRAFT_REPO_ROOT"
( timeout 1200 python -m pytest tests/test_gpu_full_size.py -m gpu -x -q 2>&1 | tail -5 )
