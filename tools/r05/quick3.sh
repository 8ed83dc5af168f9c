#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_distributed.py -m gpu -x -q -k "gamg or elast or rbm or beam" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rbm or rigid or elast or cook or gamg" 2>&1 | tail -2
