#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
for W in 0 1; do
( PFEM_RBM_RESTRICT_WIDE=$W timeout 1200 python -m pytest tests/test_distributed.py -q -m gpu -k "elast and gamg" 2>&1 | grep -E "^gamg elast|passed|failed|FAILED|Error" ) > $OUT/dist2_w$W.txt 2>&1
done
( timeout 1200 python -m pytest tests/test_distributed.py -q -m gpu -s -k "elast and gamg" 2>&1 | grep -E "^gamg elast|passed|failed|FAILED" ) > $OUT/dist2_s.txt 2>&1
cat $OUT/dist2_w0.txt $OUT/dist2_w1.txt $OUT/dist2_s.txt
