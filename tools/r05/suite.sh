#!/bin/bash
# the whole GPU suite, as the driver runs it at round end
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
t0=$(date +%s)
timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/suite_tail.txt
echo "suite wall $(( $(date +%s) - t0 )) s" | tee -a gpurun_out/suite_tail.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a gpurun_out/suite_tail.txt
