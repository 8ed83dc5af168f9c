#!/bin/bash
# A/B kernel timings: product lib vs lab lib (built with EXTRA flags) under rocprofv3 --stats
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for v in A B; do
  if [ $v = B ]; then export PFEM_AMD_LIB=$GRAFT_REPO_ROOT/pfemfort_amd/libpfem_amd_lab.so; fi
  rm -rf /tmp/ab_$v
  timeout 300 rocprofv3 --kernel-trace --stats -f csv -d /tmp/ab_$v -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /tmp/ab_$v.log 2>&1
  echo "== variant $v"; tail -1 /tmp/ab_$v.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'solve', d['solve_ms_per_step'], 'asm', d['assembly_ms_per_step'], 'its', d['iterations'])"
  python3 tools/summarize_prof.py stats /tmp/ab_$v | head -7
done
