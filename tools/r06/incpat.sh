#!/bin/bash
# round 6: incidence lists as translated patterns -- parity, then the assembly kernels with the table and with the nodes' own records
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "incidence or assembly or gather or renumbering or element" 2>&1 | tail -25 ) > $OUT/incpat_parity.txt 2>&1
tail -5 $OUT/incpat_parity.txt
for V in pat own; do
  if [ $V = own ]; then export PFEM_INC_PATTERNS=0; fi
  ( timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-jacobi-step --no-pmc --no-parity-step 2>$OUT/incpat_$V.err | tail -1 ) > $OUT/incpat_$V.json
  ( timeout 900 python bench.py --workload beam --steps 5 --warmup 2 --no-cpu-baseline --no-jacobi-step --no-pmc --no-parity-step 2>$OUT/incpat_beam_$V.err | tail -1 ) > $OUT/incpat_beam_$V.json
  for f in incpat_$V incpat_beam_$V; do python3 - <<PY
import json
d=json.load(open("$OUT/$f.json")); print("$f", d["ms_per_step"], d["iterations"], d.get("assembly_ms_per_step"), d.get("first_step_ms_including_once_per_pattern_setup"))
PY
  done
done
rm -rf /tmp/prof_ip
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_ip -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity-step --no-jacobi-step --no-pmc > $OUT/incpat_prof.log 2>&1
python tools/summarize_prof.py stats /tmp/prof_ip 80 2>&1 | grep -E "k_gather|k_incpat|k_build_inc" | cut -c1-150
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_ip$C
  timeout 900 rocprofv3 --pmc $C --kernel-include-regex "k_gather" -f csv -d /tmp/prof_ip$C -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity-step --no-jacobi-step --no-pmc > $OUT/incpat_pmc_$C.log 2>&1
  python tools/summarize_prof.py pmc /tmp/prof_ip$C $C 2>&1 | cut -c1-150
done
