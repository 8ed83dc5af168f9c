#!/bin/bash
# bench.py --jitter 0.2 at configs 3 and 4 (the lines under profiles/r05/bench_cfg*_jitter.json)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
( timeout 900 python bench.py --jitter 0.2 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 ) > $OUT/final_bench_cfg3_jitter.json
( timeout 900 python bench.py --jitter 0.2 --workload beam --steps 3 --warmup 2 --no-cpu-baseline --no-jacobi-step 2>/dev/null | tail -1 ) > $OUT/final_bench_cfg4_jitter.json
for f in cfg3_jitter cfg4_jitter; do python3 -c "
import json; d=json.load(open('$OUT/final_bench_$f.json')); p=d['preconditioner']
print('$f its', d['iterations'], 'warm', round(d['ms_per_step'],2), 'first', round(d['first_step_ms_including_once_per_pattern_setup'],1), 'rows', p['rows_per_level'], 'complexity', round(p['operator_complexity'],2), 'value dict', d['roofline']['value_dictionary_entries'])"; done
