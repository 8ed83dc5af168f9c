#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; OUT=$GRAFT_REPO_ROOT/gpurun_out
( time timeout 1200 python bench.py --steps 20 --warmup 5 2>$OUT/ak.err | tail -1 ) > $OUT/ak_bench_n1.json 2> $OUT/ak_time.txt
python3 -c "
import json
d=json.load(open('$OUT/ak_bench_n1.json')); c=d['cpu_baseline']; print(d['value'], d['ms_per_step']); print(c['value'], c['cores'], c['sample']); print(c.get('mpi_one_rank_per_core')); print(c.get('openmp_port'))"
cat $OUT/ak_time.txt | tail -4
