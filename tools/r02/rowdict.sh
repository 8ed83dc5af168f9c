#!/bin/bash
# GPU visit: gap table for the row / 3-row SpMV forms -- the cfg-4 beam must not move; the 2x refined beam with and without the table
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out
show() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print(sys.argv[2], {k:d.get(k) for k in ("value","ms_per_step","iterations","ms_per_iteration","assembly_ms_per_step","setup_s_untimed","device_memory_gb")}, d["roofline"]["kernel"][:60], d["roofline"]["avg_launch_ms"], round(d["roofline"]["frac"],3), d["config"]["free_dofs"])
except Exception as e:
    print(sys.argv[2], "no result", e)
PY
}
for rep in 1 2; do
  ( timeout 300 python bench.py --workload beam --steps 3 --warmup 1 --no-cpu-baseline 2>$OUT/beam.err | grep '^{' | tail -1 ) > $OUT/beam_default_$rep.json
  show $OUT/beam_default_$rep.json "beam 50x300x50 run $rep"
done
( timeout 500 python bench.py --workload beam --beam-scale 2 --steps 1 --warmup 0 --no-cpu-baseline 2>$OUT/beam2.err | grep '^{' | tail -1 ) > $OUT/beam_x2_table.json
show $OUT/beam_x2_table.json "beam 100x600x100 with the gap table"
tail -3 $OUT/beam2.err
( PFEM_DEBUG_NO_ROW_GAP_TABLE=1 timeout 500 python bench.py --workload beam --beam-scale 2 --steps 1 --warmup 0 --no-cpu-baseline 2>$OUT/beam2n.err | grep '^{' | tail -1 ) > $OUT/beam_x2_notable.json
show $OUT/beam_x2_notable.json "beam 100x600x100 WITHOUT the table"
