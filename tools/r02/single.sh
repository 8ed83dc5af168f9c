#!/bin/bash
# GPU visit: the single-reduction form of the CG (tests, then timings against the two-reduction loop on the same box)
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q -k "single" 2>&1 | tail -30 ) > $OUT/pytest_single.log 2>&1
tail -8 $OUT/pytest_single.log
for sr in 0 1; do
  for ov in 0 1; do
    ( PFEM_CG_SINGLE_REDUCTION=$sr PFEM_MULTI_OVERLAP=$ov timeout 300 python tools/probe_overlap.py 200 200 2>$OUT/probe_sr.err | grep '^{' | tail -1 ) > $OUT/probe_single${sr}_overlap${ov}.json
    echo "single_reduction=$sr overlap=$ov"; cat $OUT/probe_single${sr}_overlap${ov}.json; echo
  done
done
for sr in "" "--single-reduction"; do
  ( timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-parity-step $sr 2>$OUT/bench_sr.err | grep '^{' | tail -1 ) > $OUT/bench_single_${sr:+on}.json
  python - "$OUT/bench_single_${sr:+on}.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print({k:d[k] for k in ("value","ms_per_step","iterations","ms_per_iteration","solve_ms_per_step")}, d["config"]["solver"][:40])
PY
done
# small systems: latency-bound regime (30^3, 50^3), stream launches in both forms
for n in 30 50 100; do
  for sr in 0 1; do
    ( PFEM_CG_GRAPH=0 PFEM_CG_SINGLE_REDUCTION=$sr timeout 300 python bench.py --cells $n --steps 5 --warmup 2 --no-cpu-baseline --no-parity-step 2>>$OUT/bench_sr.err | grep '^{' | tail -1 ) > $OUT/bench_n${n}_single$sr.json
    python - "$OUT/bench_n${n}_single$sr.json" $n $sr <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print("n", sys.argv[2], "single", sys.argv[3], {k:d[k] for k in ("iterations","ms_per_iteration","solve_ms_per_step")})
PY
  done
done
