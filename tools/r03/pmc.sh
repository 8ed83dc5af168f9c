#!/bin/bash
# round 3: PMC passes (HBM traffic) of the default bench -- separate passes, --pmc only (MI355X_MICROARCH.md, HBM section)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
RE="k_spmv|k_cg_|k_pc_|k_amg_spmv|k_amg_cheb|k_amg_restrict|k_amg_prolong|k_amg_galerkin|k_amg_diag|k_gather"
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_$C
  timeout 900 rocprofv3 --pmc $C --kernel-include-regex "$RE" -f csv -d /tmp/prof_$C -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity-step > $OUT/r03_pmc_$C.log 2>&1
  python tools/summarize_prof.py pmc /tmp/prof_$C $C > $OUT/r03_rocprof_pmc_$C.txt 2>&1
done
python3 - <<PY > $OUT/spmv_pmc_traffic.json
import json
def val(path, pat):
    for line in open(path):
        if pat in line: return float(line.split()[-1])
out = {"entries": []}
for pat, nnz in (("k_spmvr<true, false>", 117260947),):
    out["entries"].append({"kernel": "pfem::" + pat + " (the CG SpMV of the default bench, config 3)", "nnz": nnz,
                           "FETCH_SIZE_KB": val("$OUT/r03_rocprof_pmc_FETCH_SIZE.txt", pat), "WRITE_SIZE_KB": val("$OUT/r03_rocprof_pmc_WRITE_SIZE.txt", pat),
                           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, tools/gpu_r03_pmc.sh (round 3)"})
print(json.dumps(out, indent=1))
PY
head -30 $OUT/r03_rocprof_pmc_FETCH_SIZE.txt; head -30 $OUT/r03_rocprof_pmc_WRITE_SIZE.txt; cat $OUT/spmv_pmc_traffic.json
