#!/bin/bash
# round 6: the geometry-for-nothing lab build again, now that the kernel reads its incidence records from the pattern table
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
( make -C pfemfort_amd/csrc LAB=1 EXTRA=-DPFEM_LAB_FREE_GEOMETRY 2>&1 | tail -1 ) > $OUT/geom2_build.log
for V in ship lab; do
  rm -rf /tmp/prof_g_$V
  if [ $V = lab ]; then export PFEM_AMD_LIB=$GRAFT_REPO_ROOT/pfemfort_amd/libpfem_amd_lab.so; fi
  timeout 900 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_g_$V -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity-step --no-jacobi-step --no-pmc --pc jacobi --rtol 1e-2 > $OUT/geom2_$V.log 2>&1
  echo "== $V"; python tools/summarize_prof.py stats /tmp/prof_g_$V 60 2>&1 | grep -E "k_gather" | cut -c1-150
done
