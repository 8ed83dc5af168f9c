#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
PFEM_AMG_VERBOSE=1 timeout 600 python bench.py --workload beam --steps 1 --warmup 1 --no-jacobi-step 2> gpurun_out/r03y_beam_verbose.err | cut -c1-120
grep "gamg symbolic level [01] " gpurun_out/r03y_beam_verbose.err | head -30
