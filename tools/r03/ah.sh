#!/bin/bash
# knob sweep on the config-4 beam (scale 1.8): Chebyshev degrees and the smoothing interval
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
: > gpurun_out/r03ah_beam_knobs.txt
for fd in 1 2; do for cd in 2 3; do for er in 4 8 16; do
  export PFEM_AMG_FINE_DEGREE=$fd PFEM_AMG_CHEB_DEGREE=$cd PFEM_AMG_EIG_RATIO=$er
  timeout 600 python bench.py --workload beam --steps 2 --warmup 1 --no-cpu-baseline --no-parity-step --no-jacobi-step 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.readline()); print('fine_degree $fd cheb_degree $cd eig_ratio $er :', d['iterations'], 'iterations', round(d['ms_per_step'],1), 'ms')" >> gpurun_out/r03ah_beam_knobs.txt
done; done; done
cat gpurun_out/r03ah_beam_knobs.txt
