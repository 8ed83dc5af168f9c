#!/bin/bash
# per-iteration cost of the coupled gamg cycle over RCCL with the rank as its own neighbour: config 3 box and one rank's share of config 5
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python tools/probe_coupled.py 200 30 > gpurun_out/r03ad_coupled_200.json 2> gpurun_out/r03ad_200.err; tail -3 gpurun_out/r03ad_200.err
timeout 600 python tools/probe_coupled.py 400 30 50 > gpurun_out/r03ad_coupled_400x50.json 2> gpurun_out/r03ad_400.err; tail -3 gpurun_out/r03ad_400.err
python3 - <<'PY'
import json
for f in ("gpurun_out/r03ad_coupled_200.json", "gpurun_out/r03ad_coupled_400x50.json"):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "ERR", e); continue
    for k in ("one_rank_loop", "coupled_replicated_bottom", "coupled_all_levels_distributed"):
        r = d[k]; print(d["cells"], k, r["iterations"], r["reason"], "ms/it", round(r["ms_per_iteration"], 3), "levels", r["levels"], r["distributed_levels"], "host enqueue", round(r["host_enqueue_ms_per_iteration"], 3), "numeric", round(r["numeric_setup_ms"], 2), "sym", round(r["symbolic_setup_ms"], 1))
PY
