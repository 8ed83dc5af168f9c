#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf /tmp/prof_c5
PFEM_AMG_VERBOSE=1 timeout 900 rocprofv3 --kernel-trace -f csv -d /tmp/prof_c5 -- python3 bench.py --cells 400 --steps 1 --warmup 1 --no-cpu-baseline --no-parity-step --no-jacobi-step > gpurun_out/r03t.log 2> gpurun_out/r03t.err
grep "gamg symbolic level 0 " gpurun_out/r03t.err | head -12
python3 - <<'PY'
import csv, glob, collections
mx = collections.defaultdict(lambda: [0, 0.0, 0.0])
rows = []
for f in glob.glob("/tmp/prof_c5/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        m = mx[r["Kernel_Name"][:100]]
        m[0] += 1; m[1] += d; m[2] = max(m[2], d)
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:80]))
for k, v in sorted(mx.items(), key=lambda kv: -kv[1][2])[:12]:
    print(f"{k:100s} calls {v[0]:6d} total {v[1]:10.1f} ms  max {v[2]:10.2f} ms")
rows.sort()
gaps = sorted(((rows[i+1][0] - rows[i][1]) / 1e6, rows[i][2], rows[i+1][2]) for i in range(len(rows) - 1))[-8:]
print("largest gaps between consecutive kernels (ms):")
for g in gaps: print(round(g[0], 1), "|", g[1][:60], "->", g[2][:60])
PY
