#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out
rm -rf /tmp/trace_ov
timeout 600 rocprofv3 --kernel-trace -f csv -d /tmp/trace_ov -- python3 tools/probe_overlap.py 200 60 > $OUT/trace_overlap.log 2>&1
F=$(find /tmp/trace_ov -name "*kernel_trace.csv" | head -1)
python3 - "$F" > $OUT/trace_overlap_timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last 40% of the trace = the multi-rank solves; print 3 iterations around the middle of that part
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "k_pack_send" in n]
print("columns:", list(rows[0].keys()))
if idx:
    a = idx[len(idx) // 2]
    b = idx[len(idx) // 2 + 3]
    t0 = int(rows[a - 1]["Start_Timestamp"])
    for r in rows[a - 1:b + 1]:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        print(f"{s/1000:9.1f} {e/1000:9.1f} {(e-s)/1000:8.1f} us  q={r.get('Queue_Id','?'):>3}  {r['Kernel_Name'][:70]}")
PY
cat $OUT/trace_overlap_timeline.txt | head -80; tail -3 $OUT/trace_overlap.log
