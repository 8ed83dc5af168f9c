"""Where does a solve's time go between its kernels?  Reads a rocprofv3 --kernel-trace CSV directory and prints, for the
window between the first and the last launch of a marker kernel (default k_pc_update = one per gamg CG iteration):
busy time, idle time, and the idle time by (previous kernel -> next kernel) pair.
usage: trace_gaps.py <dir> [marker_substring] [skip_first_windows]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(n):
    n = n.replace("void ", "").replace("pfem::", "")
    return n.split("(")[0][:48]


def main():
    d = sys.argv[1]
    marker = sys.argv[2] if len(sys.argv) > 2 else "k_pc_update"
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    idx = [i for i, r in enumerate(rows) if marker in r[2]]
    if len(idx) < 3:
        print("marker not found often enough", len(idx)); return
    # iterations = stretches between consecutive markers that are close in time (same solve)
    spans = []
    for a, b in zip(idx, idx[1:]):
        if rows[b][0] - rows[a][0] < 2_000_000:      # < 2 ms apart: same solve (an iteration takes 0.2 - 1.5 ms; assembly + set-up between two solves 3 ms and more)
            spans.append((a, b))
    busy = idle = 0
    pair = defaultdict(lambda: [0, 0])
    per_kernel = defaultdict(lambda: [0, 0])
    for a, b in spans:
        for i in range(a, b):
            s0, e0, n0 = rows[i]
            s1, e1, n1 = rows[i + 1]
            busy += e0 - s0
            per_kernel[short(n0)][0] += 1; per_kernel[short(n0)][1] += e0 - s0
            g = max(0, s1 - e0)
            idle += g
            p = pair[(short(n0), short(n1))]
            p[0] += 1; p[1] += g
    n = len(spans)
    print(f"{n} iterations: busy {busy / n / 1e3:.1f} us, idle {idle / n / 1e3:.1f} us per iteration, {sum(b - a for a, b in spans) / n:.1f} launches")
    print("idle by boundary (us per iteration, count per iteration, us per boundary):")
    for k, v in sorted(pair.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"  {k[0]:48s} -> {k[1]:48s} {v[1] / n / 1e3:8.2f} {v[0] / n:6.2f} {v[1] / v[0] / 1e3:8.2f}")
    print("busy by kernel (us per iteration, launches per iteration, us per launch):")
    for k, v in sorted(per_kernel.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"  {k:48s} {v[1] / n / 1e3:8.2f} {v[0] / n:6.2f} {v[1] / v[0] / 1e3:8.2f}")


if __name__ == "__main__":
    main()
