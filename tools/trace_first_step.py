"""The cold first step in a rocprofv3 --kernel-trace directory: from the first assembly kernel (k_gather*) to the last kernel
before the next assembly: wall, busy, the largest gaps with their neighbours, busy time by kernel.
usage: trace_first_step.py <dir> [which_step=0]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(n):
    n = n.replace("void ", "").replace("pfem::", "")
    return n.split("(")[0][:60]


def main():
    d = sys.argv[1]
    which = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if "k_gather" in r[2]]
    if len(starts) <= which:
        print("no such step"); return
    a = starts[which]
    b = starts[which + 1] if which + 1 < len(starts) else len(rows)
    # (the step ends with the last kernel of its solve: cut trailing kernels that start > 20 ms after their predecessor)
    seg = rows[a:b]
    cut = len(seg)
    for i in range(1, len(seg)):
        if seg[i][0] - seg[i - 1][1] > 20_000_000:
            cut = i; break
    seg = seg[:cut]
    wall = seg[-1][1] - seg[0][0]
    busy = sum(e - s for s, e, _ in seg)
    print(f"step {which}: {len(seg)} launches, wall {wall / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms, idle {(wall - busy) / 1e6:.3f} ms")
    gaps = sorted(((seg[i + 1][0] - seg[i][1], i) for i in range(len(seg) - 1)), reverse=True)[:25]
    print("largest gaps (us): after kernel -> before kernel, at ms from the start of the step")
    for g, i in gaps:
        print(f"  {g / 1e3:9.1f}  {short(seg[i][2]):60s} -> {short(seg[i + 1][2]):60s} at {(seg[i][1] - seg[0][0]) / 1e6:8.3f}")
    per = defaultdict(lambda: [0, 0])
    for s, e, n in seg:
        per[short(n)][0] += 1; per[short(n)][1] += e - s
    print("busy by kernel (ms, launches):")
    for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])[:30]:
        print(f"  {k:60s} {v[1] / 1e6:8.3f} {v[0]:6d}")


if __name__ == "__main__":
    main()
