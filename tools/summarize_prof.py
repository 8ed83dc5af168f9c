"""Condense rocprofv3 CSV output into the small text summaries committed under profiles/.

  python tools/summarize_prof.py stats <dir> [rows]     -> per-kernel count / avg / total from *kernel_stats.csv (top 25)
  python tools/summarize_prof.py pmc <dir> <COUNTER>    -> per-kernel average of a PMC counter per dispatch
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def find(d, pat):
    return sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))


def short(name):
    name = name.replace("void ", "")
    return name[:90]


def stats(d, top=25):
    files = find(d, "*kernel_stats.csv")
    if not files:
        # fall back to the raw trace
        agg = defaultdict(lambda: [0, 0.0])
        for f in find(d, "*kernel_trace.csv"):
            for r in csv.DictReader(open(f)):
                dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
                a = agg[r["Kernel_Name"]]
                a[0] += 1; a[1] += dur
        rows = [(k, v[0], v[1], v[1] / v[0]) for k, v in agg.items()]
    else:
        rows = []
        for f in files:
            for r in csv.DictReader(open(f)):
                rows.append((r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]), float(r["AverageNs"])))
    rows.sort(key=lambda r: -r[2])
    tot = sum(r[2] for r in rows) or 1.0
    print(f"{'kernel':92s} {'calls':>8s} {'avg_us':>10s} {'total_ms':>10s} {'%':>6s}")
    for name, calls, total, avg in rows[:top]:
        print(f"{short(name):92s} {calls:8d} {avg / 1e3:10.2f} {total / 1e6:10.2f} {100 * total / tot:6.2f}")


def pmc(d, counter):
    agg = defaultdict(lambda: [0, 0.0])
    for f in find(d, "*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            a = agg[r["Kernel_Name"]]
            a[0] += 1; a[1] += float(r["Counter_Value"])
    print(f"{'kernel':92s} {'dispatches':>10s} {counter + ' avg/dispatch':>28s}")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{short(k):92s} {v[0]:10d} {v[1] / v[0]:28.1f}")


def pmc_each(d, counter, needle):
    """every dispatch of the kernels whose name contains `needle`, in dispatch order (first steps differ from later ones)"""
    rows = []
    for f in find(d, "*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and needle in r["Kernel_Name"]:
                rows.append((int(r["Dispatch_Id"]), short(r["Kernel_Name"])[:60], float(r["Counter_Value"])))
    for i, k, v in sorted(rows):
        print(f"{counter} dispatch {i:6d} {k:60s} {v:16.1f}")


if __name__ == "__main__":
    if sys.argv[1] == "pmc_each":
        pmc_each(sys.argv[2], sys.argv[3], sys.argv[4])
        sys.exit(0)
    if sys.argv[1] == "stats":
        stats(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 25)
    else:
        pmc(sys.argv[2], sys.argv[3])
