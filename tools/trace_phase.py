"""One step of the default bench, kernel by kernel: from the last TYPICAL launch of a kernel (default k_gather_poisson_tet4) to the first
`stop` kernel (default k_cg_start) after it: name, start offset, duration, gap to the launch before -- where the iteration-
independent part of a step goes (assembly, value codes, numeric set-up of the multigrid).

    python tools/trace_phase.py <rocprofv3 csv dir> [first=k_gather_poisson_tet4] [stop=k_cg_start]
"""
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    first = sys.argv[2] if len(sys.argv) > 2 else "k_gather_poisson_tet4"
    stop = sys.argv[3] if len(sys.argv) > 3 else "k_cg_start"
    rows = []
    for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").split("(")[0]))
    rows.sort()
    idx = [i for i, r in enumerate(rows) if first in r[2]]
    if not idx:
        print("no launch of", first)
        return
    # (the last launch whose duration is within 15 % of the median of that kernel's launches: an outlier -- the first step's
    # variant, a launch that caught the tracer's own work -- does not stand for "a step")
    durs = sorted(rows[i][1] - rows[i][0] for i in idx)
    med = durs[len(durs) // 2]
    typical = [i for i in idx if abs((rows[i][1] - rows[i][0]) - med) <= 0.15 * med]
    i0 = typical[-1] if typical else idx[-1]
    t0 = rows[i0][0]
    prev_end = rows[i0][0]
    total = 0.0
    print(f"{'kernel':70s} {'start_us':>10s} {'dur_us':>9s} {'gap_us':>8s}")
    for s, e, n in rows[i0:]:
        print(f"{n[:70]:70s} {(s - t0) / 1e3:10.1f} {(e - s) / 1e3:9.2f} {(s - prev_end) / 1e3:8.2f}")
        total += (e - s) / 1e3
        prev_end = e
        if stop in n:
            break
    print(f"busy {total:.1f} us, span {(prev_end - t0) / 1e3:.1f} us")


if __name__ == "__main__":
    main()
