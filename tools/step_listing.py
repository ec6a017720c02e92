#!/usr/bin/env python3
"""Every kernel of one steady-state step from a rocprofv3 kernel trace of bench.py, in start order:
   python tools/step_listing.py <..._kernel_trace.csv> [step_index_from_end=10 | median]
columns: start offset (us), duration (us), queue, gap since the previous kernel on the same queue, name."""
import csv
import sys


def short(n):
    return n.replace("colvo::(anonymous namespace)::", "").replace("void ", "")[:86]


def main():
    path = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2] != "median" else 10
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"]))
    rows.sort()
    ends = [e for s, e, q, n in rows if "k_adam_pack" in n]
    if len(sys.argv) > 2 and sys.argv[2] == "median":      # the step of median length among the last 20 (a host stall makes outliers)
        cand = sorted(range(len(ends) - 20, len(ends)), key=lambda i: ends[i] - ends[i - 1])
        i = cand[len(cand) // 2]
        t0, t1 = ends[i - 1], ends[i]
    else:
        t0, t1 = ends[-back - 1], ends[-back]
    step = [r for r in rows if r[0] >= t0 and r[1] <= t1 + 1]
    print(f"step window {1e-3 * (t1 - t0):.1f} us, {len(step)} kernels")
    last = {}
    for s, e, q, n in step:
        gap = (s - last[q]) * 1e-3 if q in last else float("nan")
        last[q] = e
        print(f"{(s - t0) * 1e-3:8.1f} {(e - s) * 1e-3:7.1f}  q{q}  gap {gap:7.1f}  {short(n)}")


if __name__ == "__main__":
    main()
