#!/usr/bin/env python3
"""Step timeline from a rocprofv3 --kernel-trace CSV of bench.py: per-step GPU busy time (union of kernel intervals),
idle time, per-queue sums and the per-kernel-family breakdown, averaged over the steady-state steps.
   tools/timeline.py <kernel_trace.csv> [marker-kernel-substring=k_adam] [skip_steps=8]"""
import collections
import csv
import re
import sys


def family(name):
    m = re.search(r"(k_\w+)(<[^>]*>)?", name)
    if m:
        return m.group(1) + (m.group(2) or "")
    if "elementwise" in name or "Fill" in name:
        return "torch elementwise"
    return name[:40]


def main(path, marker="k_adam", skip=8):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]))
    rows.sort()
    # a step ends with the LAST marker kernel of a run of marker kernels (Adam runs once per network)
    ends = [i for i, r in enumerate(rows) if marker in r[2]]
    step_ends = [ends[j] for j in range(len(ends)) if j + 1 == len(ends) or ends[j + 1] - ends[j] > 3]
    steps = []
    for a, b in zip(step_ends[skip:-1], step_ends[skip + 1:]):
        steps.append(rows[a + 1:b + 1])
    if not steps:
        print("no steps found"); return
    tot_span = tot_busy = 0.0
    fam = collections.Counter(); cnt = collections.Counter(); queues = collections.Counter()
    for st in steps:
        t0, t1 = st[0][0], max(r[1] for r in st)
        tot_span += t1 - t0
        cur_s, cur_e, busy = None, None, 0
        for s, e, n, q in st:
            fam[family(n)] += e - s; cnt[family(n)] += 1; queues[q] += e - s
            if cur_e is None or s > cur_e:
                if cur_e is not None: busy += cur_e - cur_s
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
        busy += cur_e - cur_s
        tot_busy += busy
    n = len(steps)
    print(f"{n} steps: span {tot_span / n / 1e3:.1f} us  busy(union) {tot_busy / n / 1e3:.1f} us  idle {(tot_span - tot_busy) / n / 1e3:.1f} us"
          f"  sum of kernel durations {sum(fam.values()) / n / 1e3:.1f} us  launches/step {sum(cnt.values()) / n:.0f}")
    print("per queue (us/step):", {q: round(v / n / 1e3, 1) for q, v in queues.items()})
    for k, v in fam.most_common(30):
        print(f"  {v / n / 1e3:8.1f} us  x{cnt[k] / n:5.1f}  {k}")


if __name__ == "__main__":
    main(sys.argv[1], *(sys.argv[2:3]), *(map(int, sys.argv[3:4])))
