#!/usr/bin/env python3
"""Timeline of one steady-state training step from a rocprofv3 kernel trace of bench.py:
   python tools/step_timeline.py <..._kernel_trace.csv> [step_index_from_end=20]
Prints, per HW queue, busy time and the gaps between consecutive kernels, the union-busy time of the GPU and the longest
kernels/gaps.  A step = from the end of one step's last k_adam launch to the end of the next step's."""
import csv
import sys
from collections import defaultdict


def short(n):
    n = n.replace("colvo::(anonymous namespace)::", "").replace("void ", "")
    return n[:70]


def main():
    path = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"]))
    rows.sort()
    # step boundaries: the fused update k_adam_pack (one launch per step, round 3 on) -- or, in traces of older builds, every second
    # k_adam launch (one per arena)
    pack_ends = [e for s, e, q, n in rows if "k_adam_pack" in n]
    adam_ends = [e for s, e, q, n in rows if "k_adam" in n]
    bounds = pack_ends if pack_ends else adam_ends[1::2]
    t0, t1 = bounds[-back - 1], bounds[-back]
    step = [r for r in rows if r[0] >= t0 and r[1] <= t1 + 1]
    print(f"step window {1e-3 * (t1 - t0):.1f} us, {len(step)} kernels")
    perq = defaultdict(list)
    for r in step:
        perq[r[2]].append(r)
    for q, rs in sorted(perq.items()):
        busy = sum(e - s for s, e, _, _ in rs)
        gaps = [(rs[i + 1][0] - rs[i][1], short(rs[i][3]), short(rs[i + 1][3])) for i in range(len(rs) - 1)]
        small = [g for g in gaps if 0 <= g[0] < 20000]
        print(f"queue {q}: {len(rs)} kernels, busy {busy * 1e-3:.1f} us, span {(rs[-1][1] - rs[0][0]) * 1e-3:.1f} us, "
              f"back-to-back gaps (<20 us): n={len(small)} sum={sum(g[0] for g in small) * 1e-3:.1f} us "
              f"median={sorted(g[0] for g in small)[len(small) // 2] if small else 0} ns")
    # union busy
    ev = sorted((s, e) for s, e, _, _ in step)
    union, cur_s, cur_e = 0, ev[0][0], ev[0][1]
    idle = []
    for s, e in ev[1:]:
        if s > cur_e:
            union += cur_e - cur_s
            idle.append((s - cur_e, cur_e - t0))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    union += cur_e - cur_s
    print(f"GPU busy (union over queues) {union * 1e-3:.1f} us = {100.0 * union / (t1 - t0):.1f} % of the step; "
          f"idle gaps: n={len(idle)} sum={sum(g for g, _ in idle) * 1e-3:.1f} us")
    for g, at in sorted(idle, reverse=True)[:8]:
        print(f"   idle {g * 1e-3:6.2f} us at +{at * 1e-3:7.1f} us")
    if len(sys.argv) > 3:
        for s, e, q, n in step:
            print(f"{(s - t0) * 1e-3:8.1f} {(e - s) * 1e-3:6.1f} q{q} {short(n)}")


if __name__ == "__main__":
    main()
