#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per dispatch, per kernel."""
import collections
import csv
import glob
import sys


def main(dirs, filt):
    for d in dirs:
        for f in glob.glob(d + "/*/*_counter_collection.csv"):
            acc = collections.defaultdict(lambda: collections.defaultdict(list))
            for r in csv.DictReader(open(f)):
                name = r["Kernel_Name"]
                if filt and filt not in name:
                    continue
                short = name.replace("colvo::(anonymous namespace)::", "").split("(")[0][-40:]
                acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
            for k, cs in acc.items():
                print(f"[{d}] {k}: " + "  ".join(f"{c}={sum(v) / len(v):.4g}" for c, v in sorted(cs.items())) + f"  (n={len(next(iter(cs.values())))})")


if __name__ == "__main__":
    main([a for a in sys.argv[1:] if not a.startswith("--filter=")],
         next((a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--filter=")), "k_warp_loss"))
