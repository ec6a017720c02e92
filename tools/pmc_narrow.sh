#!/bin/bash
# PMC counters of the two fused full-resolution kernels (tools/bench_narrow.py), one rocprofv3 pass per group:
#   bash tools/pmc_narrow.sh <tag> [frames=64]   -> gpurun_out/pmc_narrow_<tag>/{sq1,sq2,sq3}/ + summary.txt
set -e
TAG=${1:-r5}
FRAMES=${2:-64}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_narrow_$TAG
rm -rf $OUT && mkdir -p $OUT
run() { # name counters...
  n=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -- python3 tools/bench_narrow.py $FRAMES 6 > $OUT/$n.log 2>&1
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS
run sq2 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_LDS_DATA_FIFO_FULL
run sq3 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_VMEM_WR || true
python3 - "$OUT" <<'PY' > $OUT/summary.txt
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for grp in ("sq1", "sq2", "sq3"):
    for f in glob.glob(f"{out}/{grp}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "bwd16" in n or "fwd16" in n:
                k = n.replace("colvo::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(f"{out}/{grp}/*/*_kernel_trace.csv"):
        if grp != "sq1":
            continue
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "bwd16" in n or "fwd16" in n:
                k = n.replace("colvo::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
                acc[k]["duration_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for k, c in acc.items():
    print(k)
    for name in sorted(c):
        v = c[name]
        print(f"   {name:34s} {sum(v) / len(v):16.1f}   (n={len(v)})")
PY
cat $OUT/summary.txt
