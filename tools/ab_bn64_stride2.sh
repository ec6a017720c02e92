#!/bin/bash
# A/B of 64-wide channel tiles on the stride-2 forward layers; run from the repo root on the GPU box.
O=gpurun_out/r6s; mkdir -p $O
export PYTHONPATH=$PWD COLVO_DEV=1
for B in 64 128 16 32; do
  timeout -k 10 200 python tools/bench_conv.py $B bf16 fwdonly > $O/base_$B.log 2>&1 || exit 1
  COLVO_BN64_MIN_WGS=600 timeout -k 10 200 python tools/bench_conv.py $B bf16 fwdonly > $O/bn64_$B.log 2>&1 || exit 1
  echo "== B=$B (base | bn64_min_wgs=600)"
  paste <(grep -E "^enc[2-5]a" $O/base_$B.log | awk '{print $1, $2, $3, $6}') <(grep -E "^enc[2-5]a" $O/bn64_$B.log | awk '{print $6}')
done
