#!/bin/bash
# Re-check the defaults of coivo_amd/csrc/tuning.h away from configs[1]: totals of tools/bench_conv.py (forward / dgrad / wgrad over
# the DepthNet stack, us) with one entry changed at a time, at 64 and 128 images of 256x320 (the per-GPU shapes of BASELINE
# configs[3] / [4]) and 64 images of 512x640 (configs[2]).  Runs on the GPU box: bash tools/tuning_check.sh > gpurun_out/tuning_check.txt
export COLVO_DEV=1 CONV_BENCH_ITERS=10
run() {   # label, B, HW, env...
  local label=$1 B=$2 HW=$3; shift 3
  local out
  out=$(env "$@" CONV_BENCH_HW=$HW timeout -k 10 300 python tools/bench_conv.py $B bf16 2>/dev/null | grep "^totals")
  echo "$B x $HW | $label | $out"
}
for shape in "64 256x320" "128 256x320" "64 512x640"; do
  set -- $shape
  run "defaults" $1 $2 COLVO_DEV=1
  run "bn64_min_wgs=512" $1 $2 COLVO_BN64_MIN_WGS=512
  run "bn64_min_wgs=4096" $1 $2 COLVO_BN64_MIN_WGS=4096
  run "wide=1" $1 $2 COLVO_WIDE=1
  run "quad_min_wgs=512" $1 $2 COLVO_QUAD_MIN_WGS=512
  run "quad_min_wgs=100000 (off)" $1 $2 COLVO_QUAD_MIN_WGS=100000
  run "dgrad_up2_min_wgs=0" $1 $2 COLVO_DGRAD_UP2_MIN_WGS=0
  run "lone_max_wgs=256" $1 $2 COLVO_LONE_MAX_WGS=256
  run "up2_bn16_max_wgs=0" $1 $2 COLVO_UP2_BN16_MAX_WGS=0
  run "wgrad_atomic_mb=6" $1 $2 COLVO_WGRAD_ATOMIC_MB=6
  run "wgrad_wg_hi=2048" $1 $2 COLVO_WGRAD_WG_HI=2048
  run "wgrad_mt_max=4" $1 $2 COLVO_WGRAD_MT_MAX=4
done
