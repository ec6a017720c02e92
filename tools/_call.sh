mkdir -p gpurun_out
export COLVO_NO_PERSIST=1
for W in 0 1; do
  echo "######## COLVO_WIDE=$W" 
  COLVO_WIDE=$W bash tools/ablate_conv.sh run 16
done > gpurun_out/r2_ablate_conv.log 2>&1
tail -3 gpurun_out/r2_ablate_conv.log
