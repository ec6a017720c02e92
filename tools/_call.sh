mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_conv_gpu.py -x -q > gpurun_out/r2_tests_27.log 2>&1 || { tail -30 gpurun_out/r2_tests_27.log; exit 1; }
tail -2 gpurun_out/r2_tests_27.log
timeout -k 10 900 python -m pytest tests/test_nets_gpu.py tests/test_config1_gpu.py tests/test_large_gpu.py -x -q > gpurun_out/r2_tests_28.log 2>&1 || { tail -30 gpurun_out/r2_tests_28.log; exit 1; }
tail -2 gpurun_out/r2_tests_28.log
for n in 1 0 1 0; do
  if [ $n = 1 ]; then export COLVO_NO_DGRAD_UP2=1; else unset COLVO_NO_DGRAD_UP2; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 60 > gpurun_out/r2_bench_du_$n.log 2>&1 || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/r2_bench_du_$n.log").read().strip().split("\n")[-1])
print("no_dgrad_up2=$n:", d["ms_per_step"], d["ms_per_step_hipevent_median"])
PY
done
