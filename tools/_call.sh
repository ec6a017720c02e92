mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_conv_gpu.py tests/test_nets_gpu.py -x -q > gpurun_out/r2_tests_41.log 2>&1 || { tail -30 gpurun_out/r2_tests_41.log; exit 1; }
tail -2 gpurun_out/r2_tests_41.log
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 80 > gpurun_out/r2_bench_rf2.log 2>&1 || exit 1
python - <<PY
import json
d=json.loads(open("gpurun_out/r2_bench_rf2.log").read().strip().split("\n")[-1])
print(d["ms_per_step"], d["ms_per_step_hipevent_median"], d["value"])
PY
