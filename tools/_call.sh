mkdir -p gpurun_out
timeout -k 10 400 python bench.py > gpurun_out/r2_bench_final.log 2>&1 || exit 1
python - <<PY
import json
d=json.loads(open("gpurun_out/r2_bench_final.log").read().strip().split("\n")[-1])
print(d["value"], d["ms_per_step"], d["ms_per_step_hipevent_median"], d["ms_per_step_hipevent_max"], d["steps"], d["warmup"])
PY
