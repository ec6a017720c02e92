mkdir -p gpurun_out
timeout -k 10 300 python bench.py --no-cpu-baseline --no-roofline-cfg2 --full-loss --steps 60 > gpurun_out/r2_bench_full.log 2>&1 || { tail -20 gpurun_out/r2_bench_full.log; exit 1; }
python - <<PY
import json
d=json.loads(open("gpurun_out/r2_bench_full.log").read().strip().split("\n")[-1])
print(d["value"], d["ms_per_step"], d["ms_per_step_hipevent_median"], d["final_loss"])
PY
