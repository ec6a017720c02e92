mkdir -p gpurun_out
timeout -k 10 400 python bench.py > gpurun_out/r2_bench_final.log 2>&1 || { tail -5 gpurun_out/r2_bench_final.log; exit 1; }
python - <<PY
import json
d=json.loads(open("gpurun_out/r2_bench_final.log").read().strip().split("\n")[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["fwd_us"], d["roofline"]["bwd_us"], d["roofline_cfg2"]["frac"], d["roofline_cfg2"]["fwd_us"], d["roofline_cfg2"]["bwd_us"])
PY
