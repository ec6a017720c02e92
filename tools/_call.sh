mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_program_gpu.py tests/test_nets_gpu.py tests/test_ddp_gpu.py tests/test_graph_gpu.py -x -q > gpurun_out/r2_tests_15.log 2>&1 || { tail -30 gpurun_out/r2_tests_15.log; exit 1; }
tail -2 gpurun_out/r2_tests_15.log
for n in 1 2 1 2; do
COLVO_SIDE_STREAMS=$n timeout -k 10 300 python bench.py --no-cpu-baseline --steps 60 > gpurun_out/r2_bench_side$n.log 2>&1 || exit 1
python - <<PY
import json
d=json.loads(open("gpurun_out/r2_bench_side$n.log").read().strip().split("\n")[-1])
print("side streams $n:", d["ms_per_step"], d["ms_per_step_hipevent_median"])
PY
done
