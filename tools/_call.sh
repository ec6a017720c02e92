mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_nets_gpu.py tests/test_config1_gpu.py -x -q > gpurun_out/r2_tests_38.log 2>&1 || { tail -30 gpurun_out/r2_tests_38.log; exit 1; }
tail -2 gpurun_out/r2_tests_38.log
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_hd -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/prof_hd.log 2>&1
grep -h "depth_head" gpurun_out/prof_hd/*/*_kernel_stats.csv | cut -c1-130
for n in 3 1 3 1; do
  COLVO_HEAD_WGRAD_ROWS=$n timeout -k 10 300 python bench.py --no-cpu-baseline --steps 80 > gpurun_out/r2_bench_hd_$n.log 2>&1 || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/r2_bench_hd_$n.log").read().strip().split("\n")[-1])
print("head wgrad rows=$n:", d["ms_per_step"], d["ms_per_step_hipevent_median"])
PY
done
