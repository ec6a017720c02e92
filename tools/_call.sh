mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_conv_gpu.py tests/test_nets_gpu.py tests/test_config1_gpu.py -x -q > gpurun_out/r2_tests_31.log 2>&1 || { tail -30 gpurun_out/r2_tests_31.log; exit 1; }
tail -2 gpurun_out/r2_tests_31.log
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pk -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/prof_pk.log 2>&1
grep -h "pack_nchw\|unpack_nhwc" gpurun_out/prof_pk/*/*_kernel_stats.csv | cut -c1-120
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 60 > gpurun_out/r2_bench_pk.log 2>&1 || exit 1
python - <<PY
import json
d=json.loads(open("gpurun_out/r2_bench_pk.log").read().strip().split("\n")[-1])
print(d["ms_per_step"], d["ms_per_step_hipevent_median"], d["value"])
PY
