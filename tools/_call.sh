set -x
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -q -m gpu > gpurun_out/r2_tests_6.log 2>&1
tail -12 gpurun_out/r2_tests_6.log
timeout -k 10 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r2_bench_3.json 2> gpurun_out/r2_bench_3.err
head -c 600 gpurun_out/r2_bench_3.json
COLVO_NO_DEFER_JOIN=1 timeout -k 10 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline-cfg2 > gpurun_out/r2_bench_3b.json 2> gpurun_out/r2_bench_3b.err
head -c 400 gpurun_out/r2_bench_3b.json
timeout -k 10 300 python tools/cpu_probe.py > gpurun_out/r2_cpu_probe.log 2>&1
tail -5 gpurun_out/r2_cpu_probe.log
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_r2b && mkdir -p gpurun_out/prof_r2b
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r2b/bench_trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline-cfg2 > gpurun_out/prof_r2b/bench_trace.log 2>&1
python tools/timeline.py gpurun_out/prof_r2b/bench_trace/*/*_kernel_trace.csv > gpurun_out/r2_timeline_b.log 2>&1
head -20 gpurun_out/r2_timeline_b.log
