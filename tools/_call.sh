mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_conv_gpu.py -x -q > gpurun_out/r2_tests_33.log 2>&1 || { tail -30 gpurun_out/r2_tests_33.log; exit 1; }
tail -2 gpurun_out/r2_tests_33.log
CONV_BENCH_ITERS=30 timeout -k 10 200 python tools/bench_conv.py 16 bf16 > gpurun_out/r2_bench_conv_ng2.log 2>&1 || exit 1
grep "^up1\|totals" gpurun_out/r2_bench_conv_ng2.log | cut -c1-8,40-100
