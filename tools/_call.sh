mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_conv_gpu.py -x -q > gpurun_out/r2_tests_30.log 2>&1 || { tail -30 gpurun_out/r2_tests_30.log; exit 1; }
tail -2 gpurun_out/r2_tests_30.log
CONV_BENCH_ITERS=30 timeout -k 10 200 python tools/bench_conv.py 16 bf16 > gpurun_out/r2_bench_conv_q2.log 2>&1 || exit 1
tail -2 gpurun_out/r2_bench_conv_q2.log | head -1
CONV_BENCH_HW=512x640 CONV_BENCH_ITERS=8 timeout -k 10 400 python tools/bench_conv.py 64 bf16 > gpurun_out/r2_bench_conv_cfg2.log 2>&1 || exit 1
tail -2 gpurun_out/r2_bench_conv_cfg2.log
