mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_nets_gpu.py tests/test_conv_gpu.py tests/test_config1_gpu.py -x -q > gpurun_out/r2_tests_13.log 2>&1 || { tail -30 gpurun_out/r2_tests_13.log; exit 1; }
tail -3 gpurun_out/r2_tests_13.log
CONV_BENCH_ITERS=30 timeout -k 10 200 python tools/bench_conv.py 16 bf16 > gpurun_out/r2_bench_conv_z1.log 2>&1 || exit 1
tail -2 gpurun_out/r2_bench_conv_z1.log
timeout -k 10 300 python bench.py > gpurun_out/r2_bench_z1.log 2>&1 || exit 1
tail -1 gpurun_out/r2_bench_z1.log
