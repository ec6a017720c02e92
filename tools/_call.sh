mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_conv_gpu.py tests/test_nets_gpu.py tests/test_large_gpu.py -q -m gpu -x > gpurun_out/r2_tests_12.log 2>&1
tail -3 gpurun_out/r2_tests_12.log
timeout -k 10 300 python tools/bench_conv.py 16 bf16 > gpurun_out/r2_bench_conv_y1.log 2>&1
tail -n 2 gpurun_out/r2_bench_conv_y1.log
timeout -k 10 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline-cfg2 > gpurun_out/r2_bench_4.json 2> gpurun_out/r2_bench_4.err
head -c 400 gpurun_out/r2_bench_4.json
