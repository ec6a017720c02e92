set -x
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_warp_loss_gpu.py tests/test_nets_gpu.py tests/test_config1_gpu.py tests/test_graph_gpu.py tests/test_program_gpu.py tests/test_ddp_gpu.py -q -m gpu > gpurun_out/r2_tests_4.log 2>&1
tail -12 gpurun_out/r2_tests_4.log
timeout -k 10 600 python bench.py --steps 30 --warmup 5 > gpurun_out/r2_bench_1.json 2> gpurun_out/r2_bench_1.err
cat gpurun_out/r2_bench_1.json | head -c 3000
