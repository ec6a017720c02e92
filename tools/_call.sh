mkdir -p gpurun_out
CONV_BENCH_HW=512x640 CONV_BENCH_ITERS=10 timeout -k 10 400 python tools/bench_conv.py 64 bf16 > gpurun_out/r2_bench_conv_cfg2.log 2>&1 || { tail -5 gpurun_out/r2_bench_conv_cfg2.log; exit 1; }
cat gpurun_out/r2_bench_conv_cfg2.log | cut -c1-110
