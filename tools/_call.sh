mkdir -p gpurun_out
bash tools/collect_profiles.sh r2 && bash tools/pmc_conv.sh r2 && CONV_BENCH_ITERS=30 timeout -k 10 200 python tools/bench_conv.py 16 bf16 > gpurun_out/r2_bench_conv_final.log 2>&1 && CONV_BENCH_HW=512x640 CONV_BENCH_ITERS=8 timeout -k 10 400 python tools/bench_conv.py 64 bf16 > gpurun_out/r2_bench_conv_cfg2.log 2>&1 && timeout -k 10 400 python bench.py > gpurun_out/r2_bench_final.log 2>&1
tail -2 gpurun_out/r2_bench_conv_final.log; tail -2 gpurun_out/r2_bench_conv_cfg2.log; tail -1 gpurun_out/r2_bench_final.log | cut -c1-330
