mkdir -p gpurun_out
export COLVO_LIB_PATH=$PWD/coivo_amd/lib/libcolvo_abl.so
COLVO_TRACE=35 CONV_BENCH_ITERS=30 timeout -k 10 300 python tools/bench_conv.py 16 bf16 fwdonly > gpurun_out/r2_trace_conv.log 2>&1
grep -c trace gpurun_out/r2_trace_conv.log
