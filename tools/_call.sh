mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 600 python -m pytest tests/test_ddp_gpu.py tests/test_program_gpu.py tests/test_nets_gpu.py tests/test_graph_gpu.py -x -q > gpurun_out/r2_tests_43.log 2>&1 || { tail -30 gpurun_out/r2_tests_43.log; exit 1; }
tail -2 gpurun_out/r2_tests_43.log
