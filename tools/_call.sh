set -x
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_terms_gpu.py tests/test_warp_loss_gpu.py -q -m gpu > gpurun_out/r2_tests_9.log 2>&1
tail -30 gpurun_out/r2_tests_9.log
