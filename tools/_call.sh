mkdir -p gpurun_out
timeout -k 10 300 python tools/host_profile.py 200 > gpurun_out/r2_host_profile2.log 2>&1 || { tail -20 gpurun_out/r2_host_profile2.log; exit 1; }
grep -v "^$" gpurun_out/r2_host_profile2.log | sed -n 2,40p
timeout -k 10 300 python tools/cpu_probe.py 60 > gpurun_out/r2_cpu_probe3.log 2>&1; grep rep gpurun_out/r2_cpu_probe3.log
timeout -k 10 600 python -m pytest tests/test_nets_gpu.py tests/test_program_gpu.py tests/test_ddp_gpu.py tests/test_graph_gpu.py -x -q 2>&1 | tail -2
