mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_rccl
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_rccl -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline-cfg2 --rccl-single > gpurun_out/prof_rccl.log 2>&1
ls gpurun_out/prof_rccl/*/ | head
