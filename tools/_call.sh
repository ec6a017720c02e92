mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_full -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline-cfg2 --full-loss > gpurun_out/prof_full.log 2>&1
ls gpurun_out/prof_full/*/
