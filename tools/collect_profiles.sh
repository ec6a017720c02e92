#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun); outputs under gpurun_out/prof_rN/.
#   bash tools/collect_profiles.sh r1
set -e
R=${1:-r1}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$R
mkdir -p $OUT
# 1) per-kernel time of the bench command itself (--graph off: under the tracer the host is slow enough for the one-GPU default
#    `--graph best` to pick the replay; the untraced default run launches eagerly, and that is what the profile should show)
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --graph off > $OUT/bench_trace.log 2>&1
# 1b) the same with the widened objective as the timed step
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/full_trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline-cfg2 --full-loss --graph off > $OUT/full_trace.log 2>&1
# 2) HBM traffic of the fused loss kernels at BASELINE configs[2] and configs[1] shapes: separate --pmc passes
#    (FETCH_SIZE and WRITE_SIZE do not fit one pass), k_adam in the same process as the known-byte calibration
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 tools/traffic_probe.py > $OUT/pmc_$C.log 2>&1
done
timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $OUT/pmc_tcc -- python3 tools/traffic_probe.py > $OUT/pmc_tcc.log 2>&1 || true
echo done
