#!/bin/bash
# PMC counters of the conv kernels on the DepthNet layer shapes (tools/bench_conv.py), one rocprofv3 pass per group:
#   bash tools/pmc_conv.sh <tag> [frames=16]   -> gpurun_out/pmc_conv_<tag>/{sq1,sq2,tcc}/...   then tools/pmc_conv_summary.py <tag> [frames]
set -e
TAG=${1:-r2}
FRAMES=${2:-16}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_conv_$TAG
rm -rf $OUT && mkdir -p $OUT
export CONV_BENCH_ITERS=4
run() { # name counters...
  n=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -- python3 tools/bench_conv.py $FRAMES bf16 $PMC_CONV_MODE > $OUT/$n.log 2>&1
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS
run sq2 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_LDS_DATA_FIFO_FULL
# (a TCP/TA group of six counters is refused by the hardware -- "exceeds the capabilities" -- and took the profiler down: not collected)
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum || true
echo done
