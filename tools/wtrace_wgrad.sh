#!/bin/bash
# Phase stamps of the weight-gradient kernel (developer tool).  Builds coivo_amd/lib/libcolvo_wtrace.so = the production objects with
# wgrad.hip recompiled under -DCOLVO_WTRACE, then (on the GPU box) runs the conv stack and prints, per layer, the mean workgroup's
# set-up / staging-store / compute / flush times (shader-clock stamps taken behind barriers, csrc/wgrad.hip).
#   bash tools/wtrace_wgrad.sh build            (here: cross-compile)
#   bash tools/wtrace_wgrad.sh run [B]          (on the GPU box)
set -e
cd "$(dirname "$0")/.."
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -DNDEBUG -fno-slp-vectorize -Wno-unused-function"
# build [variant]: COLVO_RT_VARIANT ablation bits of wgrad_rt.hip -> libcolvo_wtrace_v<variant>.so (run: WTRACE_VARIANT=<variant>)
V=${WTRACE_VARIANT:-}
if [ "$1" = build ] && [ -n "$2" ]; then
    /opt/rocm/bin/hipcc $FLAGS -mllvm -amdgpu-mfma-vgpr-form -DCOLVO_WTRACE -DCOLVO_RT_VARIANT=$2 -c coivo_amd/csrc/wgrad_rt.hip -o coivo_amd/lib/obj/wgrad_rt_wtrace_v$2.o
    objs=$(ls coivo_amd/lib/obj/*.o | grep -v '/wgrad.o$' | grep -v '/wgrad_rt.o$' | grep -v _wtrace | grep -v conv_abl.o)
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o coivo_amd/lib/libcolvo_wtrace_v$2.so $objs coivo_amd/lib/obj/wgrad_wtrace.o coivo_amd/lib/obj/wgrad_rt_wtrace_v$2.o
    echo built coivo_amd/lib/libcolvo_wtrace_v$2.so
elif [ "$1" = build ]; then
    python -m coivo_amd.build >/dev/null
    /opt/rocm/bin/hipcc $FLAGS -DCOLVO_WTRACE -c coivo_amd/csrc/wgrad.hip -o coivo_amd/lib/obj/wgrad_wtrace.o
    /opt/rocm/bin/hipcc $FLAGS -mllvm -amdgpu-mfma-vgpr-form -DCOLVO_WTRACE -c coivo_amd/csrc/wgrad_rt.hip -o coivo_amd/lib/obj/wgrad_rt_wtrace.o
    objs=$(ls coivo_amd/lib/obj/*.o | grep -v '/wgrad.o$' | grep -v '/wgrad_rt.o$' | grep -v _wtrace.o | grep -v conv_abl.o)
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o coivo_amd/lib/libcolvo_wtrace.so $objs coivo_amd/lib/obj/wgrad_wtrace.o coivo_amd/lib/obj/wgrad_rt_wtrace.o
    echo built coivo_amd/lib/libcolvo_wtrace.so
else
    B=${2:-16}
    export COLVO_LIB_PATH=$PWD/coivo_amd/lib/libcolvo_wtrace${V:+_v$V}.so
    # bench_conv launches every kernel 1 + CONV_BENCH_ITERS times (iters < 10): print every 6th launch = the last of each layer
    CONV_BENCH_ITERS=5 COLVO_WTRACE=6 python tools/bench_conv.py $B bf16 ${3:-} 2>&1 | grep -v amdgpu.ids
fi
