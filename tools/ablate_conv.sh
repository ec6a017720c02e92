#!/bin/bash
# Phase ablation of k_conv3x3 (developer tool).  Builds coivo_amd/lib/libcolvo_abl.so = the production objects with
# conv.hip recompiled under -DCOLVO_ABLATE, then (on the GPU box) times the conv stack with phases switched off:
#   COLVO_ABL bits: 1 no MFMA | 2 no LDS fragment reads | 4 no global loads | 8 no LDS staging stores | 16 no barriers
#                   | 32 no epilogue
# Results are wrong by construction; only the times mean anything.
#   bash tools/ablate_conv.sh build            (here: cross-compile)
#   bash tools/ablate_conv.sh run [B]          (on the GPU box)
set -e
cd "$(dirname "$0")/.."
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -DNDEBUG -fno-slp-vectorize -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form"
if [ "$1" = build ]; then
    python -m coivo_amd.build >/dev/null
    /opt/rocm/bin/hipcc $FLAGS -DCOLVO_ABLATE -c coivo_amd/csrc/conv.hip -o coivo_amd/lib/obj/conv_abl.o
    objs=$(ls coivo_amd/lib/obj/*.o | grep -v '/conv.o$' | grep -v conv_abl.o)
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o coivo_amd/lib/libcolvo_abl.so $objs coivo_amd/lib/obj/conv_abl.o
    echo built coivo_amd/lib/libcolvo_abl.so
else
    B=${2:-16}
    export COLVO_LIB_PATH=$PWD/coivo_amd/lib/libcolvo_abl.so
    for abl in 0 1 2 3 4 8 12 16 28 32 63; do
        echo "== COLVO_ABL=$abl"
        COLVO_ABL=$abl python tools/bench_conv.py $B bf16 fwdonly 2>/dev/null | grep -v amdgpu.ids
    done
fi
