#!/bin/bash
# End-of-round-6 sweep of the weight-gradient grid entries IN the step at configs[1]; run from the repo root on the GPU box.
O=gpurun_out/r6r; mkdir -p $O
export COLVO_DEV=1
run() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --config 1 --no-cpu-baseline --no-roofline-cfg2 --no-side-measurements --steps 80 --warmup 10 --graph off > $O/$tag.json 2>$O/$tag.err || { tail -3 $O/$tag.err; return 1; }
  python - <<PY
import json
d=json.loads(open("$O/$tag.json").read().strip().split("\n")[-1])
print("$tag", d.get("ms_per_step_hipevent_median"))
PY
}
run base X=1
run one_chunk0 COLVO_WGRAD_ONE_CHUNK_RULE=0
run teams1 COLVO_WGRAD_TEAMS=1
run teamwgs128 COLVO_WGRAD_TEAM_WGS=128
run teamwgs512 COLVO_WGRAD_TEAM_WGS=512
run base2 X=1
run walk16 COLVO_WGRAD_SHORT_WALK=16
run walk64 COLVO_WGRAD_SHORT_WALK=64
run atomic6 COLVO_WGRAD_ATOMIC_MB=6
run atomic1 COLVO_WGRAD_ATOMIC_MB=1.5
run wglo192 COLVO_WGRAD_WG_LO=192
run wglo384 COLVO_WGRAD_WG_LO=384
run base3 X=1
