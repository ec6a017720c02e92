mkdir -p gpurun_out
run() { tag=$1; shift; env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --steps 80 > gpurun_out/r2_abl_$tag.log 2>&1 || exit 1
python - <<PY
import json
d=json.loads(open("gpurun_out/r2_abl_$tag.log").read().strip().split("\n")[-1])
print("$tag", round(d["ms_per_step"],4), round(d["ms_per_step_hipevent_median"],4))
PY
}
for r in 1 2; do
run full_$r A=1
run off_$r COLVO_NO_LDS_AWARE_TILES=1
run nopad_$r COLVO_LDS_TILE_MAX_PAD=0
done
