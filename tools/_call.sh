mkdir -p gpurun_out
run() { tag=$1; shift; env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --steps 80 > gpurun_out/r2_sweep_$tag.log 2>&1 || exit 1
python - <<PY
import json
d=json.loads(open("gpurun_out/r2_sweep_$tag.log").read().strip().split("\n")[-1])
print("$tag", round(d["ms_per_step"],4), round(d["ms_per_step_hipevent_median"],4))
PY
}
for r in 1 2; do
run l512_$r COLVO_LONE_MAX_WGS=512
run l1024_$r COLVO_LONE_MAX_WGS=1024
run l2048_$r COLVO_LONE_MAX_WGS=2048
run l256_$r COLVO_LONE_MAX_WGS=256
run l0_$r COLVO_LONE_MAX_WGS=0
done
