mkdir -p gpurun_out
run() { tag=$1; shift; env "$@" timeout -k 10 120 python bench.py --no-cpu-baseline --steps 80 > gpurun_out/r2_abl_$tag.log 2>&1 || { tail -5 gpurun_out/r2_abl_$tag.log; return 1; }
python - <<PY
import json
d=json.loads(open("gpurun_out/r2_abl_$tag.log").read().strip().split("\n")[-1])
print("$tag", round(d["ms_per_step"],4), round(d["ms_per_step_hipevent_median"],4), d["final_loss"])
PY
}
for r in 1 2; do
run g1_$r COLVO_WGRAD_GROUP=1 && run g2_$r COLVO_WGRAD_GROUP=2 && run g3_$r COLVO_WGRAD_GROUP=3 && run g4_$r COLVO_WGRAD_GROUP=4
done
