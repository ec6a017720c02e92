mkdir -p gpurun_out
run() { tag=$1; shift; env "$@" timeout -k 10 120 python bench.py --no-cpu-baseline --steps 80 > gpurun_out/r2_abl_$tag.log 2>&1 || { tail -5 gpurun_out/r2_abl_$tag.log; return 1; }
python - <<PY
import json
d=json.loads(open("gpurun_out/r2_abl_$tag.log").read().strip().split("\n")[-1])
print("$tag", round(d["ms_per_step"],4), round(d["ms_per_step_hipevent_median"],4))
PY
}
for r in 1 2 3; do
run rule_$r A=1
run norule_$r COLVO_WGRAD_NO_ONE_CHUNK_RULE=1
done
