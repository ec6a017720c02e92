mkdir -p gpurun_out
for n in 1 0 1 0; do
  if [ $n = 1 ]; then export COLVO_NO_CONV_UP2=1; else unset COLVO_NO_CONV_UP2; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 60 > gpurun_out/r2_bench_up2_$n.log 2>&1 || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/r2_bench_up2_$n.log").read().strip().split("\n")[-1])
print("no_up2=$n:", d["ms_per_step"], d["ms_per_step_hipevent_median"])
PY
done
