mkdir -p gpurun_out
for n in old new old new; do
  if [ $n = old ]; then export COLVO_LIB_PATH=$PWD/coivo_amd/lib/libcolvo_abl.so; else unset COLVO_LIB_PATH; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 60 > gpurun_out/r2_bench_wgl_$n.log 2>&1 || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/r2_bench_wgl_$n.log").read().strip().split("\n")[-1])
print("$n", d["ms_per_step"], d["ms_per_step_hipevent_median"])
PY
done
