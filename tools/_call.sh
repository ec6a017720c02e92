mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_program_gpu.py tests/test_nets_gpu.py tests/test_ddp_gpu.py -x -q > gpurun_out/r2_tests_37.log 2>&1 || { tail -30 gpurun_out/r2_tests_37.log; exit 1; }
tail -2 gpurun_out/r2_tests_37.log
run() { tag=$1; shift; env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --steps 80 > gpurun_out/r2_abl_$tag.log 2>&1 || exit 1
python - <<PY
import json
d=json.loads(open("gpurun_out/r2_abl_$tag.log").read().strip().split("\n")[-1])
print("$tag", round(d["ms_per_step"],4), round(d["ms_per_step_hipevent_median"],4))
PY
}
for r in 1 2; do
run s2_$r COLVO_SIDE_STREAMS=2
run s3_$r COLVO_SIDE_STREAMS=3
run s4_$r COLVO_SIDE_STREAMS=4
done
