mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r2_tests_full.log 2>&1 || { tail -40 gpurun_out/r2_tests_full.log; exit 1; }
tail -3 gpurun_out/r2_tests_full.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
timeout -k 10 400 python bench.py > gpurun_out/r2_bench_final.log 2>&1 || exit 1
tail -1 gpurun_out/r2_bench_final.log | cut -c1-330
