#!/bin/bash
# round 5, call 12: full GPU suite, the default bench with its wall clock, the profile passes
set -e
mkdir -p gpurun_out/r05
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r05/tests_call12.txt 2>&1 || { tail -30 gpurun_out/r05/tests_call12.txt; exit 1; }
tail -2 gpurun_out/r05/tests_call12.txt
t0=$(date +%s)
timeout -k 10 900 python bench.py > gpurun_out/r05/bench_default.json 2> gpurun_out/r05/bench_default.err
echo "bench done in $(( $(date +%s) - t0 )) s" | tee gpurun_out/r05/bench_wall.txt
rm -rf gpurun_out/prof_r05
timeout -k 10 900 bash scripts/profile_bench.sh r05
echo "profile done"
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05/smoke.txt 2>&1; tail -2 gpurun_out/r05/smoke.txt
