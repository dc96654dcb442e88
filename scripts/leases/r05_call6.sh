#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r05
echo "== run tests"; timeout -k 10 600 python -m pytest tests/test_gpu_run.py tests/test_gpu_setup.py -m gpu -x -q > gpurun_out/r05/tests_run2.txt 2>&1; echo "rc=$?"; tail -4 gpurun_out/r05/tests_run2.txt
echo "== default bench"; timeout -k 10 900 python bench.py > gpurun_out/r05/bench_default.json 2> gpurun_out/r05/bench_default.err; echo "rc=$?"; head -c 300 gpurun_out/r05/bench_default.err
echo "== profile"; timeout -k 10 1500 bash scripts/profile_bench.sh r05 > gpurun_out/r05/profile.log 2>&1; echo "rc=$?"; tail -5 gpurun_out/r05/profile.log
python scripts/summarize_profile.py r05 > gpurun_out/r05/summarize.log 2>&1; echo "summarize rc=$?"
cp profiles/r05_kernel_stats.csv profiles/r05_pmc_summary.json gpurun_out/r05/ 2>/dev/null
