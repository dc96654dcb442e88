#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r05
echo "== run tests"; timeout -k 10 900 python -m pytest tests/test_gpu_run.py -m gpu -x -q > gpurun_out/r05/tests_run.txt 2>&1; echo "rc=$?"; tail -25 gpurun_out/r05/tests_run.txt
echo "== run cost"; timeout -k 10 600 python scripts/probes/run_cost.py > gpurun_out/r05/run_cost.txt 2>&1; echo "rc=$?"; cat gpurun_out/r05/run_cost.txt
echo "== level"; timeout -k 10 300 python scripts/probes/level_search_cost.py 4096 8192 > gpurun_out/r05/level_search_cost_d.txt 2>&1; echo "rc=$?"; cat gpurun_out/r05/level_search_cost_d.txt
