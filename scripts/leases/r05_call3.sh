#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r05
echo "== setup + level tests"; timeout -k 10 900 python -m pytest tests/test_gpu_setup.py tests/test_gpu_prox_tv.py -m gpu -x -q > gpurun_out/r05/tests_setup.txt 2>&1; echo "rc=$?"; tail -15 gpurun_out/r05/tests_setup.txt
echo "== setup cost"; timeout -k 10 600 python scripts/probes/setup_cost.py > gpurun_out/r05/setup_cost.txt 2>&1; echo "rc=$?"; cat gpurun_out/r05/setup_cost.txt
echo "== level search cost"; timeout -k 10 600 python scripts/probes/level_search_cost.py > gpurun_out/r05/level_search_cost_c.txt 2>&1; echo "rc=$?"; cat gpurun_out/r05/level_search_cost_c.txt
