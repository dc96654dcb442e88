#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r05
echo "== setup tests"; timeout -k 10 600 python -m pytest tests/test_gpu_setup.py -m gpu -x -q > gpurun_out/r05/tests_setup.txt 2>&1; echo "rc=$?"; tail -15 gpurun_out/r05/tests_setup.txt
echo "== level tests"; timeout -k 10 600 python -m pytest tests/test_gpu_prox_tv.py tests/test_gpu_dense.py -m gpu -x -q -k "level or linf or l1ball or golden" > gpurun_out/r05/tests_level.txt 2>&1; echo "rc=$?"; tail -5 gpurun_out/r05/tests_level.txt
echo "== setup cost"; timeout -k 10 600 python scripts/probes/setup_cost.py > gpurun_out/r05/setup_cost.txt 2>&1; echo "rc=$?"; cat gpurun_out/r05/setup_cost.txt
echo "== level search cost"; timeout -k 10 600 python scripts/probes/level_search_cost.py > gpurun_out/r05/level_search_cost_b.txt 2>&1; echo "rc=$?"; cat gpurun_out/r05/level_search_cost_b.txt
