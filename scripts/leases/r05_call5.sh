#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r05
echo "== full gpu suite"; timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r05/tests_full.txt 2>&1; echo "rc=$?"; tail -6 gpurun_out/r05/tests_full.txt
echo "== sizes"; timeout -k 10 600 bash scripts/sizes.sh > gpurun_out/r05/sizes.txt 2>&1; echo "rc=$?"; cat gpurun_out/r05/sizes.txt
echo "== run cost"; timeout -k 10 600 python scripts/probes/run_cost.py > gpurun_out/r05/run_cost.txt 2>&1; echo "rc=$?"; cat gpurun_out/r05/run_cost.txt
