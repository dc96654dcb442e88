#!/bin/bash
# round 6, lease 1: the new failure-path and fh_iterate tests, then who-drives-the-loop across sizes and the stencil at 512^2
set -o pipefail
mkdir -p gpurun_out/r06
timeout -k 10 900 python -m pytest tests/test_gpu_faults.py tests/test_gpu_iterate.py tests/test_gpu_run.py -x -q -m gpu > gpurun_out/r06/tests_new.txt 2>&1
echo "tests rc=$?" | tee -a gpurun_out/r06/tests_new.txt
tail -5 gpurun_out/r06/tests_new.txt
timeout -k 10 400 python scripts/probes/driver_cost.py > gpurun_out/r06/driver_cost.txt 2>&1; echo "driver_cost rc=$?"
timeout -k 10 300 python scripts/probes/tv_small.py 512 1024 2048 > gpurun_out/r06/tv_small.txt 2>&1; echo "tv_small rc=$?"
timeout -k 10 300 bash scripts/sizes.sh > gpurun_out/r06/sizes.txt 2>&1; echo "sizes rc=$?"
cat gpurun_out/r06/driver_cost.txt gpurun_out/r06/tv_small.txt gpurun_out/r06/sizes.txt
