#!/bin/bash
# round 6, lease 1: the new failure-path and fh_iterate tests, then who-drives-the-loop across sizes and the stencil at 512^2
# (a step that had to be KILLED ends the lease: no further GPU step after a hang)
mkdir -p gpurun_out/r06
step() { local limit=$1 out=$2; shift 2; timeout -k 10 "$limit" "$@" > "$out" 2>&1; local rc=$?; echo "$* -> rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed: stopping the lease"; tail -20 "$out"; exit $rc; fi; }
step 900 gpurun_out/r06/tests_new.txt python -m pytest tests/test_gpu_faults.py tests/test_gpu_iterate.py tests/test_gpu_run.py -q -m gpu
tail -15 gpurun_out/r06/tests_new.txt
step 400 gpurun_out/r06/driver_cost.txt python scripts/probes/driver_cost.py
step 300 gpurun_out/r06/tv_small.txt python scripts/probes/tv_small.py 512 1024 2048
step 300 gpurun_out/r06/sizes.txt bash scripts/sizes.sh
cat gpurun_out/r06/driver_cost.txt gpurun_out/r06/tv_small.txt gpurun_out/r06/sizes.txt
