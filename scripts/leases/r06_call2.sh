#!/bin/bash
# round 6, lease 2: launch-gap micro-benchmark (who decides between two launches), the fixed fh_iterate test, the stencil's grid at small images
mkdir -p gpurun_out/r06
step() { local limit=$1 out=$2; shift 2; timeout -k 10 "$limit" "$@" > "$out" 2>&1; local rc=$?; echo "$* -> rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed: stopping the lease"; tail -20 "$out"; exit $rc; fi; }
step 120 gpurun_out/r06/launchgap.txt scripts/probes/bench_mem/launchgap
cat gpurun_out/r06/launchgap.txt
step 900 gpurun_out/r06/tests_new.txt python -m pytest tests/test_gpu_faults.py tests/test_gpu_iterate.py tests/test_gpu_run.py -q -m gpu
tail -15 gpurun_out/r06/tests_new.txt
