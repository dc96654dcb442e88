#!/bin/bash
# round 6, lease 9: the chained form of fh_run (k_fused_chain) -- tests, then who drives the loop at 8192^2 .. 16384^2
mkdir -p gpurun_out/r06
step() { local limit=$1 out=$2; shift 2; timeout -k 10 "$limit" "$@" > "$out" 2>&1; local rc=$?; echo "$* -> rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed: stopping the lease"; tail -20 "$out"; exit $rc; fi; }
step 600 gpurun_out/r06/tests_chain.txt python -m pytest tests/test_gpu_run.py tests/test_gpu_faults.py -q -m gpu -x
tail -12 gpurun_out/r06/tests_chain.txt
step 400 gpurun_out/r06/driver_cost_chain.txt python scripts/probes/driver_cost.py 6656 6656 7168 7168 8192 8192 10000 10000 12288 12288 16384 16384 16384 4096 65536 4096 512 8192 2048 8192
cat gpurun_out/r06/driver_cost_chain.txt
