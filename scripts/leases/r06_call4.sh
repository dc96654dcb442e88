#!/bin/bash
# round 6, lease 4: sequence number with system-scope stores + release; the WHOLE -m gpu suite; sizes; who drives the loop
mkdir -p gpurun_out/r06
step() { local limit=$1 out=$2; shift 2; timeout -k 10 "$limit" "$@" > "$out" 2>&1; local rc=$?; echo "$* -> rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed: stopping the lease"; tail -20 "$out"; exit $rc; fi; }
step 300 gpurun_out/r06/tests_new4.txt python -m pytest tests/test_gpu_faults.py tests/test_gpu_iterate.py -q -m gpu
tail -8 gpurun_out/r06/tests_new4.txt
step 300 gpurun_out/r06/driver_cost_seq4.txt python scripts/probes/driver_cost.py 2048 2048 4096 4096 8192 8192 16384 16384
step 200 gpurun_out/r06/tv_small4.txt python scripts/probes/tv_small.py 512 1024
step 300 gpurun_out/r06/sizes4.txt bash scripts/sizes.sh
cat gpurun_out/r06/driver_cost_seq4.txt gpurun_out/r06/tv_small4.txt gpurun_out/r06/sizes4.txt
step 900 gpurun_out/r06/tests_full4.txt python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_faults.py --deselect tests/test_gpu_iterate.py
tail -8 gpurun_out/r06/tests_full4.txt
