#!/bin/bash
# round 6, lease 11: tests after making the chain opt-in; the whole suite
mkdir -p gpurun_out/r06
step() { local limit=$1 out=$2; shift 2; timeout -k 10 "$limit" "$@" > "$out" 2>&1; local rc=$?; echo "$* -> rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed: stopping the lease"; tail -20 "$out"; exit $rc; fi; }
step 600 gpurun_out/r06/tests_chain2.txt python -m pytest tests/test_gpu_run.py tests/test_gpu_faults.py tests/test_gpu_fuzz.py -q -m gpu
tail -12 gpurun_out/r06/tests_chain2.txt
step 900 gpurun_out/r06/tests_full11.txt python -m pytest tests -q -m gpu --deselect tests/test_gpu_run.py --deselect tests/test_gpu_faults.py --deselect tests/test_gpu_fuzz.py
tail -8 gpurun_out/r06/tests_full11.txt
