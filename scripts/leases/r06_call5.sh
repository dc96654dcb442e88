#!/bin/bash
# round 6, lease 5: fh_alloc_settle on evidence (6 interleaved cycles, wait on / off / control), then the new tests again
mkdir -p gpurun_out/r06
step() { local limit=$1 out=$2; shift 2; timeout -k 10 "$limit" "$@" > "$out" 2>&1; local rc=$?; echo "$* -> rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed: stopping the lease"; tail -20 "$out"; exit $rc; fi; }
step 900 gpurun_out/r06/alloc_settle.txt python scripts/probes/alloc_settle_ab.py 6
cat gpurun_out/r06/alloc_settle.txt
step 300 gpurun_out/r06/tests_new5.txt python -m pytest tests/test_gpu_faults.py tests/test_gpu_iterate.py -q -m gpu
tail -5 gpurun_out/r06/tests_new5.txt
