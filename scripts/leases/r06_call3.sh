#!/bin/bash
# round 6, lease 3: "the scalars are there" by sequence number (FH_TUNE_SEQ_POLL) A/B, rows per workgroup of the stencil sweep on small images, tests
mkdir -p gpurun_out/r06
step() { local limit=$1 out=$2; shift 2; timeout -k 10 "$limit" "$@" > "$out" 2>&1; local rc=$?; echo "$* -> rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed: stopping the lease"; tail -20 "$out"; exit $rc; fi; }
step 600 gpurun_out/r06/tests_new3.txt python -m pytest tests/test_gpu_faults.py tests/test_gpu_iterate.py tests/test_gpu_run.py tests/test_gpu_fused.py tests/test_gpu_prox_tv.py -q -m gpu -x
tail -8 gpurun_out/r06/tests_new3.txt
step 300 gpurun_out/r06/driver_cost_seq.txt python scripts/probes/driver_cost.py 2048 2048 4096 4096 8192 8192 16384 16384 32768 32768
FH_SEQ_POLL=0 step 300 gpurun_out/r06/driver_cost_noseq.txt python scripts/probes/driver_cost.py 2048 2048 4096 4096 8192 8192 16384 16384 32768 32768
step 400 gpurun_out/r06/tv_rows_small.txt python scripts/probes/tv_rows_small.py 512 1024 2048
step 300 gpurun_out/r06/sizes3.txt bash scripts/sizes.sh
cat gpurun_out/r06/driver_cost_seq.txt gpurun_out/r06/driver_cost_noseq.txt gpurun_out/r06/tv_rows_small.txt gpurun_out/r06/sizes3.txt
