#!/bin/bash
# round 6, lease 7: device loop against the library's host-side loop at every width fh_run can take (final build), kept-block A/B
mkdir -p gpurun_out/r06
step() { local limit=$1 out=$2; shift 2; timeout -k 10 "$limit" "$@" > "$out" 2>&1; local rc=$?; echo "$* -> rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed: stopping the lease"; tail -20 "$out"; exit $rc; fi; }
step 400 gpurun_out/r06/driver_cost_final.txt python scripts/probes/driver_cost.py
cat gpurun_out/r06/driver_cost_final.txt
step 300 gpurun_out/r06/driver_cost_shapes.txt python scripts/probes/driver_cost.py 512 6144 2048 6144 16384 6144 512 7168 2048 7168 16384 7168 512 4096 16384 4096 65536 4096
cat gpurun_out/r06/driver_cost_shapes.txt
step 900 gpurun_out/r06/alloc_settle2.txt python scripts/probes/alloc_settle_ab.py 2
tail -12 gpurun_out/r06/alloc_settle2.txt
