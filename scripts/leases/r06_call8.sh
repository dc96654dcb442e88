#!/bin/bash
# round 6, lease 8: where the device loop (fh_run) beats the library's host-side loop -- a grid of (rows, columns)
mkdir -p gpurun_out/r06
step() { local limit=$1 out=$2; shift 2; timeout -k 10 "$limit" "$@" > "$out" 2>&1; local rc=$?; echo "$* -> rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed: stopping the lease"; tail -20 "$out"; exit $rc; fi; }
args=""
for n in 1024 2048 3072 4096 5120 6144; do for m in 256 1024 4096 8192 16384 32768; do args="$args $m $n"; done; done
step 900 gpurun_out/r06/driver_cost_grid.txt python scripts/probes/driver_cost.py $args
cat gpurun_out/r06/driver_cost_grid.txt
