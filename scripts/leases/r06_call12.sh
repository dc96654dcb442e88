#!/bin/bash
# round 6, lease 12: the whole -m gpu suite on the final build; default bench line; rocprofv3 passes; sizes
mkdir -p gpurun_out/r06
step() { local limit=$1 out=$2; shift 2; timeout -k 10 "$limit" "$@" > "$out" 2>&1; local rc=$?; echo "$* -> rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed: stopping the lease"; tail -20 "$out"; exit $rc; fi; }
step 1000 gpurun_out/r06/tests_full12.txt python -m pytest tests -q -m gpu
tail -6 gpurun_out/r06/tests_full12.txt
step 100 gpurun_out/r06/smoke.txt python __graft_entry__.py smoke
tail -2 gpurun_out/r06/smoke.txt
