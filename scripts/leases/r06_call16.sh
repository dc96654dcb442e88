#!/bin/bash
# round 6, lease 16: rows dealt cyclically is the one-pass kernel's (and the set-up kernel's) default -- the whole -m gpu suite, the experimental job,
# the set-up cost probe, the default bench line
mkdir -p gpurun_out/r06
step() { local limit=$1 out=$2; shift 2; timeout -k 10 "$limit" "$@" > "$out" 2>&1; local rc=$?; echo "$* -> rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed: stopping the lease"; tail -20 "$out"; exit $rc; fi; }
step 1000 gpurun_out/r06/tests_full16.txt python -m pytest tests -q -m gpu -x
tail -6 gpurun_out/r06/tests_full16.txt
FASTA_HIP_LIB=fasta_python_amd/libfasta_hip_experimental.so step 200 gpurun_out/r06/tests_exp16.txt python -m pytest tests/test_gpu_experimental.py -q -m gpu
tail -2 gpurun_out/r06/tests_exp16.txt
step 200 gpurun_out/r06/setup_cost16.txt python scripts/probes/setup_cost.py
tail -12 gpurun_out/r06/setup_cost16.txt
step 400 gpurun_out/r06/bench16.json python bench.py
tail -c 300 gpurun_out/r06/bench16.json; echo
