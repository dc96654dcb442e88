#!/bin/bash
# round 6, lease 18 (cyclic dealing from 128 rows per team on): the whole -m gpu suite, the experimental job, the sizes table
mkdir -p gpurun_out/r06
step() { local limit=$1 out=$2; shift 2; timeout -k 10 "$limit" "$@" > "$out" 2>&1; local rc=$?; echo "$* -> rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed: stopping the lease"; tail -20 "$out"; exit $rc; fi; }
step 1000 gpurun_out/r06/tests_full18.txt python -m pytest tests -q -m gpu
tail -4 gpurun_out/r06/tests_full18.txt
FASTA_HIP_LIB=fasta_python_amd/libfasta_hip_experimental.so step 200 gpurun_out/r06/tests_exp18.txt python -m pytest tests/test_gpu_experimental.py -q -m gpu
tail -2 gpurun_out/r06/tests_exp18.txt
step 500 gpurun_out/r06/sizes18.txt bash scripts/sizes.sh
tail -30 gpurun_out/r06/sizes18.txt
