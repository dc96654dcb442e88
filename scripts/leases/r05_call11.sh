#!/bin/bash
# round 5, call 11: the device loop for n in (4096, 8192]: parity tests, rates, phase table
set -e
mkdir -p gpurun_out/r05
timeout -k 10 600 python -m pytest tests/test_gpu_run.py -m gpu -x -q > gpurun_out/r05/tests_run_wide.txt 2>&1 || { tail -30 gpurun_out/r05/tests_run_wide.txt; exit 1; }
tail -2 gpurun_out/r05/tests_run_wide.txt
timeout -k 10 300 python scripts/probes/run_cost.py 8192x8192 6000x6000 2048x8192 16384x8192 4096x4096 > gpurun_out/r05/run_cost_wide.txt 2>&1
cat gpurun_out/r05/run_cost_wide.txt
FASTA_HIP_LIB=fasta_python_amd/libfasta_hip_profrun.so timeout -k 10 200 python scripts/probes/run_cost.py 8192x8192 > gpurun_out/r05/run_prof_wide.txt 2>&1
grep "run profile" gpurun_out/r05/run_prof_wide.txt | tail -3
