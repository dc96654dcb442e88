#!/bin/bash
set -e
mkdir -p gpurun_out/r05
timeout -k 10 600 python -m pytest tests/test_gpu_run.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/r05/tests_run_bar2.txt 2>&1 || { tail -30 gpurun_out/r05/tests_run_bar2.txt; exit 1; }
tail -2 gpurun_out/r05/tests_run_bar2.txt
timeout -k 10 300 python scripts/probes/run_cost.py 512x1024 2048x2048 4096x4096 6000x6000 7000x7000 > gpurun_out/r05/run_cost_bar2.txt 2>&1
cut -c1-400 gpurun_out/r05/run_cost_bar2.txt
FASTA_HIP_LIB=fasta_python_amd/libfasta_hip_profrun.so timeout -k 10 200 python scripts/probes/run_cost.py 4096x4096 512x1024 > gpurun_out/r05/run_prof_bar2.txt 2>&1
grep "run profile" gpurun_out/r05/run_prof_bar2.txt | tail -6
