#!/bin/bash
# round 5, call 10: full GPU suite, the default bench (what the driver runs), the profile passes behind roofline.traffic, the experimental job
set -e
mkdir -p gpurun_out/r05
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r05/tests_call10.txt 2>&1 || { tail -30 gpurun_out/r05/tests_call10.txt; exit 1; }
tail -2 gpurun_out/r05/tests_call10.txt
timeout -k 10 900 python bench.py > gpurun_out/r05/bench_default.json 2> gpurun_out/r05/bench_default.err
echo "bench done"
timeout -k 10 900 bash scripts/profile_bench.sh r05
echo "profile done"
FASTA_HIP_LIB=fasta_python_amd/libfasta_hip_experimental.so timeout -k 10 300 python -m pytest tests/test_gpu_experimental.py -m gpu -x -q > gpurun_out/r05/tests_experimental.txt 2>&1
tail -2 gpurun_out/r05/tests_experimental.txt
