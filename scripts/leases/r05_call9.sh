#!/bin/bash
# round 5, call 9: the default bench (what the driver runs), the profile passes behind roofline.traffic, sizes, the experimental job
set -e
mkdir -p gpurun_out/r05
timeout -k 10 900 python bench.py > gpurun_out/r05/bench_default.json 2> gpurun_out/r05/bench_default.err
echo "bench done"
timeout -k 10 900 bash scripts/profile_bench.sh r05
python scripts/summarize_profile.py r05 > gpurun_out/r05/summarize.txt 2>&1
echo "profile done"
timeout -k 10 400 bash scripts/sizes.sh > gpurun_out/r05/sizes_final.txt 2>&1
echo "sizes done"
FASTA_HIP_LIB=fasta_python_amd/libfasta_hip_experimental.so timeout -k 10 300 python -m pytest tests/test_gpu_experimental.py -m gpu -x -q > gpurun_out/r05/tests_experimental.txt 2>&1
tail -2 gpurun_out/r05/tests_experimental.txt
