#!/bin/bash
# round 5, call 19: full GPU suite, the default bench with its wall clock, the profile passes, sizes, the experimental job
set -e
mkdir -p gpurun_out/r05
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r05/tests_call19.txt 2>&1 || { tail -30 gpurun_out/r05/tests_call19.txt; exit 1; }
tail -2 gpurun_out/r05/tests_call19.txt
t0=$(date +%s)
timeout -k 10 900 python bench.py > gpurun_out/r05/bench_default.json 2> gpurun_out/r05/bench_default.err
echo "bench done in $(( $(date +%s) - t0 )) s" | tee gpurun_out/r05/bench_wall.txt
rm -rf gpurun_out/prof_r05
timeout -k 10 900 bash scripts/profile_bench.sh r05 > gpurun_out/r05/profile.log 2>&1
echo "profile done"
timeout -k 10 400 bash scripts/sizes.sh > gpurun_out/r05/sizes_final.txt 2>&1; cat gpurun_out/r05/sizes_final.txt
FASTA_HIP_LIB=fasta_python_amd/libfasta_hip_experimental.so timeout -k 10 300 python -m pytest tests/test_gpu_experimental.py -m gpu -x -q 2>&1 | tail -1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
