#!/bin/bash
# round 5, GPU call 1: tiled-storage probe for the TV sweep, the tests touched so far, level-search cost, default bench
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r05
echo "== tvshape 12"; timeout -k 10 300 scripts/probes/bench_mem/tvshape 12 > gpurun_out/r05/tvshape12.txt 2>&1; echo "rc=$?"
echo "== level search cost"; timeout -k 10 600 python scripts/probes/level_search_cost.py > gpurun_out/r05/level_search_cost.txt 2>&1; echo "rc=$?"
echo "== tests"; timeout -k 10 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_multiprocess.py tests/test_gpu_prox_tv.py -m gpu -x -q -s > gpurun_out/r05/tests_a.txt 2>&1; echo "rc=$?"; tail -5 gpurun_out/r05/tests_a.txt
echo "== experimental job"; FASTA_HIP_LIB=$PWD/fasta_python_amd/libfasta_hip_experimental.so timeout -k 10 600 python -m pytest tests/test_gpu_experimental.py -m gpu -x -q > gpurun_out/r05/tests_exp.txt 2>&1; echo "rc=$?"; tail -3 gpurun_out/r05/tests_exp.txt
echo "== bench"; timeout -k 10 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r05/bench_a.json 2> gpurun_out/r05/bench_a.err; echo "rc=$?"; tail -c 600 gpurun_out/r05/bench_a.err
