#!/bin/bash
# round 5, call 14: two-level grid barrier in the one-pass / set-up / persistent-loop kernels: full GPU suite, sizes, the f32 and setup rates
set -e
mkdir -p gpurun_out/r05
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r05/tests_call14.txt 2>&1 || { tail -30 gpurun_out/r05/tests_call14.txt; exit 1; }
tail -2 gpurun_out/r05/tests_call14.txt
timeout -k 10 400 bash scripts/sizes.sh > gpurun_out/r05/sizes_bar2.txt 2>&1
cat gpurun_out/r05/sizes_bar2.txt
timeout -k 10 300 python bench.py --storage f32 --no-cpu-baseline --no-extra 2>/dev/null | cut -c1-330
timeout -k 10 300 python scripts/probes/setup_cost.py 65536 > gpurun_out/r05/setup_cost_bar2.txt 2>&1; cat gpurun_out/r05/setup_cost_bar2.txt
FASTA_HIP_LIB=fasta_python_amd/libfasta_hip_experimental.so timeout -k 10 300 python -m pytest tests/test_gpu_experimental.py -m gpu -x -q 2>&1 | tail -1
