#!/bin/bash
# round 5, call 8b: parity tests over every kernel that sums hand-off partials with the 16-byte sc1 loads
set -e
mkdir -p gpurun_out/r05
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r05/tests_call8.txt 2>&1 || { tail -30 gpurun_out/r05/tests_call8.txt; exit 1; }
tail -3 gpurun_out/r05/tests_call8.txt
