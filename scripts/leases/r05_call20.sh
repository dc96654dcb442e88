#!/bin/bash
# round 5, call 20: the default bench as the driver runs it (with settle_after_free) + the profile passes behind roofline.traffic
set -e
mkdir -p gpurun_out/r05
t0=$(date +%s)
timeout -k 10 900 python bench.py > gpurun_out/r05/bench_default.json 2> gpurun_out/r05/bench_default.err
echo "bench done in $(( $(date +%s) - t0 )) s" | tee gpurun_out/r05/bench_wall.txt
rm -rf gpurun_out/prof_r05
timeout -k 10 900 bash scripts/profile_bench.sh r05 > gpurun_out/r05/profile.log 2>&1
echo "profile done"
