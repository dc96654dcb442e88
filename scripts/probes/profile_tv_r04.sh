#!/bin/bash
# Round 4: the SQ / TCP / TCC / LDS counter passes of scripts/probes/profile_tv_sq.sh for three forms of the one-pass TV sweep on ONE box:
#   a  round-3 default (register-staged 2-row trips, one 228-row chunk per workgroup)
#   b  FH_TUNE_TV_RING = 2: trips prefetched by LDS-DMA into a per-wave ring (loads in flight cost no registers)
#   c  FH_TUNE_TV_SLOTS = 5 + FH_TUNE_TV_ROWS = 32: persistent workgroups walking short chunks (one compact moving window)
set -u
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
EXTRA="--tune 14=1,15=0" bash scripts/probes/profile_tv_sq.sh r04tv_a > /dev/null 2>&1; echo "a rc=$?"
EXTRA="--tune 14=2,15=0" bash scripts/probes/profile_tv_sq.sh r04tv_b > /dev/null 2>&1; echo "b rc=$?"
EXTRA="--tune 14=1,15=5,7=32" bash scripts/probes/profile_tv_sq.sh r04tv_c > /dev/null 2>&1; echo "c rc=$?"
for t in a b c; do echo "=== form $t"; cat gpurun_out/prof_r04tv_$t/summary.txt; grep -h '"metric"' gpurun_out/prof_r04tv_$t/bench_pmc_1.log | head -1 | cut -c1-400; done
