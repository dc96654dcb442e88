#!/bin/bash
# rocprofv3 kernel trace + PMC passes over the TV bench (BASELINE config 4).  Usage: bash scripts/probes/profile_tv.sh <tag>
set -u
TAG=${1:-r01tv}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 bench.py --workload tv --steps 30 --warmup 2 > "$OUT/bench_kernel_trace.log" 2>&1
echo "kernel-trace rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py --workload tv --steps 6 --warmup 1 > "$OUT/bench_pmc_fetch.log" 2>&1
echo "pmc fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 bench.py --workload tv --steps 6 --warmup 1 > "$OUT/bench_pmc_write.log" 2>&1
echo "pmc write rc=$?"
du -sh "$OUT"
