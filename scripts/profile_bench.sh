#!/bin/bash
# rocprofv3 passes over the default bench: kernel trace + stats, then HBM counters (separate passes,
# MI355X_MICROARCH.md "HBM" / "rocprofv3 PMC slots").  Usage: bash scripts/profile_bench.sh <tag>
# `--skip-extra inproc,tv_512`: the two in-process row-block sub-results launch the headline kernel's instantiation on 8192- and
# 32768-row blocks; rocprofv3 --stats keys its rows by kernel name only, so they would be averaged into the headline's row
# (likewise tv_512: the 512^2 sweeps are the 8192^2 sweeps' instantiations).
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --skip-extra inproc,tv_512 > "$OUT/bench_kernel_trace.log" 2>&1
echo "kernel-trace rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --skip-extra inproc,tv_512 > "$OUT/bench_pmc_fetch.log" 2>&1
echo "pmc fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --skip-extra inproc,tv_512 > "$OUT/bench_pmc_write.log" 2>&1
echo "pmc write rc=$?"
find "$OUT" -name "*.csv" | head -20
du -sh "$OUT"
