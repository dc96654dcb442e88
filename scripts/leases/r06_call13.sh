#!/bin/bash
# round 6, lease 13 (final build): default bench line, rocprofv3 passes (kernel trace, FETCH_SIZE, WRITE_SIZE), sizes, the new full-length test
mkdir -p gpurun_out/r06
step() { local limit=$1 out=$2; shift 2; timeout -k 10 "$limit" "$@" > "$out" 2>&1; local rc=$?; echo "$* -> rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed: stopping the lease"; tail -20 "$out"; exit $rc; fi; }
step 400 gpurun_out/r06/bench_final.json python bench.py
tail -c 600 gpurun_out/r06/bench_final.json; echo
step 200 gpurun_out/r06/tests_13.txt python -m pytest tests/test_gpu_run.py -q -m gpu -k "full_length or chain_length"
tail -3 gpurun_out/r06/tests_13.txt
step 500 gpurun_out/r06/sizes_final.txt bash scripts/sizes.sh
cat gpurun_out/r06/sizes_final.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 700 bash scripts/profile_bench.sh r06 > gpurun_out/r06/profile_bench.log 2>&1; echo "profile rc=$?"
python scripts/summarize_profile.py r06 > gpurun_out/r06/summarize.log 2>&1; echo "summarize rc=$?"
cp profiles/r06_kernel_stats.csv profiles/r06_pmc_summary.json gpurun_out/r06/ 2>/dev/null
rm -rf gpurun_out/prof_r06/pmc_fetch gpurun_out/prof_r06/pmc_write 2>/dev/null; find gpurun_out/prof_r06 -name "*kernel_trace.csv" -delete
