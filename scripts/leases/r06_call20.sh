#!/bin/bash
# round 6, lease 20 (HEAD): the new dealing test, smoke, the default bench line (with before_start), the rocprofv3 passes
mkdir -p gpurun_out/r06
step() { local limit=$1 out=$2; shift 2; timeout -k 10 "$limit" "$@" > "$out" 2>&1; local rc=$?; echo "$* -> rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed: stopping the lease"; tail -20 "$out"; exit $rc; fi; }
step 600 gpurun_out/r06/tests_20.txt python -m pytest tests/test_gpu_fused.py tests/test_gpu_dense.py tests/test_gpu_fuzz.py -q -m gpu
tail -3 gpurun_out/r06/tests_20.txt
step 100 gpurun_out/r06/smoke20.txt python __graft_entry__.py smoke
tail -1 gpurun_out/r06/smoke20.txt
step 400 gpurun_out/r06/bench_final5.json python bench.py
tail -c 300 gpurun_out/r06/bench_final5.json; echo
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 700 bash scripts/profile_bench.sh r06 > gpurun_out/r06/profile_bench.log 2>&1; echo "profile rc=$?"
python scripts/summarize_profile.py r06 > gpurun_out/r06/summarize.log 2>&1; echo "summarize rc=$?"
cp profiles/r06_kernel_stats.csv profiles/r06_pmc_summary.json gpurun_out/r06/ 2>/dev/null
cp gpurun_out/prof_r06/bench_kernel_trace.log gpurun_out/r06/bench_under_kernel_trace.log 2>/dev/null
rm -rf gpurun_out/prof_r06/pmc_fetch gpurun_out/prof_r06/pmc_write 2>/dev/null; find gpurun_out/prof_r06 -name "*kernel_trace.csv" -delete
