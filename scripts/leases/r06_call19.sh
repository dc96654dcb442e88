#!/bin/bash
# round 6, lease 19 (HEAD: cyclic dealing from 128 rows per team, plain loads for cache-sized matrices in K-fwd / K-adj): the whole -m gpu suite, smoke, the experimental job, the default bench line, the rocprofv3 passes, the sizes table
mkdir -p gpurun_out/r06
step() { local limit=$1 out=$2; shift 2; timeout -k 10 "$limit" "$@" > "$out" 2>&1; local rc=$?; echo "$* -> rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed: stopping the lease"; tail -20 "$out"; exit $rc; fi; }
step 1000 gpurun_out/r06/tests_full19.txt python -m pytest tests -q -m gpu
tail -4 gpurun_out/r06/tests_full19.txt
step 100 gpurun_out/r06/smoke19.txt python __graft_entry__.py smoke
tail -1 gpurun_out/r06/smoke19.txt
FASTA_HIP_LIB=fasta_python_amd/libfasta_hip_experimental.so step 200 gpurun_out/r06/tests_exp19.txt python -m pytest tests/test_gpu_experimental.py -q -m gpu
tail -2 gpurun_out/r06/tests_exp19.txt
step 400 gpurun_out/r06/bench_final4.json python bench.py
tail -c 300 gpurun_out/r06/bench_final4.json; echo
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 700 bash scripts/profile_bench.sh r06 > gpurun_out/r06/profile_bench.log 2>&1; echo "profile rc=$?"
python scripts/summarize_profile.py r06 > gpurun_out/r06/summarize.log 2>&1; echo "summarize rc=$?"
cp profiles/r06_kernel_stats.csv profiles/r06_pmc_summary.json gpurun_out/r06/ 2>/dev/null
cp gpurun_out/prof_r06/bench_kernel_trace.log gpurun_out/r06/bench_under_kernel_trace.log 2>/dev/null
rm -rf gpurun_out/prof_r06/pmc_fetch gpurun_out/prof_r06/pmc_write 2>/dev/null; find gpurun_out/prof_r06 -name "*kernel_trace.csv" -delete
step 500 gpurun_out/r06/sizes19.txt bash scripts/sizes.sh
tail -30 gpurun_out/r06/sizes19.txt
