#!/bin/bash
# round 6, lease 10: GPU-side gaps between consecutive one-pass launches, host-driven (library loop) against the chain
mkdir -p gpurun_out/r06
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for n in 8192 16384; do for d in library device; do
  rm -rf gpurun_out/gaps_${n}_$d
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gaps_${n}_$d -- python3 scripts/probes/chain_gaps.py $n $d > gpurun_out/r06/gaps_${n}_$d.log 2>&1; rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo killed; exit $rc; fi
  echo "n=$n $d: $(grep 'device steps' gpurun_out/r06/gaps_${n}_$d.log)"; python scripts/probes/chain_gaps_report.py gpurun_out/gaps_${n}_$d
  rm -rf gpurun_out/gaps_${n}_$d
done; done 2>&1 | tee gpurun_out/r06/chain_gaps.txt
