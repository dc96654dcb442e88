#!/bin/bash
# round 5, call 15: two-level FINAL ARRIVAL (one-pass dense, set-up, stencil sweep) against the build before it, on one lease
set -e
mkdir -p gpurun_out/r05
for lib in base new base new; do
  if [ $lib = base ]; then export FASTA_HIP_LIB=fasta_python_amd/libfasta_hip_base.so; else unset FASTA_HIP_LIB; fi
  echo "== $lib"
  timeout -k 10 200 python bench.py --workload tv --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('tv   ', round(d['value'],1), 'it/s', r['avg_launch_ms'], 'ms', round(r['frac'],4))"
  timeout -k 10 200 python bench.py --rows 4096 --cols 4096 --steps 200 --warmup 10 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('4096 ', round(d['value'],1), 'it/s', r['avg_launch_ms'], 'ms')"
  timeout -k 10 200 python bench.py --rows 8192 --cols 8192 --steps 200 --warmup 10 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('8192 ', round(d['value'],1), 'it/s', r['avg_launch_ms'], 'ms')"
done
unset FASTA_HIP_LIB
timeout -k 10 600 python -m pytest tests/test_gpu_prox_tv.py tests/test_gpu_fused.py tests/test_gpu_setup.py tests/test_gpu_dense.py -m gpu -x -q 2>&1 | tail -2
