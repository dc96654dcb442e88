#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r05
echo "== tests"; timeout -k 10 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_dense.py tests/test_gpu_setup.py tests/test_gpu_run.py tests/test_gpu_f32.py tests/test_gpu_inproc_sharding.py -m gpu -x -q > gpurun_out/r05/tests_b.txt 2>&1; echo "rc=$?"; tail -4 gpurun_out/r05/tests_b.txt
echo "== bench"; timeout -k 10 900 python bench.py --no-cpu-baseline --skip-extra inproc > gpurun_out/r05/bench_b.json 2> gpurun_out/r05/bench_b.err; echo "rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05/bench_b.json'))
print('headline', d['value'], d['roofline']['avg_launch_ms'])
for k,v in d['extra'].items():
    if 'avg_launch_ms' in v: print(k, round(v['value'],1), v['avg_launch_ms'])
print(json.dumps(d['extra']['natural_run']['lasso']))
PY
echo "== sizes"; timeout -k 10 600 bash scripts/sizes.sh > gpurun_out/r05/sizes_b.txt 2>&1; cat gpurun_out/r05/sizes_b.txt
