for t in "9=2" "9=34" ""; do
  echo "== bench --tune '$t'"
  timeout -k 10 200 python bench.py --no-extra --no-cpu-baseline --repeats 3 ${t:+--tune $t} > gpurun_out/r06/ab_bench.json 2>gpurun_out/r06/ab_bench.err || { tail -5 gpurun_out/r06/ab_bench.err; exit 1; }
  python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06/ab_bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print(round(d['value'],1), 'launch', round(r['avg_launch_ms'],3), 'wall', d['spread']['runs_ms_per_step'])
PY
done
