#!/bin/bash
mkdir -p gpurun_out/r05
for s in "64 128" "96 160" "200 1000" "256 512" "500 2000" "512 512" "300 4096" "1000 2000" "128 4096" "2000 500" "4096 256"; do
  set -- $s
  for f in auto on; do
    timeout -k 10 100 python bench.py --rows $1 --cols $2 --steps 300 --warmup 10 --no-cpu-baseline --no-extra --fused $f 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
pk=d['roofline']['per_kernel']
parts=' | '.join('%s %7.4f ms' % (k.split('(')[1][:-1], v['avg_ms']) for k,v in pk.items() if v['launches'])
print('%6d x %6d fused=%-4s %8.0f it/s  %8.4f ms/step | %s' % (d['config']['m'], d['config']['n'], '$f', d['value'], d['ms_per_step'], parts))"
  done
done
