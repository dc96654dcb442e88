#!/bin/bash
# round 5, call 16: where does the one-pass kernel start to pay now that its fixed cost is lower?  fused on / auto across small sizes, one lease
mkdir -p gpurun_out/r05
for s in "512 1024" "1024 1024" "1024 2048" "2048 2048" "1024 4096" "2048 4096" "3000 3000" "4096 2048" "8192 1024" "256 8192" "512 16384"; do
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
