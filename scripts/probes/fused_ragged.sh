#!/bin/bash
# one-pass vs two-launch at sizes that are NOT multiples of 4096 (GPU box)
for s in "20000 20000" "24000 24000" "30000 30000" "40000 40000" "50000 50000" "60000 60000" "65000 65000"; do
  set -- $s
  for f in on off; do
    python bench.py --rows $1 --cols $2 --steps 30 --warmup 4 --no-cpu-baseline --fused $f 2>&1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%6d x %6d fused=%-3s %9.0f it/s  %8.4f ms/step   matrix read at %5.0f GB/s per pass-equivalent' % (d['config']['m'], d['config']['n'], '$f', d['value'], d['ms_per_step'], d['config']['m']*d['config']['n']*8/d['ms_per_step']/1e6))"
  done
done
