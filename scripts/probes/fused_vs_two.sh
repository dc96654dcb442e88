#!/bin/bash
# it/s with the one-pass kernel on / off across sizes (GPU box): decides where `fused="auto"` should switch it on.
for s in "4096 4096" "8192 8192" "16384 16384" "32768 32768" "65536 65536" "8192 65536" "65536 16384" "4096 65536"; do
  set -- $s
  for f in on off; do
    python bench.py --rows $1 --cols $2 --steps 30 --warmup 4 --no-cpu-baseline --fused $f 2>&1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%6d x %6d fused=%-3s %9.0f it/s  %8.4f ms/step' % (d['config']['m'], d['config']['n'], '$f', d['value'], d['ms_per_step']))"
  done
done
