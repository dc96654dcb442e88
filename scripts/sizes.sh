#!/bin/bash
# it/s and per-kernel rates of the dense LASSO bench across matrix sizes (GPU box): bench.py lines (library loop, HIP-event records around every
# launch: they cost 2-3 us per launch), then wall-clock rates without records by driver.  Usage: bash scripts/sizes.sh
for s in "512 1024" "2048 2048" "4096 4096" "8192 8192" "16384 16384" "32768 32768" "65536 65536"; do
  set -- $s
  python bench.py --rows $1 --cols $2 --steps 40 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
pk=d['roofline']['per_kernel']
parts=' | '.join('%s %7.4f ms %5.0f GB/s' % (k.split('(')[1][:-1], v['avg_ms'], v['GB/s']) for k,v in pk.items() if v['launches'])
print('%6d x %6d  %8.0f it/s  %8.4f ms/step | %s' % (d['config']['m'], d['config']['n'], d['value'], d['ms_per_step'], parts))"
done
# the same sizes WITHOUT event records around the launches, and who drives the loop (python / library / device): wall clock only
python scripts/probes/driver_cost.py 512 1024 2048 2048 4096 4096 8192 8192 16384 16384 32768 32768

