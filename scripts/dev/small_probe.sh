#!/bin/bash
# dev aid: where the step time goes at a rank's share of the strong-scaling job (256 pairs)
python bench.py --pairs ${1:-256} --steps 8 --warmup 3 --no-alt --no-ragged --no-cpu-baseline --no-forward-test --no-train-step 2>/dev/null | \
  python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']
print('ms_per_step', round(d['ms_per_step'],2), 'kernel sum', round(sum(k.values()),2))
for n,v in sorted(k.items(), key=lambda x:-x[1]): print('  %-22s %6.2f' % (n, v))
print({x: d[x] for x in d if 'launch' in x or 'host' in x})
"
