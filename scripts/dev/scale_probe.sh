#!/bin/bash
# dev aid: step time of the per-rank share of the 2048-pair job at N = 1, 2, 4, 8 (what strong scaling leaves each rank)
for p in 2048 1024 512 256; do
  python bench.py --pairs $p --steps 8 --warmup 3 --no-alt --no-ragged --no-cpu-baseline --no-forward-test --no-train-step 2>/dev/null | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pairs', $p, round(d['ms_per_step'],2), 'ms', round(d['value']), 'pairs/s')"
done
