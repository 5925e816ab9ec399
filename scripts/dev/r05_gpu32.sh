#!/bin/bash
# what the per-launch HIP events of bench.py's profiler cost inside the timed region
cd /root/repo
L="--steps 5 --warmup 2 --no-alt --no-ragged --no-forward-test --no-train-step --no-shard-projection --no-cpu-baseline"
for rep in 1 2 3; do
timeout -k 10 300 python bench.py $L 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('events on ', round(d['ms_per_step'],2), round(d['roofline']['frac'],4))"
timeout -k 10 300 python bench.py $L --no-prof 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('events off', round(d['ms_per_step'],2))"
done
