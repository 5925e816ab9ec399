#!/bin/bash
# dev aid: A/B an environment switch on the bench step.  usage: ab_env.sh VAR value value ...
VAR=$1; shift
for v in "$@"; do
  env $VAR=$v python bench.py --steps 8 --warmup 3 --no-alt --no-ragged --no-cpu-baseline --no-forward-test --no-train-step 2>/dev/null | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$VAR=$v', round(d['ms_per_step'],2), 'ms  big', round(k.get('gemm_bf16x3_big',0),2), 'frac', round(d['roofline']['frac'],4))"
done
