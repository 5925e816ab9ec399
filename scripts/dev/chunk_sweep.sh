#!/bin/bash
# dev aid: pairs-per-launch-wave sweep of the bench step (cache residency vs tail effects)
for c in "$@"; do
  python bench.py --steps 6 --warmup 2 --pair-chunk $c --no-alt --no-ragged --no-cpu-baseline --no-forward-test 2>/dev/null | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('chunk', $c, round(d['ms_per_step'],2), 'ms', round(d['value']), 'pairs/s  big', round(k.get('gemm_bf16x3_big',0),2), 'dma', round(k.get('gemm_bf16x3_dma',0),2), 'dwconv', round(k['dwconv_ln'],2), 'ln', round(k['layernorm'],2))"
done
