#!/bin/bash
# dev aid: the bench step in exact-f32 mode with and without the 256 x 256 f32 tiles
for v in 0 1 0 1; do
  VRD_F32_BIG=$v python bench.py --precision f32 --steps 4 --warmup 2 --no-alt --no-ragged --no-cpu-baseline --no-forward-test --no-train-step 2>/dev/null | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; r=d['roofline']; print('VRD_F32_BIG=$v', round(d['ms_per_step'],1), 'ms', round(d['value']), 'pairs/s  gemm', round(k.get('gemm_f32_mfma',0),1), 'frac', round(r['frac'],3), r['kernel'])"
done
