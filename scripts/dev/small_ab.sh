#!/bin/bash
# dev aid: 64 x 64-tile GEMM variant on/off -- training step (graphs) and the eval step at a rank's 256-pair share
for v in 1 0 1 0; do
  VRD_X3_SMALL=$v python scripts/train_step.py --steps 14 --graphs 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=sorted(d['step_ms'][2:]); print('small=$v train graphs median', round(s[len(s)//2],1), 'min', round(s[0],1))"
  VRD_X3_SMALL=$v python bench.py --pairs 256 --steps 8 --warmup 3 --no-alt --no-ragged --no-cpu-baseline --no-forward-test --no-train-step 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('small=$v eval 256 pairs', round(d['ms_per_step'],2), 'ms  x3 gemm', round(k.get('gemm_bf16x3_mfma',0),2))"
done
