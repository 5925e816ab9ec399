#!/bin/bash
# dev aid: training step (HIP graphs) with and without the zero arena, interleaved
for v in 1 0 1 0; do
  VRD_ZERO_ARENA=$v python scripts/train_step.py --steps 14 --graphs 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=sorted(d['step_ms'][2:]); print('arena=$v graphs median', round(s[len(s)//2],1), 'min', round(s[0],1))"
  VRD_ZERO_ARENA=$v python scripts/train_step.py --steps 14 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=sorted(d['step_ms'][2:]); print('arena=$v eager  median', round(s[len(s)//2],1), 'min', round(s[0],1))"
done
