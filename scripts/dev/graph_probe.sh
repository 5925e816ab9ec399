#!/bin/bash
# dev aid: training step with and without HIP graphs
python scripts/train_step.py --steps 14 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('eager ', [round(x,1) for x in d['step_ms']], d['nonfinite_grads'], d['params_without_grad'])"
python scripts/train_step.py --steps 14 --graphs 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('graphs', [round(x,1) for x in d['step_ms']], d['nonfinite_grads'], d['params_without_grad'])"
