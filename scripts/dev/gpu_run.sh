#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r06_train; mkdir -p $O
timeout -k 10 800 python -m pytest tests/test_gpu_train.py tests/test_gpu_backward.py -x -q > $O/tests.txt 2>&1; echo "pytest rc $?"; tail -3 $O/tests.txt
grep -q passed $O/tests.txt || { tail -40 $O/tests.txt; exit 1; }
timeout -k 10 400 python bench.py --steps 2 --warmup 1 --no-alt --no-ragged --no-cpu-baseline --no-forward-test --no-shard-projection > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_train/bench.json').read().strip().splitlines()[-1])
ts=d['train_step']; print('train', ts['ms_forward_backward'], ts['ms_forward_backward_hip_graphs'], ts['launches_forward_backward'], {k:v for k,v in ts.get('vidor_48x512',{}).items() if 'ms' in k or 'launch' in k})
PY
