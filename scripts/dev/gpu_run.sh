#!/bin/bash
# one more sample of the default bench line on the final tree (another box of the pool)
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r06_collect; mkdir -p $O
timeout -k 10 600 python bench.py --steps 5 --warmup 2 > $O/bench_sample2.json 2> $O/bench_sample2.err || exit 1
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_collect/bench_sample2.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['roofline']['frac'], d['kernel_ms_per_step'])
PY
