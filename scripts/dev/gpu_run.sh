#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r06_g19; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu 2>&1 | tail -5 | tee $O/ops_tests.txt || exit 1
timeout -k 10 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "f32" 2>&1 | tail -5 | tee $O/model_f32_tests.txt || exit 1
Q="--steps 3 --warmup 1 --no-alt --no-ragged --no-cpu-baseline --no-forward-test --no-train-step --no-shard-projection"
for i in 1 2; do
timeout -k 10 300 python bench.py $Q --precision f32 2>$O/bench_f32.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('f32', round(d['ms_per_step'],2), 'ms', {n: round(v,2) for n,v in sorted(k.items(), key=lambda t:-t[1])[:6]}, 'frac', round(d['roofline']['frac'],3), 'skipped', d['roofline']['padding_flops_skipped_per_launch'])" | tee -a $O/ab.txt
VRDONE_SKIP_PADDING=0 timeout -k 10 300 python bench.py $Q --precision f32 2>$O/bench_f32.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('f32 noskip', round(d['ms_per_step'],2), 'ms', {n: round(v,2) for n,v in sorted(k.items(), key=lambda t:-t[1])[:6]}, 'frac', round(d['roofline']['frac'],3))" | tee -a $O/ab.txt
done
timeout -k 10 300 python scripts/dev/ft_small.py 2>&1 | grep "row space True" | tee $O/ft_small.txt
