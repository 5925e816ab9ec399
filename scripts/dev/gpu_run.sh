#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r06_g26; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "few_channel" 2>&1 | tail -12 | tee $O/ops_tests.txt || exit 1
Q="--steps 6 --warmup 2 --no-alt --no-ragged --no-cpu-baseline --no-forward-test --no-train-step --no-shard-projection"
ab() { ( [ -n "$2" ] && export $2; timeout -k 10 200 python bench.py $Q 2>$O/bench_$1.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$1', round(d['ms_per_step'],2), 'ms', {n: round(v,2) for n,v in sorted(k.items(), key=lambda t:-t[1])}, d['roofline']['frac'])" ) | tee -a $O/ab.txt; }
ab fused "" && ab apart VRDONE_CONV_LN=0 && ab fused "" && ab apart VRDONE_CONV_LN=0
timeout -k 10 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "golden or row_space or tight or oracle or entity_stage or sharing" 2>&1 | tail -6 | tee $O/model_tests.txt
