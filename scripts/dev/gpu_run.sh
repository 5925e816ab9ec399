#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r06_g23; mkdir -p $O
VRD_BIG_PERSIST_ROWIN=1 timeout -k 10 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm or conv or transformer" 2>&1 | tail -4 | tee $O/ops_tests.txt || exit 1
Q="--steps 4 --warmup 1 --no-alt --no-ragged --no-cpu-baseline --no-forward-test --no-train-step --no-shard-projection"
ab() { ( [ -n "$2" ] && export $2; timeout -k 10 200 python bench.py $Q 2>$O/bench_$1.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$1', round(d['ms_per_step'],2), 'ms', {n: round(v,2) for n,v in sorted(k.items(), key=lambda t:-t[1])[:4]}, d['roofline']['frac'])" ) | tee -a $O/ab.txt; }
ab base "" && ab prow VRD_BIG_PERSIST_ROWIN=1 && ab base "" && ab prow VRD_BIG_PERSIST_ROWIN=1
