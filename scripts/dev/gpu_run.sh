#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r06_g25; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm or conv or transformer" 2>&1 | tail -4 | tee $O/ops_tests.txt || exit 1
export GEMM_LAB_F16=1 GEMM_LAB_SHAPES=18 GEMM_LAB_SUSTAIN_MS=1200
for rep in 1 2; do for b in tree nob1; do echo "== $b" | tee -a $O/lab.txt; timeout -k 10 120 scripts/lab/r06/bin/gemm6_$b 0 2>&1 | grep -v "^lab mode\|consumer\|producer\|waves" | tee -a $O/lab.txt; done; done
Q="--steps 6 --warmup 2 --no-alt --no-ragged --no-cpu-baseline --no-forward-test --no-train-step --no-shard-projection"
ab() { ( [ "$2" = tree ] || export VRDONE_HIP_LIB=$PWD/scripts/lab/libs/$2; timeout -k 10 200 python bench.py $Q 2>$O/bench_$1.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$1', round(d['ms_per_step'],2), 'ms', {n: round(v,2) for n,v in sorted(k.items(), key=lambda t:-t[1])[:4]}, d['roofline']['frac'])" ) | tee -a $O/ab.txt; }
ab head libvrdone_head.so && ab nob1 tree && ab head libvrdone_head.so && ab nob1 tree
