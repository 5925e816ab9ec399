#!/bin/bash
# the current gpurun call (overwritten per call; results land in gpurun_out/ and, summarised, in profiles/ and LABNOTES.md)
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r06_g7; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu 2>&1 | tail -15 | tee $O/ops_tests.txt || exit 1
Q="--steps 6 --warmup 2 --no-alt --no-ragged --no-cpu-baseline --no-forward-test --no-train-step --no-shard-projection"
ab() { # label, lib, extra env
  ( [ "$2" = tree ] || export VRDONE_HIP_LIB=$PWD/scripts/lab/libs/$2; [ -n "$3" ] && export $3; timeout -k 10 200 python bench.py $Q 2>$O/bench_$1.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$1', round(d['ms_per_step'],2), 'ms', {n: round(v,2) for n,v in sorted(k.items(), key=lambda t:-t[1])[:8]}, 'frac', d['roofline']['frac'])" ) | tee -a $O/ab.txt
}
ab r05 libvrdone_r05.so "" && ab new tree "" && ab r05 libvrdone_r05.so "" && ab new tree "" || exit 1
export GEMM_LAB_F16=1 GEMM_LAB_SHAPES=30 GEMM_LAB_SUSTAIN_MS=1200
for rep in 1 2; do
for b in carry swz; do
    echo "== $b" | tee -a $O/lab.txt
    timeout -k 10 120 scripts/lab/r06/bin/gemm6_$b 0 2>&1 | grep -v "^lab mode\|consumer\|producer" | tee -a $O/lab.txt || exit 1
done
done
