#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r06_g12; mkdir -p $O
timeout -k 10 120 scripts/lab/r06/bin/hbm_rw | tee $O/hbm_rw.txt
Q="--steps 4 --warmup 1 --no-alt --no-ragged --no-cpu-baseline --no-forward-test --no-train-step --no-shard-projection"
ab() { # label, lib, extra env
  ( [ "$2" = tree ] || export VRDONE_HIP_LIB=$PWD/scripts/lab/libs/$2; [ -n "$3" ] && export $3; timeout -k 10 200 python bench.py $Q 2>$O/bench_$1.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$1', round(d['ms_per_step'],2), 'ms', {n: round(v,2) for n,v in sorted(k.items(), key=lambda t:-t[1])[:8]})" ) | tee -a $O/ab.txt
}
for rw in 16 9 12 15 17 18 20 24 16; do ab rw$rw tree VRD_DW_RW=$rw; done
