#!/bin/bash
# lab aid: the bench step with alternative builds of the library (timing ablations; results are wrong by design)
for lib in "$@"; do
  if [ "$lib" = base ]; then unset VRDONE_HIP_LIB; else export VRDONE_HIP_LIB=$PWD/scripts/lab/libs/$lib; fi
  python bench.py --steps 6 --warmup 2 --no-alt --no-ragged --no-cpu-baseline --no-forward-test --no-train-step 2>/dev/null | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$lib', round(d['ms_per_step'],2), 'ms  flash', round(k.get('attn_flash',0),2), 'big', round(k.get('gemm_x3_big',0),2))"
done
