#!/bin/bash
# the full-length headline batch at padded length 264 (VRDONE_TIGHT_UNIT=8 / 16) instead of 288
cd /root/repo
L="--steps 4 --warmup 1 --no-alt --no-ragged --no-forward-test --no-train-step --no-shard-projection --no-cpu-baseline"
for u in 32 8 16 32 8; do
VRDONE_TIGHT_UNIT=$u timeout -k 10 300 python bench.py $L 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('unit $u: step', round(d['ms_per_step'],2), 'gemm', k['gemm_x3_big'], 'dwconv', k['dwconv_ln'], 'ln', k['layernorm'], 'attn', k['attn_flash'], 'local', k['local_attn'])"
done
