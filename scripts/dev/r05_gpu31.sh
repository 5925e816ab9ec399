#!/bin/bash
# persistent 256 x 256 GEMM: start offsets between the CUs of an XCD (VRD_BIG_PSTAGGER x 128 cycles per CU slot)
cd /root/repo
L="--steps 5 --warmup 2 --no-alt --no-ragged --no-forward-test --no-train-step --no-shard-projection --no-cpu-baseline"
for ps in 0 1 2 4 8 0 2; do
VRD_BIG_PSTAGGER=$ps timeout -k 10 300 python bench.py $L 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pstagger $ps', round(d['ms_per_step'],2), round(d['roofline']['frac'],4), d['kernel_ms_per_step']['gemm_x3_big'])"
done
