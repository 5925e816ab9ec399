#!/bin/bash
# dwconv_ln strip length / strips per wave at the headline shape (the kernel gained the row-group table this round)
cd /root/repo
L="--steps 4 --warmup 1 --no-alt --no-forward-test --no-train-step --no-shard-projection --no-cpu-baseline"
for cfg in "0 0" "8 0" "12 0" "24 0" "32 0" "0 2" "0 0"; do
set -- $cfg
VRD_DW_RW=$1 VRD_DW_SPW=$2 timeout -k 10 300 python bench.py $L 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('rw $1 spw $2: step', round(d['ms_per_step'],2), 'dwconv_ln', d['kernel_ms_per_step']['dwconv_ln'], 'ragged', round(d['ragged_variant']['ms_per_step'],2), d['ragged_variant']['kernel_ms_per_step']['dwconv_ln'])"
done
