#!/bin/bash
mkdir -p gpurun_out/r05
cd /root/repo
echo "== row-space form: parity"
timeout -k 10 900 python -m pytest tests/test_gpu_model.py -q -m gpu -k "row_space or tight_padding" > gpurun_out/r05/t20.txt 2>&1; echo "rc $?"; tail -5 gpurun_out/r05/t20.txt
for cfg in "0 262144" "1 0" "1 8192"; do
set -- $cfg
VRDONE_ROW_SPACE=$1 VRDONE_ROWS_MIN_ROWS=$2 timeout -k 10 400 python bench.py --steps 4 --warmup 1 --no-alt --no-forward-test --no-train-step --no-cpu-baseline --no-shard-projection > gpurun_out/b_rag.json 2> gpurun_out/b_rag.err
python -c "
import json; d=json.load(open('gpurun_out/b_rag.json')); r=d['ragged_variant']; k=r['kernel_ms_per_step']
print('row_space $1 min_rows $2: headline', round(d['ms_per_step'],1), 'ragged', round(r['ms_per_step'],1), 'sum of kernels', round(sum(k.values()),1), {a:b for a,b in k.items() if b>0.5})"
done
