#!/bin/bash
mkdir -p gpurun_out/r05
cd /root/repo
echo "== forward_test goldens"
timeout -k 10 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "forward_test" > gpurun_out/r05/t23.txt 2>&1; echo "rc $?"; tail -4 gpurun_out/r05/t23.txt
for cfg in "1 16384" "1 4096" "0 0"; do
set -- $cfg
VRDONE_ROW_SPACE=$1 VRDONE_ROWS_MIN_ROWS=$2 timeout -k 10 500 python bench.py --steps 4 --warmup 1 --no-alt --no-train-step --no-cpu-baseline --no-shard-projection > gpurun_out/b_ft.json 2> gpurun_out/b_ft.err
python -c "
import json; d=json.load(open('gpurun_out/b_ft.json')); r=d['ragged_variant']; k=r['kernel_ms_per_step']; f=d.get('forward_test') or {}
print('row_space $1 min_rows $2: headline', round(d['ms_per_step'],1), 'ragged', round(r['ms_per_step'],1), {a:b for a,b in k.items() if b>0.5}, 'forward_test', round(f.get('ms',0),1), 'from tracklets', round(f.get('from_tracklets',{}).get('forward_test_ms',0),1))"
done
