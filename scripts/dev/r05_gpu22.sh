#!/bin/bash
mkdir -p gpurun_out/r05
cd /root/repo
echo "== model tests"
timeout -k 10 1100 python -m pytest tests/test_gpu_model.py -x -q -m gpu > gpurun_out/r05/t22.txt 2>&1; echo "rc $?"; tail -8 gpurun_out/r05/t22.txt
for rs in 1 0; do
VRDONE_ROW_SPACE=$rs timeout -k 10 500 python bench.py --steps 4 --warmup 1 --no-alt --no-train-step --no-cpu-baseline --no-shard-projection > gpurun_out/b_ft$rs.json 2> gpurun_out/b_ft$rs.err
python -c "
import json; d=json.load(open('gpurun_out/b_ft$rs.json')); r=d['ragged_variant']
print('row_space $rs: headline', round(d['ms_per_step'],1), 'ragged', round(r['ms_per_step'],1), 'forward_test', json.dumps(d.get('forward_test'))[:900])"
done
