#!/bin/bash
mkdir -p gpurun_out/r05
cd /root/repo
echo "== lab persist=1"
GEMM_LAB_F16=1 timeout -k 10 120 scripts/lab/r05/gemm5_lab_dma1 0 > gpurun_out/r05/lab11.txt 2>&1; echo "rc $?"
grep -v "consumer 0\|producer 0" gpurun_out/r05/lab11.txt | grep -A3 "chunk1024\|mlp up\|fault"
grep -q "fault" gpurun_out/r05/lab11.txt && exit 1
echo "== full GPU suite"
timeout -k 10 1700 python -m pytest tests -x -q -m gpu > gpurun_out/r05/gpu_suite.txt 2>&1
echo "suite rc $?"
tail -6 gpurun_out/r05/gpu_suite.txt
echo "== step"
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-alt --no-ragged --no-cpu-baseline --no-forward-test --no-train-step --no-shard-projection 2>/dev/null > gpurun_out/r05/b11.json
python -c "import json,sys; d=json.load(open('gpurun_out/r05/b11.json')); k=d['kernel_ms_per_step']; print('step', round(d['ms_per_step'],2), 'ms  big', round(k.get('gemm_x3_big',0),2), 'frac', round(d['roofline']['frac'],4))"
done
