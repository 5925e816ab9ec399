#!/bin/bash
# round-5 GPU call 5: full GPU suite (output to a file, not behind a pipe) + issue-priority experiment
mkdir -p gpurun_out/r05
cd /root/repo
for pr in 0 1 2 0 1 2; do
  echo "== gemm5_lab prio=$pr"; GEMM_LAB_F16=1 VRD_BIG_PRIO=$pr timeout -k 10 120 scripts/lab/r05/gemm5_lab_dma1 0 | grep -v "consumer 0\|producer 0" | grep -A2 "chunk1024\|mlp up" | tee -a gpurun_out/r05/gemm5_lab_prio$pr.txt
done
echo "== whole step A/B"
for pr in 0 1 2 0 1 2; do
VRD_BIG_PRIO=$pr timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-alt --no-ragged --no-cpu-baseline --no-forward-test --no-train-step --no-shard-projection 2>/dev/null > gpurun_out/r05/b_prio.json
python -c "import json,sys; d=json.load(open('gpurun_out/r05/b_prio.json')); k=d['kernel_ms_per_step']; print('prio $pr', round(d['ms_per_step'],2), 'ms  big', round(k.get('gemm_x3_big',0),2), 'frac', round(d['roofline']['frac'],4))" | tee -a gpurun_out/r05/ab_prio.txt
done
echo "== full GPU suite"
timeout -k 10 1700 python -m pytest tests -x -q -m gpu > gpurun_out/r05/gpu_suite.txt 2>&1
echo "suite rc $?"
tail -15 gpurun_out/r05/gpu_suite.txt
