#!/bin/bash
# round-5 GPU call 12: 16x16x32 MFMA shape in the (persistent) 256x256 kernel, no register spills in the K loop now
mkdir -p gpurun_out/r05
cd /root/repo
for m in 0 1 0 1; do
  echo "== gemm5_lab M16=$m"; GEMM_LAB_F16=1 VRD_BIG_M16=$m timeout -k 10 120 scripts/lab/r05/gemm5_lab_dma1 0 > gpurun_out/r05/lab12.txt 2>&1; grep -q fault gpurun_out/r05/lab12.txt && { echo FAULT; exit 1; }
  grep -v "consumer 0\|producer 0" gpurun_out/r05/lab12.txt | grep -A3 "chunk1024\|mlp up\|mlp down" | tee -a gpurun_out/r05/gemm5_lab_m16_$m.txt
done
echo "== tests with M16"
VRD_BIG_M16=1 timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm or conv" > gpurun_out/r05/t12.txt 2>&1; echo "rc $?"; tail -2 gpurun_out/r05/t12.txt
echo "== whole step A/B"
for m in 0 1 0 1; do
VRD_BIG_M16=$m timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-alt --no-ragged --no-cpu-baseline --no-forward-test --no-train-step --no-shard-projection 2>/dev/null > gpurun_out/r05/b12.json
python -c "import json,sys; d=json.load(open('gpurun_out/r05/b12.json')); k=d['kernel_ms_per_step']; print('M16=$m', round(d['ms_per_step'],2), 'ms  big', round(k.get('gemm_x3_big',0),2), 'frac', round(d['roofline']['frac'],4))" | tee -a gpurun_out/r05/ab_m16.txt
done
