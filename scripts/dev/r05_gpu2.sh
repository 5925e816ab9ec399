#!/bin/bash
# round-5 GPU call 2: persistent 256x256 GEMM (VRD_BIG_PERSIST=1): correctness, per-tile stamps, whole step A/B
set -o pipefail
mkdir -p gpurun_out/r05
cd /root/repo
echo "== GEMM tests, persistent"
VRD_BIG_PERSIST=1 timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm or conv" 2>&1 | tail -5 | tee gpurun_out/r05/persist_tests_ops.txt || exit 1
for pv in 0 1 0 1; do
  echo "== gemm5_lab persist=$pv f16 random"; GEMM_LAB_F16=1 VRD_BIG_PERSIST=$pv timeout -k 10 120 scripts/lab/r05/gemm5_lab_dma1 0 | grep -v "consumer 0\|producer 0" | tee -a gpurun_out/r05/gemm5_lab_persist$pv.txt
done
echo "== whole step A/B"
for pv in 0 1 0 1; do
VRD_BIG_PERSIST=$pv timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-alt --no-ragged --no-cpu-baseline --no-forward-test --no-train-step 2>/dev/null | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('persist $pv', round(d['ms_per_step'],2), 'ms  flash', round(k.get('attn_flash',0),2), 'big', round(k.get('gemm_x3_big',0),2), 'frac', round(d['roofline']['frac'],4))" | tee -a gpurun_out/r05/ab_persist.txt
done
echo "== model tests, persistent"
VRD_BIG_PERSIST=1 timeout -k 10 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r05/persist_tests_model.txt
