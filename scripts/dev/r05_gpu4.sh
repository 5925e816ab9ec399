#!/bin/bash
# round-5 GPU call 4: persistent 256x256 GEMM v3 (look-ahead inside the last K steps, no compiler-inserted drains) + range flag
set -o pipefail
mkdir -p gpurun_out/r05
cd /root/repo
rm -f gpurun_out/r05/gemm5_lab_persist*.txt gpurun_out/r05/ab_persist.txt
echo "== GEMM tests (persistent default)"
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm or conv" 2>&1 | tail -5 || exit 1
for pv in 0 1 0 1; do
  echo "== gemm5_lab persist=$pv f16 random"; GEMM_LAB_F16=1 VRD_BIG_PERSIST=$pv timeout -k 10 120 scripts/lab/r05/gemm5_lab_dma1 0 | grep -v "consumer 0\|producer 0" | tee -a gpurun_out/r05/gemm5_lab_persist$pv.txt
done
echo "== whole step A/B"
for pv in 0 1 0 1; do
VRD_BIG_PERSIST=$pv timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-alt --no-ragged --no-cpu-baseline --no-forward-test --no-train-step 2>/dev/null | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('persist $pv', round(d['ms_per_step'],2), 'ms  flash', round(k.get('attn_flash',0),2), 'big', round(k.get('gemm_x3_big',0),2), 'dwconv', round(k.get('dwconv_ln',0),2), 'ln', round(k.get('layernorm',0),2), 'frac', round(d['roofline']['frac'],4))" | tee -a gpurun_out/r05/ab_persist.txt
done
echo "== full GPU suite"
timeout -k 10 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r05/gpu_suite.txt
