#!/bin/bash
# round-5 GPU call 6: issue-priority modes 2/3/4; f16 backward GEMMs (gradient distance to the float64 reference)
mkdir -p gpurun_out/r05
cd /root/repo
for pr in 2 3 4 2 3 4; do
  echo "== gemm5_lab prio=$pr"; GEMM_LAB_F16=1 VRD_BIG_PRIO=$pr timeout -k 10 120 scripts/lab/r05/gemm5_lab_dma1 0 | grep -v "consumer 0\|producer 0" | grep -A2 "chunk1024\|mlp up\|mlp down" | tee -a gpurun_out/r05/gemm5_lab_prio$pr.txt
done
echo "== whole step A/B"
for pr in 2 3 4 2 3 4; do
VRD_BIG_PRIO=$pr timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-alt --no-ragged --no-cpu-baseline --no-forward-test --no-train-step --no-shard-projection 2>/dev/null > gpurun_out/r05/b_prio.json
python -c "import json,sys; d=json.load(open('gpurun_out/r05/b_prio.json')); k=d['kernel_ms_per_step']; print('prio $pr', round(d['ms_per_step'],2), 'ms  big', round(k.get('gemm_x3_big',0),2), 'frac', round(d['roofline']['frac'],4))" | tee -a gpurun_out/r05/ab_prio.txt
done
echo "== backward kernels"
timeout -k 10 900 python -m pytest tests/test_gpu_backward.py -x -q -m gpu > gpurun_out/r05/bwd_tests.txt 2>&1; echo "rc $?"; tail -3 gpurun_out/r05/bwd_tests.txt
echo "== gradients vs float64 reference, f16 backward"
timeout -k 10 900 python -m pytest tests/test_gpu_train.py -q -m gpu -s -k "float64 or matches_reference" > gpurun_out/r05/grad_f64_f16bwd.txt 2>&1; echo "rc $?"; grep "ours - ref64\|passed\|failed\|beyond" gpurun_out/r05/grad_f64_f16bwd.txt | cut -c1-300
echo "== gradients vs float64 reference, bf16 backward (VRDONE_F16_BACKWARD=0)"
VRDONE_F16_BACKWARD=0 timeout -k 10 900 python -m pytest tests/test_gpu_train.py -q -m gpu -s -k "float64" > gpurun_out/r05/grad_f64_bf16bwd.txt 2>&1; echo "rc $?"; grep "ours - ref64\|passed\|failed" gpurun_out/r05/grad_f64_bf16bwd.txt | cut -c1-300
echo "== rest of the training tests"
timeout -k 10 900 python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "not float64 and not matches_reference" > gpurun_out/r05/train_tests.txt 2>&1; echo "rc $?"; tail -3 gpurun_out/r05/train_tests.txt
