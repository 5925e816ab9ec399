#!/bin/bash
# round-5 GPU call 7: backward / training tests with the f16 backward, vidor-size step timing, lab stamps of the tile start
mkdir -p gpurun_out/r05
cd /root/repo
echo "== backward kernels"
timeout -k 10 900 python -m pytest tests/test_gpu_backward.py -x -q -m gpu -s > gpurun_out/r05/bwd_tests.txt 2>&1; echo "rc $?"; grep "dW error\|passed\|failed" gpurun_out/r05/bwd_tests.txt | tail -8
echo "== training tests"
timeout -k 10 1200 python -m pytest tests/test_gpu_train.py -x -q -m gpu > gpurun_out/r05/train_tests.txt 2>&1; echo "rc $?"; tail -4 gpurun_out/r05/train_tests.txt
echo "== vidor-size training step (48 pairs x 512 frames)"
for fb in 1 0 1 0; do
  echo "-- VRDONE_F16_BACKWARD=$fb"; VRDONE_F16_BACKWARD=$fb timeout -k 10 300 python scripts/train_step.py --config vidor --pairs 48 --steps 6 2>&1 | tail -4
done
echo "== lab: tile start"
GEMM_LAB_F16=1 timeout -k 10 120 scripts/lab/r05/gemm5_lab_dma1 0 | grep -v "consumer 0\|producer 0" | grep -A3 "chunk1024\|mlp up"
