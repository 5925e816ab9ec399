#!/bin/bash
mkdir -p gpurun_out/r05
cd /root/repo
echo "== backward kernels"
timeout -k 10 900 python -m pytest tests/test_gpu_backward.py -x -q -m gpu > gpurun_out/r05/t15b.txt 2>&1; echo "rc $?"; tail -3 gpurun_out/r05/t15b.txt
echo "== training tests"
timeout -k 10 1200 python -m pytest tests/test_gpu_train.py -x -q -m gpu > gpurun_out/r05/t15t.txt 2>&1; echo "rc $?"; tail -3 gpurun_out/r05/t15t.txt
echo "== vidor-size training step (48 pairs x 512 frames)"
for g in "" "--graphs"; do
for fb in 1 0; do
  echo "-- $g VRDONE_F16_BACKWARD=$fb"; VRDONE_F16_BACKWARD=$fb timeout -k 10 300 python scripts/train_step.py --config vidor --pairs 48 --steps 8 $g 2>&1 | grep "^step [4567]"
done
done
