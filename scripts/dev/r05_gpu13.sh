#!/bin/bash
mkdir -p gpurun_out/r05
cd /root/repo
echo "== backward kernels"
timeout -k 10 900 python -m pytest tests/test_gpu_backward.py -x -q -m gpu > gpurun_out/r05/t13b.txt 2>&1; echo "rc $?"; tail -3 gpurun_out/r05/t13b.txt
echo "== training tests"
timeout -k 10 1200 python -m pytest tests/test_gpu_train.py -x -q -m gpu -s > gpurun_out/r05/t13t.txt 2>&1; echo "rc $?"; grep "relative gradient error\|passed\|failed" gpurun_out/r05/t13t.txt | cut -c1-200
echo "== vidor-size training step (48 pairs x 512 frames)"
for fb in 1 0; do
  echo "-- VRDONE_F16_BACKWARD=$fb"; VRDONE_F16_BACKWARD=$fb timeout -k 10 300 python scripts/train_step.py --config vidor --pairs 48 --steps 6 2>&1 | grep "^step [345]"
done
