#!/bin/bash
mkdir -p gpurun_out/r05
cd /root/repo
timeout -k 10 900 python scripts/dev/ragged_sweep.py 262144:1 262144:2 262144:3 131072:1 131072:2 131072:3 65536:1 65536:2 65536:3 65536:4 0:1 0:2 0:3 0:4 2>&1 | grep -v amdgpu.ids
echo "== training step: range guard on / off (bf16 backward)"
for g in "" "--graphs"; do
for rg in 1 0; do
  echo "-- $g VRDONE_RANGE_GUARD=$rg"; VRDONE_RANGE_GUARD=$rg VRDONE_F16_BACKWARD=0 timeout -k 10 300 python scripts/train_step.py --config vidor --pairs 48 --steps 8 $g 2>&1 | grep "^step [4567]"
done
done
