#!/bin/bash
# the switches a user can flip: row space off (bucket by bucket), buckets on side streams, f16 backward off, persistent GEMM off
mkdir -p gpurun_out/r05
cd /root/repo
VRDONE_ROW_SPACE=0 timeout -k 10 600 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "forward_test or tight or sharded" > gpurun_out/r05/t30a.txt 2>&1; echo "row space off rc $?"; tail -2 gpurun_out/r05/t30a.txt
VRDONE_ROW_SPACE=0 VRDONE_TIGHT_STREAMS=3 timeout -k 10 600 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "forward_test or tight" > gpurun_out/r05/t30b.txt 2>&1; echo "side streams rc $?"; tail -2 gpurun_out/r05/t30b.txt
VRDONE_F16_BACKWARD=0 timeout -k 10 600 python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "not float64" > gpurun_out/r05/t30c.txt 2>&1; echo "bf16 backward rc $?"; tail -2 gpurun_out/r05/t30c.txt
VRD_BIG_PERSIST=0 timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm" > gpurun_out/r05/t30d.txt 2>&1; echo "persist off rc $?"; tail -2 gpurun_out/r05/t30d.txt
