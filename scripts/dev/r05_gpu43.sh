#!/bin/bash
mkdir -p gpurun_out/r05
cd /root/repo
timeout -k 10 1100 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py -x -q -m gpu -k "row or tight or forward_test or filler or full_size_properties or dwconv or local" > gpurun_out/r05/t43.txt 2>&1; echo "rc $?"; tail -3 gpurun_out/r05/t43.txt
timeout -k 10 600 python scripts/dev/ragged_sweep.py rows:4096:2 rows:4096:2 2>&1 | grep "row space" | cut -c1-90
