#!/bin/bash
mkdir -p gpurun_out/r05
cd /root/repo
timeout -k 10 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "filler or one_row_space" > gpurun_out/r05/t37.txt 2>&1; echo "rc $?"; tail -25 gpurun_out/r05/t37.txt
