#!/bin/bash
mkdir -p gpurun_out/r05
cd /root/repo
timeout -k 10 1100 python -m pytest tests/test_gpu_model.py -x -q -m gpu > gpurun_out/r05/t35.txt 2>&1; echo "rc $?"; tail -3 gpurun_out/r05/t35.txt
