#!/bin/bash
mkdir -p gpurun_out/r05
cd /root/repo
timeout -k 10 900 python -m pytest tests/test_gpu_bench.py -x -q -m gpu > gpurun_out/r05/t33.txt 2>&1; echo "rc $?"; tail -3 gpurun_out/r05/t33.txt
