#!/bin/bash
mkdir -p gpurun_out/r05
cd /root/repo
timeout -k 10 1150 python -m pytest tests -x -q -m gpu > gpurun_out/r05/t26_full.txt 2>&1; echo "rc $?"; tail -6 gpurun_out/r05/t26_full.txt
