#!/bin/bash
mkdir -p gpurun_out/r05
cd /root/repo
echo "== row-space form: parity"
timeout -k 10 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "row_space or tight_padding" > gpurun_out/r05/t18.txt 2>&1; echo "rc $?"; tail -15 gpurun_out/r05/t18.txt
echo "== ragged sweep"
timeout -k 10 600 python scripts/dev/ragged_sweep.py 262144:1 rows:65536 rows:32768 rows:16384 rows:8192 rows:0 2>&1 | grep -v amdgpu.ids
