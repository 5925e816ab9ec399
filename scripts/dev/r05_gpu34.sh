#!/bin/bash
cd /root/repo
timeout -k 10 600 python scripts/dev/ragged_sweep.py rows:4096:1 rows:4096:2 rows:4096:3 rows:4096:4 rows:4096:1 rows:4096:2 2>&1 | grep -v amdgpu.ids | cut -c1-120
timeout -k 10 600 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "row_space" 2>&1 | tail -2
