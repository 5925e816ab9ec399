#!/bin/bash
mkdir -p gpurun_out/r05
cd /root/repo
timeout -k 10 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_bench.py -x -q -m gpu -k "forward_test or row_space or tight or bench or smoke or loss" > gpurun_out/r05/t27.txt 2>&1; echo "rc $?"; tail -4 gpurun_out/r05/t27.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
