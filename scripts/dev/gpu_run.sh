#!/bin/bash
# the current gpurun call (overwritten per call; results land in gpurun_out/ and, summarised, in profiles/ and LABNOTES.md)
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r06_g5; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -q -m gpu --deselect tests/test_gpu_ops.py 2>&1 | tail -60 | tee $O/gpu_tests.txt
